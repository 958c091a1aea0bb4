"""CPU experiment (test infrastructure): where does the whole-Unet distance to the fp32 oracle come from?

Runs the Unet planner's launch list through the plan interpreter (tests/plan_interp.py: every op restated from its contract, fp16
storage, fp32 arithmetic — the rounding points of the HIP path without its kernels) against the oracle, with switches that keep chosen
tensors at higher precision:

  --stream-lo   the residual stream (outputs of ResnetBlock tails, attention / feed-forward residual adds) carries an fp16 "lo" word
                beside the fp16 value (value = hi + lo): only the next residual add reads it, every GEMM / conv still reads hi.

    python tools/parity_budget.py [--config u1|u2|hd32|memeff] [--size 64] [--stream-lo]
"""
from __future__ import annotations

import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

README_U1 = dict(dim=32, cond_dim=512, dim_mults=(1, 2, 4, 8), num_resnet_blocks=3, layer_attns=(False, True, True, True),
                 layer_cross_attns=(False, True, True, True))
README_U2 = dict(dim=32, cond_dim=512, dim_mults=(1, 2, 4, 8), num_resnet_blocks=(2, 4, 8, 8), layer_attns=(False, False, False, True),
                 layer_cross_attns=(False, False, False, True), lowres_cond=True)
MEMEFF = dict(dim=32, cond_dim=64, dim_mults=(1, 2, 4), num_resnet_blocks=(1, 2, 2), layer_attns=(False, False, True),
              layer_cross_attns=(False, True, True), memory_efficient=True, lowres_cond=True, attn_heads=4)
HD32 = dict(README_U1, attn_dim_head=32, attn_heads=16)
CONFIGS = dict(u1=README_U1, u2=README_U2, memeff=MEMEFF, hd32=HD32)


def nerr(a, b):
    return float((a.float() - b.float()).norm() / b.float().norm())


def case(kw, S, B, seed=0):
    from imagen_pytorch_amd import Unet
    torch.manual_seed(seed)
    u = Unet(**kw).eval()
    torch.nn.init.normal_(u.final_conv.weight, std=0.05)
    torch.nn.init.normal_(u.final_conv.bias, std=0.05)
    sd = {k: v.clone() for k, v in u.state_dict().items()}
    x, t = torch.randn(B, 3, S, S), torch.tensor([0.3, -1.2, 0.9, 2.0][:B])
    te = torch.randn(B, 24, kw.get("text_embed_dim", 768))
    mask = torch.ones(B, 24, dtype=torch.bool)
    mask[1, 18:] = False
    extra = dict(lowres_cond_img=torch.randn(B, 3, S, S), lowres_noise_times=torch.full((B,), 0.5)) if kw.get("lowres_cond") else {}
    return u, sd, x, t, te, mask, extra


def run_interp(u, x, t, te, mask, extra, S, B, interp_cls):
    from imagen_pytorch_amd.engine import UnetEngine
    rows = 2 * B
    eng = UnetEngine(u, rows, B, S, "cpu", dry=True)
    keep = torch.ones(rows, dtype=torch.bool)
    keep[B:] = False
    eng.set_conditioning(text_embeds=te, text_mask=mask, keep=keep, lowres_noise_times=extra.get("lowres_noise_times"))
    it = interp_cls()
    for tt in (eng.x_in, eng.lowres_in, eng.times, eng.lowres_times, eng.out, eng.keep_u8, eng.src_idx, eng.arange_idx, eng.t_const.t):
        it.mem.register(tt)
    it.run(eng._static_plans[te.shape[1]][0])
    eng.x_in.copy_(x)
    if eng.lowres:
        eng.lowres_in.copy_(extra["lowres_cond_img"])
    eng.times.copy_(t.repeat(rows // B))
    it.run(eng.step_plan)
    return eng.out.clone(), eng


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="u1")
    ap.add_argument("--size", type=int, default=64)
    ap.add_argument("--batch", type=int, default=2)
    ap.add_argument("--seeds", type=int, nargs="*", default=[0], help="weight / input seeds (0 = the GPU test's case)")
    args = ap.parse_args()
    from imagen_pytorch_amd import ops
    from oracle import unet_oracle as uo
    from plan_interp import Interpreter
    ops.KEEP_REFERENCE_WEIGHTS = True
    kw, S, B = CONFIGS[args.config], args.size, args.batch
    for seed in args.seeds:
        u, sd, x, t, te, mask, extra = case(kw, S, B, seed)
        with torch.no_grad():
            ref = uo.unet_forward(sd, kw, x, t, text_embeds=te, text_mask=mask, **extra)
            ref_null = uo.unet_forward(sd, kw, x, t, text_embeds=te, text_mask=mask, cond_drop_prob=1.0, **extra)
            r16 = lambda v: v.half().float() if torch.is_tensor(v) and v.is_floating_point() else v
            sd16 = {k: r16(v) for k, v in sd.items()}
            ex16 = {k: r16(v) for k, v in extra.items()}
            calib = nerr(uo.unet_forward(sd16, kw, r16(x), t, text_embeds=r16(te), text_mask=mask, cond_drop_prob=1.0, **ex16), ref_null)
        out, eng = run_interp(u, x, t, te, mask, extra, S, B, Interpreter)
        print(f"{args.config}@{S} B{B} seed {seed}: fp16 params + inputs only (null) {calib:.3e};  plan through the interpreter: cond {nerr(out[:B], ref):.3e} "
              f"null {nerr(out[B:], ref_null):.3e}", flush=True)
        ops.REFERENCE_WEIGHTS.clear()


if __name__ == "__main__":
    main()
