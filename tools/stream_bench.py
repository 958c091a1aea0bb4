#!/usr/bin/env python
"""Kernel-level A/B of the 32-channel 3x3 conv families (conv_stream.hip = family 3, conv_pro.hip = family 6, the wave-specialised kernel
= family 0) on the benchmark's 256^2 / 128^2 layer shapes (rows 16 = batch 8 under CFG): every candidate is launched `--iters` times over
four rotating input / output buffer sets (same weights and statistics), timed with events on the launch stream, and its output compared
with the first candidate's.

    python tools/stream_bench.py [--iters 30] [--out file.jsonl]

One JSON line per (shape, variant, candidate): us per launch, algorithmic GB/s (inputs + output once), normwise distance to the first candidate."""
import argparse
import json
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from imagen_pytorch_amd import ops  # noqa: E402

# (H, C1, C2, Cout): the up path's concat convs and the single-input convs of the 256^2 and 128^2 levels
SHAPES = [(256, 32, 32, 32), (256, 32, 0, 32), (128, 64, 32, 64), (128, 64, 0, 64), (128, 32, 0, 32)]
VARIANTS = {"pro+post": dict(pro=True, post=True), "pro+ssq": dict(pro=True, post=False), "raw+ssq": dict(pro=False, post=False)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=30)
    ap.add_argument("--rows", type=int, default=16)
    ap.add_argument("--out", default=None)
    ap.add_argument("--shapes", nargs="*", default=None, help="H:C1:C2:Cout, e.g. 256:32:32:32 (default: all of SHAPES)")
    ap.add_argument("--variants", nargs="*", default=None, help="subset of pro+post pro+ssq raw+ssq")
    ap.add_argument("--cands", nargs="*", default=None, help="subset of stream pro fam0")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    B = args.rows
    g = torch.Generator().manual_seed(0)
    lines = []
    shapes = [tuple(map(int, s.split(':'))) for s in args.shapes] if args.shapes else SHAPES
    for H, C1, C2, Co in shapes:
        Cin = C1 + C2
        cands = {"stream": (ops.stream_cfg(), 16, 16) if Co == 32 else "dma", "pro": (ops.pro_cfg(Co), 8, 16), "fam0": None}
        w = torch.randn(Co, Cin, 3, 3, generator=g) / math.sqrt(9 * Cin)
        pw = ops.pack_weight(w, torch.randn(Co, generator=g) * 0.1, dev, G=4)
        sets = []
        for _ in range(4):
            x1 = ops.act_from_nchw((torch.randn(B, C1, H, H, generator=g) * 0.7).to(dev))
            x2 = ops.act_from_nchw((torch.randn(B, C2, H, H, generator=g) * 0.7).to(dev)) if C2 else None
            sa = (x1.t.float() ** 2).sum(-1).reshape(-1).contiguous()
            sb = (x2.t.float() ** 2).sum(-1).reshape(-1).contiguous() if C2 else None
            sets.append((x1, x2, sa, sb, ops.new_act(B, H, H, Co, dev), torch.empty(B * H * H, device=dev)))
        pa = (1 + 0.2 * torch.randn(Cin, generator=g)).to(dev)
        post = dict(pa=(1 + 0.2 * torch.randn(B, Co, generator=g)).to(dev), ps=(0.2 * torch.randn(B, Co, generator=g)).to(dev), pstride=Co)
        for vname, v in VARIANTS.items():
            first = None
            if args.variants and vname not in args.variants:
                continue
            if Co == 64 and not v["post"] and v["pro"]:
                continue                      # (64 couts: the family emits no ssq_out; the benchmark's launches are pro + post and raw)
            for cname, cfg in cands.items():
                if (cname == "fam0" and not v["pro"]) or (args.cands and cname not in args.cands):
                    continue
                if cfg == "dma":              # 64 couts: the all-DMA family takes the raw single-input launches only
                    if v["pro"] or C2:
                        continue
                    cfg = ops.pick_cfg(4, Co, H, H, B, 3, 3, 1, full_cout=True, raw=True, family=2)
                try:
                    plan = ops.Plan("bench")
                    for x1, x2, sa, sb, y, sq in sets:
                        kw = dict(x2=x2)
                        if v["pro"]:
                            kw.update(ssq_a=sa, ssq_b=sb, ssq_wb=0.5, pa=pa, pstride=0, act_in=ops.ACT_SILU)
                        if v["post"]:
                            kw.update(post=post)
                        elif Co == 32:
                            kw.update(ssq_out=sq)
                        if cfg is None:
                            kw.update(cfg=ops.pick_cfg(4, Co, H, H, B, 3, 3, 1, full_cout=True, family=0))
                        else:
                            kw.update(cfg=cfg)
                        ops.igemm(plan, x1, pw, y, label=cname, **kw)
                    plan.run()
                    torch.cuda.synchronize()
                except Exception as e:   # noqa: BLE001
                    lines.append(dict(H=H, Cin=Cin, Cout=Co, variant=vname, cand=cname, error=str(e)[:200]))
                    print(json.dumps(lines[-1]), flush=True)
                    continue
                out = sets[0][4].t.float().clone()
                if first is None:
                    first = out
                err = ((out - first).norm() / first.norm()).item()
                for _ in range(3):
                    plan.run()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(args.iters):
                    plan.run()
                e1.record()
                torch.cuda.synchronize()
                us = e0.elapsed_time(e1) * 1e3 / (4 * args.iters)
                nbytes = B * H * H * (Cin + Co) * 2
                lines.append(dict(H=H, Cin=Cin, Cout=Co, variant=vname, cand=cname, us=round(us, 2), gbs=round(nbytes / us / 1e3, 1), dist_to_first=float(f"{err:.3e}")))
                print(json.dumps(lines[-1]), flush=True)
    if args.out:
        with open(args.out, "a") as f:
            for ln in lines:
                f.write(json.dumps(ln) + "\n")


if __name__ == "__main__":
    main()
