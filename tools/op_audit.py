"""Per-op contract audit (test infrastructure; imports tests/plan_interp.py and, for the whole-denoiser figure, the oracle — never imported by the product).

A whole-denoiser distance to the oracle says THAT the HIP path and the fp32 restatement differ, a level tap says roughly WHERE; neither
separates one kernel's internal rounding from the noise it inherits.  This tool executes a denoiser plan launch by launch twice — by the
kernel library (MI355X, or the CPU emulation with --emul) and by the CPU plan interpreter (every op restated from its contract in
include/imagen_hip.h: fp16 storage, fp32 arithmetic) — FROM IDENTICAL INPUTS: after each launch the interpreter-side copies of the buffers
that launch wrote are overwritten with the kernel's results.  Two fp32 computations of the same quantity round to the same fp16 value
almost everywhere, so a launch whose kernel follows its contract shows ~1e-5 .. 2e-4 here; one that rounds something its contract keeps in
fp32 shows its own error, alone, and `coherence` says whether that error is one vector repeated over the rows (a bias that survives
averaging) or noise.  (Found with the arithmetic this tool automates, round 5: the MFMA temporal attention rounded the fp32 null value to
fp16 — 5e-5 normwise on the op, invisible beside the output rounding, 4 % of the C5 denoiser's distance to the oracle.)

    python tools/op_audit.py --config c5 [--frames 16 --size 64] [--null] [--emul] [--top 30] [--json out.json]
    python tools/op_audit.py --config u1 --size 64 --null          (u2: README unet2 with its low-res conditioning, --size 256)
"""
from __future__ import annotations

import argparse
import contextlib
import ctypes
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

README_U1 = dict(dim=32, cond_dim=512, dim_mults=(1, 2, 4, 8), num_resnet_blocks=3, layer_attns=(False, True, True, True),
                 layer_cross_attns=(False, True, True, True))
README_U2 = dict(dim=32, cond_dim=512, dim_mults=(1, 2, 4, 8), num_resnet_blocks=(2, 4, 8, 8), layer_attns=(False, False, False, True),
                 layer_cross_attns=(False, False, False, True), lowres_cond=True)


def emulate_cuda():
    """The conftest.py shim of IMAGEN_EMUL_TESTS=1: engines on the CPU, launches through the emulated kernel library (IMAGEN_LIB_PATH)."""
    from imagen_pytorch_amd import ops, unet as unet_mod

    class Stream:
        device = torch.device("cpu")
        cuda_stream = 0

        def __init__(self, *a, **k):
            pass

        def synchronize(self):
            pass

        def wait_stream(self, other):
            pass

    torch.cuda.synchronize = lambda *a, **k: None
    torch.cuda.Stream = Stream
    torch.cuda.current_stream = lambda *a, **k: Stream()
    torch.cuda.device = lambda *a, **k: contextlib.nullcontext()
    torch.cuda.stream = lambda *a, **k: contextlib.nullcontext()
    ops.current_stream_handle = lambda: 0
    unet_mod._ENGINE_DEVICE_TYPES = ("cuda", "cpu")


def storage_of(mem, addr):
    for base, st in mem.storages.items():
        if base <= addr < base + st.nbytes():
            return base, st
    return None, None


# GlobalContext partials are (reference logit, sum of exponentials, weighted sums) per tile: a kernel may pick another reference logit than
# the interpreter's and still describe the same softmax pooling — compared where they are merged (GCA_FINAL / GCA_TAIL), not element by element
OPAQUE_FIELDS = ("gca_part", "part")


def pointer_fields(p, opaque=False):
    return [getattr(p, name) for name, ct in p._fields_ if ct is ctypes.c_void_p and getattr(p, name) and (name in OPAQUE_FIELDS) == opaque]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", choices=("c5", "u1", "u2"), default="c5", help="BASELINE C5's Unet3D(dim 64) | README unet1 | README unet2 (lowres_cond)")
    ap.add_argument("--frames", type=int, default=16)
    ap.add_argument("--size", type=int, default=64)
    ap.add_argument("--null", action="store_true", help="the unconditional row (cond_drop_prob = 1) instead of the conditional one")
    ap.add_argument("--emul", action="store_true", help="kernels on the CPU emulation (IMAGEN_LIB_PATH = the emulated library)")
    ap.add_argument("--top", type=int, default=30)
    ap.add_argument("--json", default=None)
    ap.add_argument("--threads", type=int, default=min(os.cpu_count() or 1, 32))
    args = ap.parse_args()
    torch.set_num_threads(args.threads)

    from imagen_pytorch_amd import _abi, ops
    from plan_interp import Interpreter
    if args.emul:
        assert "emul" in os.path.basename(os.environ.get("IMAGEN_LIB_PATH", "")), "--emul needs IMAGEN_LIB_PATH=<emulated library>"
        emulate_cuda()
        dev = torch.device("cpu")
    else:
        dev = torch.device("cuda", 0)
        torch.cuda.set_device(dev)
    lib = _abi.load_library()
    inv = {v: k[len("IMAGEN_OP_"):] for k, v in _abi.ENUMS.items() if k.startswith("IMAGEN_OP_")}

    torch.manual_seed(0)
    keep = torch.tensor([not args.null])
    te = mask = None
    if args.config == "c5":
        from imagen_pytorch_amd import Unet3D
        from imagen_pytorch_amd.engine3d import UnetEngine3D
        from test_video_gpu import _derandomise_unet3d as derandomise_unet3d   # (the C5 test's own weights: dirac + dense temporal convs)
        kw = dict(dim=64, dim_mults=(1, 2, 4, 8))
        u = Unet3D(**kw).eval()
        derandomise_unet3d(u)
        x, t = torch.randn(1, 3, args.frames, args.size, args.size), torch.tensor([0.3])
        te = torch.randn(1, 24, 768)
        mask = torch.ones(1, 24, dtype=torch.bool)
        mask[0, 19:] = False
        make = lambda d, dry: UnetEngine3D(u, 1, 1, args.frames, args.size, d, dry=dry)
        x_in = x.permute(0, 2, 1, 3, 4)
    else:
        from imagen_pytorch_amd import Unet
        from imagen_pytorch_amd.engine import UnetEngine
        kw = README_U1 if args.config == "u1" else README_U2
        u = Unet(**kw).eval()
        torch.nn.init.normal_(u.final_conv.weight, std=0.05)
        torch.nn.init.normal_(u.final_conv.bias, std=0.05)
        x, t = torch.randn(1, 3, args.size, args.size), torch.tensor([0.3])
        te = torch.randn(1, 24, 768)
        mask = torch.ones(1, 24, dtype=torch.bool)
        mask[0, 18:] = False
        make = lambda d, dry: UnetEngine(u, 1, 1, args.size, d, dry=dry)
        x_in = x

    lowres = bool(kw.get("lowres_cond"))
    lnt = torch.full((1,), 0.5) if lowres else None
    low = torch.randn(1, 3, args.size, args.size) if lowres else None
    ops.KEEP_REFERENCE_WEIGHTS = True
    eng_i = make("cpu", True)
    eng_i.set_conditioning(text_embeds=te, text_mask=mask, keep=keep, lowres_noise_times=lnt)
    if not args.emul:
        ops.KEEP_REFERENCE_WEIGHTS = False
    eng_k = make(dev, args.emul)          # (under emulation both engines are dry: the launches below go through imagen_launch by hand)
    if args.emul:
        eng_k.set_conditioning(text_embeds=te, text_mask=mask, keep=keep, lowres_noise_times=lnt)
    else:
        eng_k.dry = True                  # set_conditioning must not run the static plan itself: it is audited launch by launch below
        eng_k.set_conditioning(text_embeds=te.to(dev), text_mask=mask.to(dev), keep=keep, lowres_noise_times=lnt)
    n_tok = te.shape[1]
    plans = [("static", eng_k._static_plans[n_tok][0], eng_i._static_plans[n_tok][0]), ("step", eng_k.step_plan, eng_i.step_plan)]

    it = Interpreter()
    twin = {}     # interpreter-side storage base address -> (kernel-side tensor whose storage is its twin, dtype)

    unpaired = [0]

    def pair(tk, ti):
        if isinstance(tk, torch.Tensor) and isinstance(ti, torch.Tensor) and ti.numel() > 0:
            it.mem.register(ti)
            if tk.shape != ti.shape or tk.dtype != ti.dtype or tk.untyped_storage().nbytes() != ti.untyped_storage().nbytes():
                unpaired[0] += 1      # (not expected: both engines run the same planner; such a buffer is simply not audited)
                return
            twin.setdefault(ti.untyped_storage().data_ptr(), (tk, ti))
    for _, pk, pi in plans:
        assert len(pk.ops) == len(pi.ops) and len(pk.keep) == len(pi.keep), "the two engines planned different launch lists"
        for a, b in zip(pk.keep, pi.keep):
            pair(a, b)
    for name in ("x_in", "lowres_in", "times", "lowres_times", "out", "keep_u8", "src_idx", "arange_idx"):
        pair(getattr(eng_k, name, None), getattr(eng_i, name, None))
    pair(eng_k.t_const.t, eng_i.t_const.t)

    def whole(tens):
        """The tensor's whole storage as a flat tensor of its dtype."""
        st = tens.untyped_storage()
        return torch.empty(0, dtype=tens.dtype, device=tens.device).set_(st, 0, (st.nbytes() // tens.element_size(),), (1,))

    eng_i.x_in.copy_(x_in)
    eng_i.times.copy_(t)
    eng_k.x_in.copy_(x_in.to(dev))
    eng_k.times.copy_(t.to(dev))
    if lowres:
        eng_i.lowres_in.copy_(low)
        eng_k.lowres_in.copy_(low.to(dev))
    st_job = _abi.STRUCTS["ImagenKvPrepParams"]
    K_MULTI = _abi.ENUMS["IMAGEN_OP_KV_PREP_MULTI"]
    rows = []
    t0 = time.time()
    h = ops.current_stream_handle()
    for pname, pk, pi in plans:
        for idx, ((kind, sk, label), (_, si, _)) in enumerate(zip(pk.ops, pi.ops)):
            ptrs = pointer_fields(si)
            if kind == K_MULTI:      # the jobs of a KV_PREP_MULTI launch live in a buffer: their pointers count
                for j in range(si.n):
                    ptrs += pointer_fields(st_job.from_address(si.jobs + j * ctypes.sizeof(st_job)))
            touched = {}
            for a in ptrs + pointer_fields(si, opaque=True):
                base, _ = storage_of(it.mem, a)
                if base is not None and base in twin:
                    touched[base] = whole(twin[base][1]).clone()
            opaque = {storage_of(it.mem, a)[0] for a in pointer_fields(si, opaque=True)}
            _abi.check(lib.imagen_launch(kind, ctypes.addressof(sk), ctypes.sizeof(sk), h), f"{pname}[{idx}] {label}")
            if not args.emul:
                torch.cuda.synchronize()
            it.DISPATCH[kind](it, si)
            for base, before in touched.items():
                tk, ti = twin[base]
                after = whole(ti)
                bits = {2: torch.int16, 4: torch.int32, 1: torch.uint8, 8: torch.int64}[before.element_size()]
                changed = after.view(bits) != before.view(bits)      # bit patterns: uninitialised NaN garbage nobody wrote is not a change
                n = int(changed.sum())
                if n == 0:
                    continue
                got = whole(tk).to("cpu")
                if base in opaque:
                    whole(ti).copy_(got)
                    continue
                if before.dtype.is_floating_point:
                    a, b = got.float()[changed], after.float()[changed]
                    d = a - b
                    err = float(d.norm() / b.norm().clamp(min=1e-30))
                    # coherence: the share of the error that is ONE vector repeated over the rows of the tensor (last dim = channels)
                    C = ti.shape[-1] if ti.ndim >= 2 else 1
                    coh = None
                    full = (got.float() - after.float()) * changed
                    if C > 1 and full.numel() % C == 0 and full.numel() // C >= 16:
                        m = full.reshape(-1, C)
                        live = changed.reshape(-1, C).any(dim=1)
                        if int(live.sum()) >= 16:
                            mm = m[live]
                            coh = float(mm.mean(dim=0).norm() * mm.shape[0] ** 0.5 / mm.norm().clamp(min=1e-30))
                    rows.append(dict(plan=pname, idx=idx, kind=inv.get(kind, str(kind)), label=label, dtype=str(before.dtype).replace("torch.", ""),
                                     elems=n, err=err, max_abs=float(d.abs().max()), coherence=coh, nan=bool(a.isnan().any())))
                else:
                    rows.append(dict(plan=pname, idx=idx, kind=inv.get(kind, str(kind)), label=label, dtype=str(before.dtype).replace("torch.", ""),
                                     elems=n, err=float((got[changed] != after[changed]).float().mean()), max_abs=0.0, coherence=None, nan=False))
                whole(ti).copy_(got)           # the next launch starts from the kernel's results on both sides
    dt = time.time() - t0
    out_k = eng_k.out.to("cpu").float()
    print(f"# {args.config} {'null' if args.null else 'cond'} row, {len(rows)} written buffers over {sum(len(p[1].ops) for p in plans)} launches, {dt:.0f} s "
          f"({'emulated kernels' if args.emul else torch.cuda.get_device_name(0)})")
    fl = [r for r in rows if r["dtype"].startswith("float")]
    if unpaired[0]:
        print(f"# {unpaired[0]} buffers of the two engines did not pair up (not audited)")
    print(f"# median per-launch distance kernel vs contract (identical inputs): {sorted(r['err'] for r in fl)[len(fl) // 2]:.2e}")
    print("# largest distances:")
    for r in sorted(fl, key=lambda r: -r["err"])[: args.top]:
        coh = "   -  " if r["coherence"] is None else f"{r['coherence']:6.2f}"
        print(f"  {r['plan']:6s}{r['idx']:4d} {r['kind']:18s} {r['label'][:44]:44s} {r['dtype']:8s} n={r['elems']:9d} err={r['err']:.2e} max={r['max_abs']:.2e} coherence={coh}")
    print("# most coherent errors (coherence x err: a bias of the whole map, what averaging over pixels does not remove):")
    for r in sorted((r for r in fl if r["coherence"] is not None), key=lambda r: -r["coherence"] * r["err"])[: args.top]:
        print(f"  {r['plan']:6s}{r['idx']:4d} {r['kind']:18s} {r['label'][:44]:44s} {r['dtype']:8s} n={r['elems']:9d} err={r['err']:.2e} coherence={r['coherence']:6.2f}")
    if args.json:
        json.dump(dict(config=args.config, null=args.null, frames=args.frames, size=args.size, emulated=args.emul, seconds=dt, rows=rows),
                  open(args.json, "w"))
    assert torch.isfinite(out_k).all()


if __name__ == "__main__":
    main()
