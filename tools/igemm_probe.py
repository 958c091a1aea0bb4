#!/usr/bin/env python
"""Micro-benchmark / ablation of the igemm kernel on the conv shapes of the README cascade (MI355X).

    python tools/igemm_probe.py            # prints TFLOP/s per shape and ablation variant

Variants: full | no-prologue (raw copy staging) | dbg1 (stage only chunk 0) | dbg2 (no MFMA) — the ablation
switches only exist to attribute time (cdna_hip_programming.md §5 'ablate before optimizing'); results of the
dbg variants are numerically meaningless.
"""
import os, sys, time, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from imagen_pytorch_amd import ops, _abi

dev = torch.device("cuda:0")
SHAPES = [  # (name, B, H, W, C1, C2, Cout, K)
    ("u2.L3 384->256 3x3 @32", 16, 32, 32, 256, 128, 256, 3),
    ("u2.L3 256->256 3x3 @32", 16, 32, 32, 256, 0, 256, 3),
    ("u2.L3 128->128 3x3 @32", 16, 32, 32, 128, 0, 128, 3),
    ("u2.L2 192->128 3x3 @64", 16, 64, 64, 128, 64, 128, 3),
    ("u2.L2 64->64 3x3 @64", 16, 64, 64, 64, 0, 64, 3),
    ("u2.L1 96->64 3x3 @128", 16, 128, 128, 64, 32, 64, 3),
    ("u2.L0 32->32 3x3 @256", 16, 256, 256, 32, 0, 32, 3),
    ("u2.L0 64->32 3x3 @256", 16, 256, 256, 32, 32, 32, 3),
    ("u1.L3 256->256 3x3 @8", 16, 8, 8, 256, 0, 256, 3),
    ("u1.L2 128->128 3x3 @16", 16, 16, 16, 128, 0, 128, 3),
    ("u1.L1 64->64 3x3 @32", 16, 32, 32, 64, 0, 64, 3),
    ("lin 256->1024 M=16384", 16, 1, 1024, 256, 0, 1024, 1),
    ("res 192->128 1x1 @64", 16, 64, 64, 128, 64, 128, 1),
    ("res 96->64 1x1 @128", 16, 128, 128, 64, 32, 64, 1),
    ("res 64->32 1x1 @256", 16, 256, 256, 32, 32, 32, 1),
]

EXTRA_SHAPES = [  # the small-map layers of unet1 and the token GEMMs (few workgroups: tile choice decides how many CUs work)
    ("u1.L3 384->256 3x3 @8", 16, 8, 8, 256, 128, 256, 3),
    ("u1.L3 128->128 3x3 @8", 16, 8, 8, 128, 0, 128, 3),
    ("u1.L2 192->128 3x3 @16", 16, 16, 16, 128, 64, 128, 3),
    ("u1.L2 64->64 3x3 @16", 16, 16, 16, 64, 0, 64, 3),
    ("u1.L0 32->32 3x3 @64", 16, 64, 64, 32, 0, 32, 3),
    ("u2.L1 64->64 3x3 @128", 16, 128, 128, 64, 0, 64, 3),
    ("u2.L1 32->32 3x3 @128", 16, 128, 128, 32, 0, 32, 3),
    ("u2.L2 128->128 3x3 @64", 16, 64, 64, 128, 0, 128, 3),
    ("qkv 64->640 M=16384", 16, 1, 1024, 64, 0, 640, 1),
    ("to_q 256->512 M=1024", 16, 1, 64, 256, 0, 512, 1),
    ("ups 64->128 1x1 @128", 16, 128, 128, 64, 0, 128, 1),
    ("res 384->256 1x1 @32", 16, 32, 32, 256, 128, 256, 1),
    # BASELINE config C2 (base unet at dim 128, 64^2, rows 16)
    ("c2.L3 1024->1024 3x3 @8", 16, 8, 8, 1024, 0, 1024, 3),
    ("c2.L3 1536->1024 3x3 @8", 16, 8, 8, 1024, 512, 1024, 3),
    ("c2.L2 512->512 3x3 @16", 16, 16, 16, 512, 0, 512, 3),
    ("c2.L1 256->256 3x3 @32", 16, 32, 32, 256, 0, 256, 3),
    ("c2.L0 128->128 3x3 @64", 16, 64, 64, 128, 0, 128, 3),
    ("c2.L0 256->128 3x3 @64", 16, 64, 64, 128, 128, 128, 3),
]


def run(name, B, H, W, C1, C2, Cout, K, variant, cfg=None, G=None):
    torch.manual_seed(0)
    x1 = ops.new_act(B, H, W, C1, dev); x1.t.normal_()
    x2 = None
    if C2:
        x2 = ops.new_act(B, H, W, C2, dev); x2.t.normal_()
    C = C1 + C2
    pw = ops.pack_weight(torch.randn(Cout, C, K, K) / (C * K * K) ** 0.5, torch.zeros(Cout), dev, G=G)
    y = ops.new_act(B, H, W, Cout, dev)
    plan = ops.Plan()
    kw = {}
    if variant not in ("noprologue", "addend") and not variant.startswith("raw"):
        rs = torch.rand(B * H * W, device=dev) + 0.5
        pa = torch.rand(B, pw.Cin_pad, device=dev) + 0.5
        ps = torch.rand(B, pw.Cin_pad, device=dev)
        kw = dict(rs=rs, pa=pa, ps=ps, pstride=pw.Cin_pad, act_in=ops.ACT_SILU)
    if variant == "addend":    # res_conv form: out = conv(x) + gate[b, c] * addend (no prologue)
        kw = dict(addend=ops.new_act(B, H, W, Cout, dev, zero=True), gate=torch.rand(B, Cout, device=dev))
    p = ops.igemm(plan, x1, pw, y, x2=x2, cfg=cfg, **kw)
    p.dbg = int(variant[3:]) if variant.startswith(("dbg", "raw")) and variant[3:] else 0   # bit mask, see ImagenIgemmParams.dbg ("rawN": no prologue + dbg N)
    for _ in range(3):
        plan.run()
    torch.cuda.synchronize()
    n = 20
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(n):
        plan.run()
    t1.record(); torch.cuda.synchronize()
    us = t0.elapsed_time(t1) * 1e3 / n
    fl = 2.0 * B * H * W * Cout * K * K * C
    byts = (B * H * W * (C + Cout)) * 2
    return us, fl / us / 1e6, byts / us / 1e3, (p.cfg, p.TH, p.TW)

def sweep(shapes, G_list=None, variant="full"):
    """Every (cfg, tile shape) the launcher accepts, per shape and k-chunk depth G: the table pick_cfg() should reproduce."""
    import math
    tab = ops.cfg_table()
    for shp in shapes:
        name, B, H, W, C1, C2, Cout, K = shp
        C = C1 + C2
        for G in (G_list or sorted({ops.choose_G(C, K * K), 4})):
            if C % (8 * G) and G != ops.choose_G(C, K * K):
                continue
            res = []
            for i, (tp, bn, g, fam) in enumerate(tab):
                if g != G:
                    continue
                for th, tw in ops._tile_shapes(tp, H, W):
                    if ops.load_library().imagen_igemm_lds_bytes(i, K, K, 1, th, tw) <= 0:
                        continue
                    if bn > 32 and bn >= 2 * max(32, Cout):
                        continue
                    try:
                        us, tf, gbs, _ = run(name, B, H, W, C1, C2, Cout, K, variant, cfg=(i, th, tw), G=G)
                    except Exception as e:  # launcher refused the combination
                        continue
                    res.append((us, i, th, tw, tp, bn))
            res.sort()
            pk = ops.pick_cfg(G, Cout, H, W, B, K, K, 1, raw=(variant == "noprologue" and C2 == 0 and C % 32 == 0))
            best = " ".join(f"cfg{i}({tp}x{bn}) t{th}x{tw}:{us:.1f}" for us, i, th, tw, tp, bn in res[:6])
            best0 = [r for r in res if tab[r[1]][3] == 0][:1]
            best += "".join(f" | best family-0 cfg{i}({tp}x{bn}) t{th}x{tw}:{us:.1f}" for us, i, th, tw, tp, bn in best0)
            mine = [r for r in res if (r[1], r[2], r[3]) == tuple(pk)]
            print(f"{name:26s} G={G:2d} | pick cfg{pk[0]} t{pk[1]}x{pk[2]}: {mine[0][0] if mine else float('nan'):.1f}us | best: {best}", flush=True)



def timeline(shapes):
    """s_memtime stamps of workgroup 0's first consumer / producer wave (cycles relative to the first stamp).  Needs the trace
    build: `bash tools/build_trace_lib.sh; IMAGEN_LIB_PATH=imagen-pytorch_amd/libimagen_hip_trace.so python tools/igemm_probe.py --timeline`"""
    for shp in shapes:
        name, B, H, W, C1, C2, Cout, K = shp
        for variant in ("full", "noprologue"):
            torch.manual_seed(0)
            x1 = ops.new_act(B, H, W, C1, dev); x1.t.normal_()
            x2 = None
            if C2:
                x2 = ops.new_act(B, H, W, C2, dev); x2.t.normal_()
            C = C1 + C2
            pw = ops.pack_weight(torch.randn(Cout, C, K, K) / (C * K * K) ** 0.5, torch.zeros(Cout), dev)
            y = ops.new_act(B, H, W, Cout, dev)
            plan = ops.Plan()
            kw = {}
            if variant == "full":
                kw = dict(rs=torch.rand(B * H * W, device=dev) + 0.5, pa=torch.rand(B, pw.Cin_pad, device=dev) + 0.5,
                          ps=torch.rand(B, pw.Cin_pad, device=dev), pstride=pw.Cin_pad, act_in=ops.ACT_SILU)
            p = ops.igemm(plan, x1, pw, y, x2=x2, **kw)
            trace = torch.zeros(128, dtype=torch.int64, device=dev)
            p.gate = trace.data_ptr()
            for _ in range(3):
                plan.run()
            torch.cuda.synchronize()
            p.dbg = 128
            trace.zero_()
            plan.run()
            torch.cuda.synchronize()
            t = trace.cpu().tolist()
            t0 = min(v for v in t if v)
            cons = [v - t0 for v in t[:64] if v]
            prod = [v - t0 for v in t[64:] if v]
            print(f"{name} [{variant}] cfg{(p.cfg, p.TH, p.TW)}")
            print("  consumer (per phase: start, compute done, barrier passed; per tile: + epilogue done):", cons[:44])
            print("  producer (per phase: top, loads issued, set written, before barrier):", prod[:44], flush=True)


if __name__ == "__main__":
    args = sys.argv[1:]
    if args and args[0] == "--timeline":
        sel = args[1].split(",") if len(args) > 1 else None
        timeline([s_ for s_ in SHAPES + EXTRA_SHAPES if not sel or any(o in s_[0] for o in sel)])
        sys.exit(0)
    if args and args[0] in ("--sweep", "--sweep-raw"):
        sel = args[1].split(",") if len(args) > 1 else None
        raw = args[0] == "--sweep-raw"   # prologue-free single-input variant of every shape (the concat becomes one tensor): all three families
        shp = [s_ for s_ in SHAPES + EXTRA_SHAPES if not sel or any(o in s_[0] for o in sel)]
        if raw:
            shp = [(n, B, H, W, C1 + C2, 0, Co, K) for n, B, H, W, C1, C2, Co, K in shp if K == 3]
        globals()["sweep"](shp, variant="noprologue" if raw else "full")
        sys.exit(0)
    only = None
    if args and args[0].startswith("--only="):
        only = args[0][7:].split(",")
        args = args[1:]
    variants = args or ["full", "noprologue", "dbg1", "dbg2"]
    for shp in SHAPES:
        if only and not any(o in shp[0] for o in only):
            continue
        row = []
        for v in variants:
            us, tf, gbs, cfg = run(*shp, v)
            row.append(f"{v}: {us:7.1f}us {tf:7.1f}TF {gbs:6.0f}GB/s")
        print(f"{shp[0]:26s} cfg{cfg} | " + " | ".join(row), flush=True)
