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

def run(name, B, H, W, C1, C2, Cout, K, variant, cfg=None):
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
    if variant != "noprologue":
        rs = torch.rand(B * H * W, device=dev) + 0.5
        pa = torch.rand(B, pw.Cin_pad, device=dev) + 0.5
        ps = torch.rand(B, pw.Cin_pad, device=dev)
        kw = dict(rs=rs, pa=pa, ps=ps, pstride=pw.Cin_pad, act_in=ops.ACT_SILU)
    p = ops.igemm(plan, x1, pw, y, x2=x2, cfg=cfg, **kw)
    p.dbg = int(variant[3:]) if variant.startswith("dbg") else 0   # bit mask, see ImagenIgemmParams.dbg
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

if __name__ == "__main__":
    args = sys.argv[1:]
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
