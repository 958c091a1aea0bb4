#!/usr/bin/env python
"""Ablation probe of the LDS-staged conv kernel (csrc/conv_lds.hip) on the benchmark's 3x3 shapes.

    bash tools/build_probe_lib.sh
    IMAGEN_LIB_PATH=imagen-pytorch_amd/libimagen_hip_probe.so python tools/conv_probe.py [shape-substring,...]

Per shape and tile cfg: time of the full kernel and of ablated variants (ImagenIgemmParams.dbg bits, compiled in only with -DCL_PROBE:
1 no weight-DMA waits, 2 no MFMA, 4 no activation staging in the loop, 8 no stores, 16 no weight DMA, 32 no chunk barrier, 64 no
B-fragment reads, 128 no A-fragment reads, 256 no chunk rotation).  Ablated results are numerically meaningless.
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from imagen_pytorch_amd import ops

dev = torch.device("cuda:0")
SHAPES = [  # (name, B, H, W, C1, C2, Cout, K, [(cfg, th, tw), ...])
    ("384->256 @32", 16, 32, 32, 256, 128, 256, 3, [(16, 8, 16), (28, 8, 16), (17, 8, 8), (29, 8, 8), (18, 8, 8), (1, 8, 16)]),
    ("192->128 @64", 16, 64, 64, 128, 64, 128, 3, [(16, 8, 16), (28, 8, 16), (1, 16, 8)]),
    ("128->128 @32", 16, 32, 32, 128, 0, 128, 3, [(16, 8, 16), (17, 8, 8), (29, 8, 8), (3, 8, 8)]),
    ("64->64 @64", 16, 64, 64, 64, 0, 64, 3, [(20, 16, 8), (21, 8, 8), (2, 32, 8)]),
    ("32->32 @256", 16, 256, 256, 32, 0, 32, 3, [(22, 16, 16), (23, 8, 16), (0, 8, 32)]),
]
VARIANTS = [0, 256, 1, 17, 2, 4, 21, 21 + 64, 21 + 128, 21 + 192, 32, 8]
ONLY_FULL = os.environ.get("CONV_PROBE_ONLY_FULL") == "1"   # counter passes: one variant per kernel symbol
if ONLY_FULL:
    VARIANTS = [0]


def run(B, H, W, C1, C2, Cout, K, cfg, dbg, raw=False):
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
    if not raw:
        kw = dict(rs=torch.rand(B * H * W, device=dev) + 0.5, pa=torch.rand(B, pw.Cin_pad, device=dev) + 0.5,
                  ps=torch.rand(B, pw.Cin_pad, device=dev), pstride=pw.Cin_pad, act_in=ops.ACT_SILU)
    p = ops.igemm(plan, x1, pw, y, x2=x2, cfg=cfg, **kw)
    p.dbg = dbg
    for _ in range(3):
        plan.run()
    torch.cuda.synchronize()
    n = 20
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(n):
        plan.run()
    t1.record(); torch.cuda.synchronize()
    return t0.elapsed_time(t1) * 1e3 / n


if __name__ == "__main__":
    sel = sys.argv[1].split(",") if len(sys.argv) > 1 else None
    tab = ops.cfg_table()
    print("variants (dbg):", VARIANTS, "+ raw (no prologue)")
    for name, B, H, W, C1, C2, Cout, K, cfgs in SHAPES:
        if sel and not any(s in name for s in sel):
            continue
        gf = 2.0 * B * H * W * Cout * K * K * (C1 + C2) / 1e9
        for cfg in cfgs:
            fam = tab[cfg[0]][3]
            row = []
            for v in (VARIANTS if fam == 1 else [0]):
                try:
                    us = run(B, H, W, C1, C2, Cout, K, cfg, v)
                    row.append(f"{v}:{us:6.1f}")
                except Exception as e:
                    row.append(f"{v}:ERR")
            if not ONLY_FULL:
                us = run(B, H, W, C1, C2, Cout, K, cfg, 0, raw=True)
                row.append(f"raw:{us:6.1f}")
            print(f"{name:14s} {gf:5.1f}GF cfg{cfg} | " + " ".join(row), flush=True)
