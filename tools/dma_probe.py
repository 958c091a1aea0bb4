#!/usr/bin/env python
"""Time attribution of the all-DMA conv kernel (csrc/conv_dma.hip) by COMPILE-TIME ablation variants of cfg 0 (128 px x 128 co, ring 3,
prefetch 1), each its own kernel instantiation in the probe library:

    bash tools/build_probe_lib.sh
    IMAGEN_LIB_PATH=imagen-pytorch_amd/libimagen_hip_probe.so python tools/dma_probe.py
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from imagen_pytorch_amd import ops

dev = torch.device("cuda:0")
VARIANTS = ["full", "no weight DMA", "no act DMA", "no DMA", "no B reads", "no A reads", "no frag reads", "MFMA + loop only", "no MFMA", "no chunk barrier",
            "no vmcnt waits"]
SHAPES = [("384->256 @32", 16, 32, 32, 384, 256), ("256->256 @32", 16, 32, 32, 256, 256), ("192->128 @64", 16, 64, 64, 192, 128), ("128->128 @64", 16, 64, 64, 128, 128)]


def run(B, H, W, C, Cout, cfg):
    torch.manual_seed(0)
    x = ops.new_act(B, H, W, C, dev); x.t.normal_()
    pw = ops.pack_weight(torch.randn(Cout, C, 3, 3) / (C * 9) ** 0.5, torch.zeros(Cout), dev)
    y = ops.new_act(B, H, W, Cout, dev)
    plan = ops.Plan()
    ops.igemm(plan, x, pw, y, cfg=(cfg, 8, 16))
    for _ in range(3):
        plan.run()
    torch.cuda.synchronize()
    n = 30
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(n):
        plan.run()
    t1.record(); torch.cuda.synchronize()
    return t0.elapsed_time(t1) * 1e3 / n


if __name__ == "__main__":
    tab = ops.cfg_table()
    base = next(i for i, c in enumerate(tab) if c[3] == 2)
    ids = [base] + [base + 20 + i for i in range(10)]
    assert len(tab) >= base + 30, "probe library needed (IMAGEN_LIB_PATH=.../libimagen_hip_probe.so)"
    for name, B, H, W, C, Cout in SHAPES:
        gf = 2.0 * B * H * W * Cout * 9 * C / 1e9
        print(f"{name} ({gf:.1f} GF; MFMA floor {gf / 2.5:.1f} us): " + " | ".join(f"{v}: {run(B, H, W, C, Cout, i):.1f}" for v, i in zip(VARIANTS, ids)), flush=True)
