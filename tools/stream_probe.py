#!/usr/bin/env python
"""The streaming conv family (csrc/conv_stream.hip) against the kernels the planner used before it, on the 32-channel 3x3 layers of the
README cascade: raw input / ssq-statistics Block prologue, one / two inputs, plain + ssq_out | post_pa epilogue.

    python tools/stream_probe.py
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from imagen_pytorch_amd import ops

dev = torch.device("cuda:0")


def build(B, H, C2, pro, post, stream):
    torch.manual_seed(0)
    x1 = ops.new_act(B, H, H, 32, dev); x1.t.normal_()
    x2 = None
    if C2:
        x2 = ops.new_act(B, H, H, C2, dev); x2.t.normal_()
    C = 32 + C2
    pw = ops.pack_weight(torch.randn(32, C, 3, 3) / (C * 9) ** 0.5, torch.zeros(32), dev)
    y = ops.new_act(B, H, H, 32, dev)
    kw = {}
    if pro:
        kw = dict(ssq_a=torch.rand(B * H * H, device=dev) * 32 + 1, pa=torch.rand(1, pw.Cin_pad, device=dev) + 0.5, pstride=0, act_in=ops.ACT_SILU)
        if C2:
            kw.update(ssq_b=torch.rand(B * H * H, device=dev) * 32 + 1, ssq_wb=0.5)
    if post:
        kw.update(post=dict(pa=torch.rand(B, 32, device=dev), ps=torch.rand(B, 32, device=dev), pstride=32))
    else:
        kw.update(ssq_out=torch.zeros(B * H * H, device=dev))
    plan = ops.Plan()
    old = ops.CONV_STREAM
    ops.CONV_STREAM = 1 if stream else 0
    try:
        p = ops.igemm(plan, x1, pw, y, x2=x2, **kw)
        p.dbg = int(os.environ.get("STREAM_PROBE_DBG", "0")) if stream else 0   # probe library: 8 = no output stores
    finally:
        ops.CONV_STREAM = old
    return plan, p


def timed(plan, n=30):
    for _ in range(3):
        plan.run()
    torch.cuda.synchronize()
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(n):
        plan.run()
    t1.record(); torch.cuda.synchronize()
    return t0.elapsed_time(t1) * 1e3 / n


if __name__ == "__main__":
    tab = ops.cfg_table()
    for B, H in ((16, 256), (16, 128), (16, 64)):
        for C2, pro, post in ((0, False, False), (0, True, True), (0, True, False), (32, True, True), (32, True, False), (32, False, False)):
            by = 2.0 * B * H * H * (32 + C2 + 32)
            res = []
            for stream in (False, True):
                plan, p = build(B, H, C2, pro, post, stream)
                us = timed(plan)
                res.append(f"cfg{p.cfg}{tab[p.cfg]}: {us:6.1f} us {by / us / 1e3:6.0f} GB/s")
            print(f"{32 + C2}->32 @{H} {'pro' if pro else 'raw'}{' post' if post else ' ssq_out'}: " + "  |  ".join(res), flush=True)
