#!/usr/bin/env python
"""Kernel-level A/B of the attention tilings on the benchmark's sites (rows 16 = batch 8 under CFG): the online softmax
(softmax_mode 0) against the bounded-logit softmax (softmax_mode 1), both through the planner's own call (QNORM fused into the q load,
KV_PREP outside the timed loop).  Times with events on the launch stream over four rotating q / o buffer sets.

    python tools/attn_bench.py [--iters 30] [--out file.jsonl]

One JSON line per (site, mode): us per launch, TFLOP/s (4 * rows * J * D per image and head group), fraction of the 2.5 PFLOP/s dense fp16
MFMA peak, normwise distance of the two modes' outputs."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from imagen_pytorch_amd import ops  # noqa: E402

# (name, B, heads, rows, J, shared k/v): the self-attention sites of unet2 / unet1 (1024 / 256 / 64 tokens + 2 context + null) and the
# cross-attention sites (per-head k/v, 39 + 2 text / time tokens)
SITES = [("self-1024", 16, 1, 8 * 1024, 1027, True), ("self-256", 16, 1, 8 * 256, 259, True), ("self-64", 16, 1, 8 * 64, 67, True),
         ("cross-1024", 16, 8, 1024, 41, False), ("cross-4096", 16, 8, 4096, 41, False)]
PEAK = 2.5e15


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=30)
    ap.add_argument("--sites", nargs="*", default=None)
    ap.add_argument("--modes", nargs="*", type=int, default=[0, 1])
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    D = 64
    g = torch.Generator().manual_seed(0)
    lines = []
    for name, B, heads, rows, J, shared in SITES:
        if args.sites and name not in args.sites:
            continue
        Jp = (J + 31) // 32 * 32
        qs, ks = torch.ones(D), torch.ones(D)
        k = torch.randn(B, J, heads, D, generator=g).half().to(dev)
        v = torch.randn(B, J, heads, D, generator=g).half().to(dev)
        khat = torch.zeros(B, heads, Jp, D, dtype=torch.float16, device=dev)
        vt = torch.zeros(B, heads, D, Jp, dtype=torch.float16, device=dev)
        prep = ops.Plan("prep")
        ops.kv_prep(prep, k, v, ks.to(dev), khat, vt, B=B, heads=heads, rows=J, r0=0, src_strides=(J * heads * D, heads * D, D),
                    k_strides=(heads * Jp * D, Jp * D, D), vt_strides=(heads * D * Jp, D * Jp, Jp), head_dim=D)
        prep.run()
        sets = [(torch.randn(B, rows, heads, D, generator=g).half().to(dev), torch.empty(B, rows, heads, D, dtype=torch.float16, device=dev)) for _ in range(4)]
        bound = ops.attention_logit_bound(qs, ks, 8 * ops.LOG2E)
        first = None
        for mode in args.modes:
            plan = ops.Plan("bench")
            for q, o in sets:
                p = ops.attention(plan, q, khat, vt, o, B=B, heads=heads, rows=rows, J=J, head_dim=D, q_strides=(rows * heads * D, D, heads * D),
                                  k_strides=(heads * Jp * D, Jp * D, D), vt_strides=(heads * D * Jp, D * Jp, Jp), o_strides=(rows * heads * D, D, heads * D),
                                  q_scale=qs.to(dev), q_mult=8 * ops.LOG2E, logit_bound=bound if mode else None)
                assert p.softmax_mode == mode
            plan.run()
            torch.cuda.synchronize()
            out = sets[0][1].float().clone()
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
            flops = 4.0 * B * heads * rows * J * D
            lines.append(dict(site=name, B=B, heads=heads, rows=rows, J=J, softmax_mode=mode, us=round(us, 2), tflops=round(flops / us / 1e6, 1),
                              frac_of_mfma_peak=round(flops / us / 1e-6 / PEAK, 3), dist_to_first=float(f"{err:.3e}")))
            print(json.dumps(lines[-1]), flush=True)
    if args.out:
        with open(args.out, "a") as f:
            for ln in lines:
                f.write(json.dumps(ln) + "\n")


if __name__ == "__main__":
    main()
