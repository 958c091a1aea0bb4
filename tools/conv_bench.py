#!/usr/bin/env python
"""Kernel-level A/B of the prologue-free 3x3 conv families on the benchmark's C >= 128 layer shapes (rows 16 = batch 8 under CFG):
each candidate tile configuration is launched `--iters` times over four rotating input / output buffer sets (same weights), timed with
events on the launch stream, and its output compared with the first candidate's (the planner's current pick).

    python tools/conv_bench.py [--iters 40] [--shapes 128:128:64 ...] [--out file.jsonl]

Candidates: "pick" (ops.pick_cfg as the planner calls it), "dma:<tile px>x<tile couts>", "big:<n>" (n-th configuration of family 5),
"cfg:<id>:<th>:<tw>".  One JSON line per (shape, candidate): us per launch, TFLOP/s, normwise distance to the first candidate."""
import argparse
import json
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from imagen_pytorch_amd import ops  # noqa: E402

DEFAULT_SHAPES = ["128:128:64", "192:128:64", "256:256:32", "384:256:32", "128:128:32"]
DEFAULT_CANDS = {64: ["pick", "big:0", "big:3"], 32: ["pick", "big:1", "big:2"]}


def resolve(spec, Cout, H, B):
    tab = ops.cfg_table()
    if spec == "pick":
        return ops.pick_cfg(4, Cout, H, H, B, 3, 3, 1, full_cout=False, raw=True, family=2)
    kind, rest = spec.split(":", 1)
    if kind == "cfg":
        i, th, tw = map(int, rest.split(":"))
        return (i, th, tw)
    if kind == "big":
        i = [j for j, c in enumerate(tab) if c[3] == 5][int(rest)]
    else:
        tp, bn = map(int, rest.split("x"))
        i = next(j for j, c in enumerate(tab) if c[3] == 2 and (c[0], c[1]) == (tp, bn))
    sh = ops.launchable_shapes(i, H, H, 3, 3, 1)
    return (i, sh[0][2], sh[0][3])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=40)
    ap.add_argument("--rows", type=int, default=16)
    ap.add_argument("--shapes", nargs="*", default=DEFAULT_SHAPES)
    ap.add_argument("--cands", nargs="*", default=None)
    ap.add_argument("--gca", action="store_true", help="GlobalContext partials from the epilogue (Cout <= 128 shapes)")
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    B = args.rows
    g = torch.Generator().manual_seed(0)
    lines = []
    for shp in args.shapes:
        Cin, Cout, H = map(int, shp.split(":"))
        w = torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(9 * Cin)
        pw = ops.pack_weight(w, torch.randn(Cout, generator=g) * 0.1, dev, G=4)
        xs = [ops.act_from_nchw((torch.randn(B, Cin, H, H, generator=g) * 0.7).to(dev)) for _ in range(4)]
        ys = [ops.new_act(B, H, H, Cout, dev) for _ in range(4)]
        wk = torch.randn(Cout, generator=g).to(dev) * 0.3
        first = None
        for spec in (args.cands or DEFAULT_CANDS[H]):
            try:
                cfg = resolve(spec, Cout, H, B)
                plan = ops.Plan("bench")
                for i in range(4):
                    kw = dict(gca=dict(wk=wk, bk=0.1)) if (args.gca and Cout <= 128) else {}
                    ops.igemm(plan, xs[i], pw, ys[i], cfg=cfg, label=spec, **kw)
                plan.run()
                torch.cuda.synchronize()
            except Exception as e:   # noqa: BLE001
                lines.append(dict(shape=shp, cand=spec, error=str(e)[:200]))
                print(json.dumps(lines[-1]), flush=True)
                continue
            out = ys[0].t.float().clone()
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
            fl = 2.0 * B * H * H * Cout * 9 * Cin
            lines.append(dict(shape=shp, cand=spec, cfg=list(cfg), fam=ops.cfg_table()[cfg[0]][3], us=round(us, 2), tflops=round(fl / us / 1e6, 1),
                              dist_to_first=float(f"{err:.3e}"), gca=bool(args.gca and Cout <= 128)))
            print(json.dumps(lines[-1]), flush=True)
    if args.out:
        with open(args.out, "a") as f:
            for ln in lines:
                f.write(json.dumps(ln) + "\n")


if __name__ == "__main__":
    main()
