#!/usr/bin/env python
"""Fast A/B timer of the C3 sampling loop: ms per DDPM step of each cascade stage (graph replay, one request at a time).

    python tools/step_time.py [--steps 60] [--batch 8] [--reps 3] [--tag name]

Builds the benchmark's Imagen (bench.build_imagen), warms both stages (weight packing + graph capture, 2 steps), then times `reps`
sample() calls of `steps` DDPM steps per stage with torch.cuda events around each stage's loop (IMAGEN_TIMING prints them too).  A whole
call takes ~25 s instead of bench.py's minutes: the tool every kernel A/B of a GPU call goes through; the headline number stays
bench.py's.  Prints one JSON line {tag, u1_ms, u2_ms, pair_ms, lib, knobs}."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--tag", default="")
    ap.add_argument("--lanes", type=int, default=0, help="also time N whole cascades side by side (one thread + stream each), `steps` DDPM steps per stage")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    imagen = bench.build_imagen(1000, dev)
    te = torch.randn(args.batch, 256, 768, generator=torch.Generator().manual_seed(1234)).to(dev)
    imagen.sample(text_embeds=te, cond_scale=3.0, use_tqdm=False, seed=1, max_steps=2)
    torch.cuda.synchronize()
    # per-stage times: sample one stage at a time through stop_at / start_at
    best = [1e9, 1e9]
    for r in range(args.reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        low = imagen.sample(text_embeds=te, cond_scale=3.0, use_tqdm=False, seed=2 + r, max_steps=args.steps, stop_at_unet_number=1)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        imagen.sample(text_embeds=te, cond_scale=3.0, use_tqdm=False, seed=2 + r, max_steps=args.steps, start_at_unet_number=2,
                      start_image_or_video=low)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        best[0] = min(best[0], (t1 - t0) / args.steps * 1e3)
        best[1] = min(best[1], (t2 - t1) / args.steps * 1e3)
    lanes_ms = None
    if args.lanes > 0:
        import threading

        def run(lane, reps):
            with imagen.lane(lane), torch.cuda.device(dev):
                for r in range(reps):
                    imagen.sample(text_embeds=te, cond_scale=3.0, use_tqdm=False, seed=50 + 10 * lane + r, max_steps=args.steps)

        for reps in (1, 2):     # first round: every lane builds its stages / graphs
            th = [threading.Thread(target=run, args=(1 + l, reps)) for l in range(args.lanes)]
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            [t.start() for t in th]
            [t.join() for t in th]
            torch.cuda.synchronize()
            lanes_ms = (time.perf_counter() - t0) / (reps * args.lanes * args.steps) * 1e3   # ms per DDPM step pair, aggregate
    from imagen_pytorch_amd import _abi
    knobs = {k: v for k, v in os.environ.items() if k.startswith("IMAGEN_") and k != "IMAGEN_LIB_PATH"}
    print(json.dumps({"tag": args.tag, "u1_ms": round(best[0], 4), "u2_ms": round(best[1], 4), "pair_ms": round(best[0] + best[1], 4),
                      "images_per_s_sequential_est": round(args.batch / (best[0] + best[1]), 4),
                      "lanes": args.lanes, "lanes_pair_ms": None if lanes_ms is None else round(lanes_ms, 4),
                      "images_per_s_lanes_est": None if lanes_ms is None else round(args.batch / lanes_ms, 4),
                      "lib": os.path.basename(_abi.LIB_PATH), "knobs": knobs, "steps": args.steps}), flush=True)


if __name__ == "__main__":
    main()
