#!/bin/bash
# Call Q: what request MERGING would buy over request LANES — the sequential step time of ONE sample() call at batch 8 / 16 / 24 / 48
# (tools/step_time.py; informational: the headline metric stays batch 8 per request).
set -u
cd "$(dirname "$0")/../.."
OUT=$PWD/gpurun_out/r06_q
mkdir -p $OUT
for b in 8 16 24 48; do
  timeout 600 python tools/step_time.py --batch $b --steps 40 --reps 2 --tag batch$b 2>>$OUT/err.txt | tail -n 1 | tee -a $OUT/batch_sweep.jsonl | cut -c1-200
done
tail -n 3 $OUT/err.txt
