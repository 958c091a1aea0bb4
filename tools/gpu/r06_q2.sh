#!/bin/bash
# Call Q, second part: 48 images in flight as 6 lanes x 8 (the bench's schedule), 3 x 16, 2 x 24, 1 x 48 (merged requests) on one box.
set -u
cd "$(dirname "$0")/../.."
OUT=$PWD/gpurun_out/r06_q
mkdir -p $OUT
for cfg in "8 6" "16 3" "24 2" "48 1" "24 3" "48 2"; do
  set -- $cfg
  timeout 900 python tools/step_time.py --batch $1 --lanes $2 --steps 40 --reps 1 --tag "batch$1 x lanes$2" 2>>$OUT/err2.txt | tail -n 1 | tee -a $OUT/batch_lanes_sweep.jsonl | python -c "import sys,json; r=json.loads(sys.stdin.read()); print(r['tag'], 'sequential', r['images_per_s_sequential_est'], 'lanes', r['images_per_s_lanes_est'], 'pair_ms', r['pair_ms'], 'lanes_pair_ms', r['lanes_pair_ms'])"
done
tail -n 2 $OUT/err2.txt
