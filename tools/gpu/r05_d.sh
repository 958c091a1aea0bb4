#!/bin/bash
# Round 5, call D: where a ROWCHAIN launch spends its time (s_memtime stamps of a -DROWCHAIN_TRACE variant library), lanes sweep.
set -u
cd "$(dirname "$0")/../.."
R=$PWD
OUT=$R/gpurun_out/r05_d
mkdir -p $OUT
echo "=== phase timeline"
IMAGEN_LIB_PATH=$R/imagen-pytorch_amd/libimagen_hip_trace.so timeout 300 python tools/chain_bench.py --trace --tag trace --reps 8 2>/dev/null | tail -n 1 | tee $OUT/chain_trace.json | cut -c1-4000
echo "=== lanes sweep (60 steps per stage)"
for l in 4 6 8; do timeout 300 python tools/step_time.py --steps 60 --reps 2 --lanes $l --tag lanes$l 2>/dev/null | tail -n 1 | tee -a $OUT/lanes_sweep.jsonl; done
