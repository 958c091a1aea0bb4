#!/bin/bash
# The driver's own round-end command, timed (does the line still arrive within minutes with the merged-requests leg?).
set -u
cd "$(dirname "$0")/../.."
OUT=$PWD/gpurun_out/r06_driver_cmd
mkdir -p $OUT
SECONDS=0; timeout 1500 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err
echo "wall seconds: $SECONDS"; grep "bench +" $OUT/bench.err | tail -n 12; cut -c1-400 $OUT/bench.json
