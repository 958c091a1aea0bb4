#!/bin/bash
# The bench leg of r06_final.sh alone (after the roofline leg learned to skip the merged-requests leg's batch-24 stages).
set -u
cd "$(dirname "$0")/../.."
R=$PWD
OUT=$R/gpurun_out/r06_final
mkdir -p $OUT
timeout 1200 python bench.py --steps 6 --warmup 6 > $OUT/bench_n1.json 2> $OUT/bench_n1.err
tail -n 6 $OUT/bench_n1.err; cut -c1-1500 $OUT/bench_n1.json
