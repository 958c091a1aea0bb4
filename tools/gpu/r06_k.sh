#!/bin/bash
# Round 6, call K: kernel table of BASELINE C5 (Imagen-Video) after the one-launch temporal conv.
set -u
cd "$(dirname "$0")/../.."
R=$PWD
OUT=$R/gpurun_out/r06_k
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_c5
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c5 -- python $R/bench.py --config c5 --steps 2 --warmup 1 --config-steps 40 > $OUT/bench_c5_profiled.json 2> $OUT/bench_c5_profiled.err
f=$(find /tmp/prof_c5 -name '*kernel_stats.csv' | head -n 1); cp $f $OUT/c5_kernel_stats.csv
head -n 24 $OUT/c5_kernel_stats.csv | cut -c1-210
tail -n 1 $OUT/bench_c5_profiled.json | cut -c1-300
