#!/bin/bash
# round 4, GPU call Y: kernel table of C5 after the MFMA temporal attention
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r04_y; mkdir -p $O
export TMPDIR=/tmp
cd /tmp && rm -rf /tmp/prof_c5
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c5 -- python $GRAFT_REPO_ROOT/bench.py --config c5 --steps 1 --warmup 1 --config-steps 30 > $O/bench_c5_profiled.json 2> $O/bench_c5_profiled.err
cp $(find /tmp/prof_c5 -name '*kernel_stats.csv' | head -n 1) $O/c5_kernel_stats.csv
head -n 24 $O/c5_kernel_stats.csv | cut -c1-170
