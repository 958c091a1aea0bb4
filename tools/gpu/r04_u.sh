#!/bin/bash
# round 4, GPU call U: kernel table of BASELINE config 2 (rocprofv3 --kernel-trace --stats over the C2 leg of bench.py)
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r04_u; mkdir -p $O
export TMPDIR=/tmp
cd /tmp && rm -rf /tmp/c2prof
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/c2prof -- python $GRAFT_REPO_ROOT/bench.py --config c2 --steps 1 --warmup 1 --config-steps 40 > $O/bench_c2_profiled.json 2> $O/bench_c2_profiled.err
cp $(find /tmp/c2prof -name '*kernel_stats.csv' | head -n 1) $O/c2_kernel_stats.csv
head -n 16 $O/c2_kernel_stats.csv | cut -c1-170
