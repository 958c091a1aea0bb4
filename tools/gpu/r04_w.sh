#!/bin/bash
# round 4, GPU call W: kernel tables of BASELINE configs 5 and 4 (rocprofv3 --kernel-trace --stats over the legs of bench.py)
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r04_w; mkdir -p $O
export TMPDIR=/tmp
for c in c5 c4; do
  cd /tmp && rm -rf /tmp/prof_$c
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$c -- python $GRAFT_REPO_ROOT/bench.py --config $c --steps 1 --warmup 1 --config-steps 30 > $O/bench_${c}_profiled.json 2> $O/bench_${c}_profiled.err
  cp $(find /tmp/prof_$c -name '*kernel_stats.csv' | head -n 1) $O/${c}_kernel_stats.csv
  echo "== $c"; head -n 14 $O/${c}_kernel_stats.csv | cut -c1-160
done
