set -x
cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
timeout 600 python bench.py > gpurun_out/bench_ckpt.json 2> gpurun_out/bench_ckpt.err; tail -2 gpurun_out/bench_ckpt.err; cat gpurun_out/bench_ckpt.json | cut -c1-600
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -- python $GRAFT_REPO_ROOT/bench.py --timesteps 50 --steps 1 --warmup 1 --no-cpu-baseline > /tmp/prof_stats.log 2>&1
f=$(find /tmp/prof_stats -name "*kernel_stats.csv" | head -1); cp $f $GRAFT_REPO_ROOT/gpurun_out/r01_kernel_stats_v4.csv; head -12 $f | cut -c1-160
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $c --output-format csv -d /tmp/pmc_$c -- python $GRAFT_REPO_ROOT/bench.py --timesteps 10 --steps 1 --warmup 0 --no-cpu-baseline --no-roofline > /tmp/pmc_$c.log 2>&1
  f=$(find /tmp/pmc_$c -name "*counter_collection.csv" | head -1)
  python $GRAFT_REPO_ROOT/tools/pmc_summary.py $f $GRAFT_REPO_ROOT/gpurun_out/r01_pmc_${c}_v2.json | head -3 | cut -c1-200
done
