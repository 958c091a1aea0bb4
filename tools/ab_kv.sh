set -x
cd $GRAFT_REPO_ROOT
timeout 200 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
for v in 0 1 0 1; do
  IMAGEN_KV_BATCH=$v timeout 60 python bench.py --timesteps 60 --steps 3 --warmup 1 --no-cpu-baseline --no-roofline 2>/dev/null | cut -c1-140
done
timeout 200 python bench.py > gpurun_out/bench_kvb.json 2> gpurun_out/bench_kvb.err; tail -2 gpurun_out/bench_kvb.err; cut -c1-700 gpurun_out/bench_kvb.json
