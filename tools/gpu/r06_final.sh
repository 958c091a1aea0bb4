#!/bin/bash
# Round-6 measurement pass: the summaries the round's numbers come from (copied from gpurun_out/r06_final/ to profiles/r06_final_*).
#   gpurun --timeout 3000 -- 'bash tools/gpu/r06_final.sh'
set -u
cd "$(dirname "$0")/../.."
R=$PWD
OUT=$R/gpurun_out/r06_final
mkdir -p $OUT
rm -f $R/gpurun_out/parity_measured.json
python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1
echo "=== pytest -m gpu"
timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > $OUT/pytest_gpu.txt 2>&1
tail -n 14 $OUT/pytest_gpu.txt | cut -c1-250
cp $R/gpurun_out/parity_measured.json $OUT/parity.json 2>/dev/null
cp $R/gpurun_out/parity_bench_shapes.json $OUT/parity_bench_shapes.json 2>/dev/null
echo "=== smoke"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1; tail -n 2 $OUT/smoke.txt
echo "=== bench (default: 6 lanes, 1000 steps) with the live PMC passes, calibration and the CPU baseline"
timeout 1000 python bench.py --steps 6 --warmup 6 > $OUT/bench_n1.json 2> $OUT/bench_n1.err
tail -n 6 $OUT/bench_n1.err; cut -c1-3000 $OUT/bench_n1.json
cd /tmp && export TMPDIR=/tmp
echo "=== rocprofv3 kernel stats of the sampling loop (sequential, 60 timesteps)"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -- python $R/bench.py --mode sequential --timesteps 60 --steps 1 --warmup 1 --no-cpu-baseline --no-roofline > /tmp/prof_stats.log 2>&1
f=$(find /tmp/prof_stats -name "*kernel_stats.csv" | head -1); cp $f $OUT/kernel_stats.csv; head -14 $f | cut -c1-170
echo "=== in-graph per-op profile"
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/gp -- python $R/tools/graph_profile.py run --steps 12 --plan-out /tmp/plan.json > /tmp/gp.log 2>&1
f=$(find /tmp/gp -name "*kernel_trace.csv" | head -1)
python $R/tools/graph_profile.py analyze $f /tmp/plan.json --top 60 --csv $OUT/graph_profile > $OUT/graph_profile.txt 2>&1
grep -A 14 "===" $OUT/graph_profile.txt | cut -c1-120
echo "=== PMC passes joined with the plan"
for c in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU GRBM_GUI_ACTIVE"; do
  tag=$(echo $c | cut -d' ' -f1)
  rm -rf /tmp/pmc_$tag
  timeout 300 rocprofv3 --pmc $c --output-format csv -d /tmp/pmc_$tag -- python $R/tools/graph_profile.py run --steps 5 --plan-out /tmp/plan_$tag.json > /tmp/pmc_$tag.log 2>&1
  f=$(find /tmp/pmc_$tag -name "*counter_collection.csv" | head -1)
  python $R/tools/graph_profile.py pmc $f /tmp/plan_$tag.json $OUT/pmc_$tag.json | cut -c1-300
done
echo "=== BASELINE's other configurations (bench.py --config)"
cd $R
for c in c2 c4 c5; do timeout 400 python bench.py --config $c --steps 2 --config-steps 50 2>$OUT/bench_$c.err | tail -n 1 > $OUT/bench_$c.json; cut -c1-600 $OUT/bench_$c.json; done
