# Round-2 measurement pass (run on the GPU box through gpurun; summaries land in gpurun_out/, the ones to keep are copied to profiles/).
#   gpurun --timeout 2400 -- 'bash tools/measure_round2.sh'
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
mkdir -p gpurun_out
if [ -z "$SKIP_PYTEST" ]; then rm -f gpurun_out/r02_parity_model.json; fi
echo "=== pytest -m gpu"
if [ -z "$SKIP_PYTEST" ]; then timeout 1500 python -m pytest tests -m gpu -x -q --tb=short 2>&1 | tail -6 | tee gpurun_out/r02_pytest_gpu.txt; fi
echo "=== bench (default schedule, 1000 steps) with live PMC passes"
timeout 1200 python bench.py --steps 3 --warmup 3 > gpurun_out/r02_bench.json 2> gpurun_out/r02_bench.err; tail -12 gpurun_out/r02_bench.err; cut -c1-1500 gpurun_out/r02_bench.json
cd /tmp && export TMPDIR=/tmp
echo "=== rocprofv3 kernel stats of the bench (sequential, 60 timesteps)"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -- python $R/bench.py --mode sequential --timesteps 60 --steps 1 --warmup 1 --no-cpu-baseline --no-roofline > /tmp/prof_stats.log 2>&1
f=$(find /tmp/prof_stats -name "*kernel_stats.csv" | head -1); cp $f $R/gpurun_out/r02_kernel_stats.csv; head -14 $f | cut -c1-170
echo "=== in-graph per-op profile"
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/gp -- python $R/tools/graph_profile.py run --steps 12 --plan-out /tmp/plan.json > /tmp/gp.log 2>&1
f=$(find /tmp/gp -name "*kernel_trace.csv" | head -1)
python $R/tools/graph_profile.py analyze $f /tmp/plan.json --top 60 --csv $R/gpurun_out/r02_graph_profile > $R/gpurun_out/r02_graph_profile.txt 2>&1
grep -A 14 "===" $R/gpurun_out/r02_graph_profile.txt | cut -c1-120
echo "=== PMC passes joined with the plan"
for c in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU GRBM_GUI_ACTIVE"; do
  tag=$(echo $c | cut -d' ' -f1)
  rm -rf /tmp/pmc_$tag
  timeout 600 rocprofv3 --pmc $c --output-format csv -d /tmp/pmc_$tag -- python $R/tools/graph_profile.py run --steps 5 --plan-out /tmp/plan_$tag.json > /tmp/pmc_$tag.log 2>&1
  f=$(find /tmp/pmc_$tag -name "*counter_collection.csv" | head -1)
  python $R/tools/graph_profile.py pmc $f /tmp/plan_$tag.json $R/gpurun_out/r02_pmc_$tag.json | cut -c1-300
done
echo "=== C5 (Imagen-Video) timing"
cd $R; timeout 600 python tools/time_c5.py 2>&1 | tail -4 | tee gpurun_out/r02_c5.txt
