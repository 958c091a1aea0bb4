#!/bin/bash
# Round 5, call A: the GPU tests (ROWCHAIN first), the step A/B of the fused token chains, a bench line with the new calibration, the in-graph profile.
#   gpurun --timeout 1500 -- 'bash tools/gpu/r05_a.sh'
set -u
cd "$(dirname "$0")/../.."
R=$PWD
OUT=$R/gpurun_out/r05_a
mkdir -p $OUT
rm -f $R/gpurun_out/parity_measured.json
python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1; tail -n 1 $OUT/build.log
echo "=== rowchain tests"
timeout 300 python -m pytest tests/test_rowchain_gpu.py -m gpu -q --tb=short -p no:cacheprovider > $OUT/pytest_rowchain.txt 2>&1; tail -n 40 $OUT/pytest_rowchain.txt | cut -c1-220
echo "=== step A/B"
for rc in 1 0; do IMAGEN_ROWCHAIN=$rc timeout 200 python tools/step_time.py --steps 60 --reps 3 --tag rowchain$rc 2>/dev/null | tail -n 1 | tee -a $OUT/step_ab.jsonl; done
echo "=== pytest -m gpu"
timeout 700 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x > $OUT/pytest_gpu.txt 2>&1
tail -n 40 $OUT/pytest_gpu.txt | cut -c1-250
cp $R/gpurun_out/parity_measured.json $OUT/parity.json 2>/dev/null
echo "=== smoke"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1; tail -n 3 $OUT/smoke.txt
echo "=== bench (6 lanes) with the new calibration"
timeout 600 python bench.py --steps 6 --warmup 6 --no-cpu-baseline --no-pmc > $OUT/bench_n1.json 2> $OUT/bench_n1.err
tail -n 4 $OUT/bench_n1.err; cut -c1-1800 $OUT/bench_n1.json
cd /tmp && export TMPDIR=/tmp
echo "=== in-graph per-op profile"
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/gp -- python $R/tools/graph_profile.py run --steps 12 --plan-out /tmp/plan.json > /tmp/gp.log 2>&1
f=$(find /tmp/gp -name "*kernel_trace.csv" | head -1)
python $R/tools/graph_profile.py analyze $f /tmp/plan.json --top 40 --csv $OUT/graph_profile > $OUT/graph_profile.txt 2>&1
grep -A 18 "===" $OUT/graph_profile.txt | cut -c1-120
