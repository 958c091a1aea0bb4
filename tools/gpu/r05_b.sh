#!/bin/bash
# Round 5, call B: ROWCHAIN v2 (16-deep weight rings requested a stage ahead, rows in registers, K^/V^T prefetch) against call A's v1 on the
# step and per launch; lanes vs staged lanes on the bench.
#   gpurun --timeout 1500 -- 'bash tools/gpu/r05_b.sh'
set -u
cd "$(dirname "$0")/../.."
R=$PWD
OUT=$R/gpurun_out/r05_b
mkdir -p $OUT
python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1; tail -n 1 $OUT/build.log
echo "=== rowchain + video tests"
timeout 300 python -m pytest tests/test_rowchain_gpu.py tests/test_video_gpu.py -m gpu -q --tb=short -p no:cacheprovider > $OUT/pytest_b.txt 2>&1; tail -n 8 $OUT/pytest_b.txt | cut -c1-220
echo "=== step A/B"
for rc in 1 0; do IMAGEN_ROWCHAIN=$rc timeout 200 python tools/step_time.py --steps 60 --reps 3 --tag rowchain$rc 2>/dev/null | tail -n 1 | tee -a $OUT/step_ab.jsonl; done
echo "=== bench: lanes vs staged (12 passes each)"
for m in lanes staged; do
  timeout 500 python bench.py --steps 12 --warmup 6 --mode $m --no-cpu-baseline --no-pmc --no-roofline > $OUT/bench_$m.json 2> $OUT/bench_$m.err
  tail -n 3 $OUT/bench_$m.err | cut -c1-200; python -c "
import json,sys
r=json.load(open('$OUT/bench_$m.json'))
print('$m', r['value'], r['ms_per_step'], 'seq', r.get('sequential',{}).get('value'), r.get('sequential',{}).get('in_graph_step_ms'), r.get('calibration'), r.get('calibration_error'))"
done
cd /tmp && export TMPDIR=/tmp
echo "=== in-graph per-op profile"
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/gp -- python $R/tools/graph_profile.py run --steps 12 --plan-out /tmp/plan.json > /tmp/gp.log 2>&1
f=$(find /tmp/gp -name "*kernel_trace.csv" | head -1)
python $R/tools/graph_profile.py analyze $f /tmp/plan.json --top 40 --csv $OUT/graph_profile > $OUT/graph_profile.txt 2>&1
grep -A 8 "===" $OUT/graph_profile.txt | cut -c1-120
grep "chain" $OUT/graph_profile.stage1.csv $OUT/graph_profile.stage2.csv | cut -d, -f1,4 | tr '\n' ' '
