#!/bin/bash
# (historical: the IMAGEN_TAIL_* / IMAGEN_LN_STATS_FUSED switches this call used were folded into module constants of engine.py after the measurement)
# Round-3 GPU call B: the cleaned-up library (remat epilogue as the product build) + GCA_TAIL on hardware.
#   (1) whole -m gpu suite, (2) step-time A/B of the planner switches on one box, (3) in-graph profile of the new default.
set -u
cd "$(dirname "$0")/../.."
R=$PWD
OUT=$R/gpurun_out/r03_b
mkdir -p $OUT
rm -f $R/gpurun_out/parity_measured.json
timeout 420 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > $OUT/pytest.log 2>&1
tail -n 25 $OUT/pytest.log
cp $R/gpurun_out/parity_measured.json $OUT/ 2>/dev/null
T="timeout 150 python tools/step_time.py"
$T --tag default --lanes 3 2>$OUT/step_default.err | tee -a $OUT/step_times.jsonl
IMAGEN_TAIL_FUSED=0 IMAGEN_LN_STATS_FUSED=0 $T --tag "no tail fusion, no fused LN stats (round-2 plan)" 2>>$OUT/step.err | tee -a $OUT/step_times.jsonl
IMAGEN_TAIL_ACT=0 $T --tag "tail fused, no activated tensor" 2>>$OUT/step.err | tee -a $OUT/step_times.jsonl
IMAGEN_LN_STATS_FUSED=0 $T --tag "no fused LN stats" 2>>$OUT/step.err | tee -a $OUT/step_times.jsonl
$T --tag "default again" 2>>$OUT/step.err | tee -a $OUT/step_times.jsonl
cd /tmp && export TMPDIR=/tmp
timeout 240 rocprofv3 --kernel-trace --output-format csv -d /tmp/gp -- python $R/tools/graph_profile.py run --steps 12 --plan-out /tmp/plan.json > /tmp/gp.log 2>&1
f=$(find /tmp/gp -name "*kernel_trace.csv" | head -1)
python $R/tools/graph_profile.py analyze $f /tmp/plan.json --top 60 --csv $OUT/graph_profile > $OUT/graph_profile.txt 2>&1
grep -A 14 "===" $OUT/graph_profile.txt | cut -c1-120
