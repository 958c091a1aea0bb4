#!/bin/bash
# Round-3 GPU call L: the streaming pointwise family (conv_pw.hip) for the large res_conv launches; 16-byte pixel-shuffle stores.
set -u
cd "$(dirname "$0")/../.."
R=$PWD
OUT=$R/gpurun_out/r03_l
mkdir -p $OUT
rm -f $R/gpurun_out/parity_measured.json
timeout 420 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > $OUT/pytest.log 2>&1
tail -n 4 $OUT/pytest.log | cut -c1-300
T="timeout 240 python tools/step_time.py"
$T --tag "conv_pw on (product)" --lanes 6 2>$OUT/step.err | tee -a $OUT/step_times.jsonl
IMAGEN_CONV_PW=0 $T --tag "conv_pw off" --lanes 6 2>>$OUT/step.err | tee -a $OUT/step_times.jsonl
$T --tag "conv_pw on again" 2>>$OUT/step.err | tee -a $OUT/step_times.jsonl
cd /tmp && export TMPDIR=/tmp
timeout 240 rocprofv3 --kernel-trace --output-format csv -d /tmp/gp -- python $R/tools/graph_profile.py run --steps 12 --plan-out /tmp/plan.json > /tmp/gp.log 2>&1
f=$(find /tmp/gp -name "*kernel_trace.csv" | head -1)
python $R/tools/graph_profile.py analyze $f /tmp/plan.json --top 60 --csv $OUT/graph_profile > $OUT/graph_profile.txt 2>&1
grep -E "res_conv|ups.#.# \[" $OUT/graph_profile.txt | head -14 | cut -c1-120
