#!/bin/bash
# Round-3 GPU call K: producers touch the next tile's epilogue operands (gate*addend | residual rows) into L2 vs the build without it.
set -u
cd "$(dirname "$0")/../.."
R=$PWD
OUT=$R/gpurun_out/r03_k
mkdir -p $OUT
rm -f $R/gpurun_out/parity_measured.json
timeout 420 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > $OUT/pytest.log 2>&1
tail -n 4 $OUT/pytest.log | cut -c1-300
T="timeout 240 python tools/step_time.py"
$T --tag "operand touch (product)" --lanes 6 2>$OUT/step.err | tee -a $OUT/step_times.jsonl
IMAGEN_LIB_PATH=$R/imagen-pytorch_amd/libimagen_hip_notouch.so $T --tag "no operand touch (-DIGEMM_NO_TOUCH)" --lanes 6 2>>$OUT/step.err | tee -a $OUT/step_times.jsonl
$T --tag "operand touch again" 2>>$OUT/step.err | tee -a $OUT/step_times.jsonl
cd /tmp && export TMPDIR=/tmp
timeout 240 rocprofv3 --kernel-trace --output-format csv -d /tmp/gp -- python $R/tools/graph_profile.py run --steps 12 --plan-out /tmp/plan.json > /tmp/gp.log 2>&1
f=$(find /tmp/gp -name "*kernel_trace.csv" | head -1)
python $R/tools/graph_profile.py analyze $f /tmp/plan.json --top 60 --csv $OUT/graph_profile > $OUT/graph_profile.txt 2>&1
grep "res_conv" $OUT/graph_profile.txt | head -12 | cut -c1-120
