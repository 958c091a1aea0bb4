#!/bin/bash
# Round-3 GPU call H: producers with three staging register sets (every load gets two phase periods to arrive) vs the two-set build.
set -u
cd "$(dirname "$0")/../.."
R=$PWD
OUT=$R/gpurun_out/r03_h
mkdir -p $OUT
rm -f $R/gpurun_out/parity_measured.json
timeout 420 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > $OUT/pytest.log 2>&1
tail -n 5 $OUT/pytest.log | cut -c1-300
cp $R/gpurun_out/parity_measured.json $OUT/ 2>/dev/null
T="timeout 240 python tools/step_time.py"
$T --tag "three staging sets (product)" --lanes 6 2>$OUT/step.err | tee -a $OUT/step_times.jsonl
IMAGEN_LIB_PATH=$R/imagen-pytorch_amd/libimagen_hip_ns2.so $T --tag "two staging sets (-DIGEMM_FORCE_NS2)" --lanes 6 2>>$OUT/step.err | tee -a $OUT/step_times.jsonl
$T --tag "three staging sets again" 2>>$OUT/step.err | tee -a $OUT/step_times.jsonl
cd /tmp && export TMPDIR=/tmp
timeout 240 rocprofv3 --kernel-trace --output-format csv -d /tmp/gp -- python $R/tools/graph_profile.py run --steps 12 --plan-out /tmp/plan.json > /tmp/gp.log 2>&1
f=$(find /tmp/gp -name "*kernel_trace.csv" | head -1)
python $R/tools/graph_profile.py analyze $f /tmp/plan.json --top 60 --csv $OUT/graph_profile > $OUT/graph_profile.txt 2>&1
grep -A 8 "===" $OUT/graph_profile.txt | cut -c1-120
