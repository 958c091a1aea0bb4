#!/bin/bash
# Round-3 GPU call S: conv_big in the model with the self-stat ACT_PREP (no ROWSTAT in front of it): parity (kernel cases, every distinct
# launch of the bench plans, whole Unets vs the oracle), step pair A/B, in-graph profile.
set -u
cd "$(dirname "$0")/../.."
R=$PWD
OUT=$R/gpurun_out/r03_s
mkdir -p $OUT
rm -f $R/gpurun_out/parity_measured.json
timeout 600 python -m pytest tests/test_igemm_cfgs_gpu.py tests/test_bench_shapes_gpu.py tests/test_model_gpu.py -m gpu -q --tb=short -p no:cacheprovider -k "act_prep or conv_big or bench or unet_forward" > $OUT/pytest.log 2>&1
tail -n 25 $OUT/pytest.log | cut -c1-300
T="timeout 240 python tools/step_time.py"
$T --tag "conv_big + self-stat prep (product)" --lanes 6 2>$OUT/step.err | tee -a $OUT/step_times.jsonl
IMAGEN_BIG_PREP=0 $T --tag "conv_big for raw launches only" --lanes 6 2>>$OUT/step.err | tee -a $OUT/step_times.jsonl
IMAGEN_CONV_BIG=0 $T --tag "conv_big off" --lanes 6 2>>$OUT/step.err | tee -a $OUT/step_times.jsonl
$T --tag "product again" 2>>$OUT/step.err | tee -a $OUT/step_times.jsonl
cd /tmp && export TMPDIR=/tmp
timeout 240 rocprofv3 --kernel-trace --output-format csv -d /tmp/gp -- python $R/tools/graph_profile.py run --steps 12 --plan-out /tmp/plan.json > /tmp/gp.log 2>&1
f=$(find /tmp/gp -name "*kernel_trace.csv" | head -1)
python $R/tools/graph_profile.py analyze $f /tmp/plan.json --top 60 --csv $OUT/graph_profile > $OUT/graph_profile.txt 2>&1
grep -E "=== stage|k3 @64\]|k3 @32\]|prep|act_prep|rowstat" $OUT/graph_profile.txt | head -40 | cut -c1-150
