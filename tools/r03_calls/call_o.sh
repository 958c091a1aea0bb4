#!/bin/bash
# Round-3 GPU call O: the single-buffered form of the streaming conv (two workgroups per CU) for the two-input prologue convs of the
# 256^2 level (3 launches of 104 us per unet2 step).
set -u
cd "$(dirname "$0")/../.."
R=$PWD
OUT=$R/gpurun_out/r03_o
mkdir -p $OUT
timeout 300 python -m pytest tests/test_igemm_cfgs_gpu.py tests/test_bench_shapes_gpu.py -m gpu -q --tb=short -p no:cacheprovider -k "conv_stream_family or bench_igemm" > $OUT/pytest.log 2>&1
tail -n 4 $OUT/pytest.log | cut -c1-300
T="timeout 300 python tools/step_time.py"
$T --tag "stream SB on (product)" --lanes 6 2>$OUT/step.err | tee -a $OUT/step_times.jsonl
IMAGEN_STREAM_SB=0 $T --tag "stream SB off" --lanes 6 2>>$OUT/step.err | tee -a $OUT/step_times.jsonl
$T --tag "stream SB on again" 2>>$OUT/step.err | tee -a $OUT/step_times.jsonl
cd /tmp && export TMPDIR=/tmp
timeout 240 rocprofv3 --kernel-trace --output-format csv -d /tmp/gp -- python $R/tools/graph_profile.py run --steps 12 --plan-out /tmp/plan.json > /tmp/gp.log 2>&1
f=$(find /tmp/gp -name "*kernel_trace.csv" | head -1)
python $R/tools/graph_profile.py analyze $f /tmp/plan.json --top 60 --csv $OUT/graph_profile > $OUT/graph_profile.txt 2>&1
grep -E "64->32 k3 @256|=== stage" $OUT/graph_profile.txt | head -8 | cut -c1-140
