#!/bin/bash
# Round-3 GPU call N: (1) every conv / GEMM launch issued twice inside the step graphs — the in-graph trace then holds each launch cold
# (operands last touched by other kernels) and warm (just read by itself): how much of the distance to the isolated-kernel rates is
# first-touch cost; (2) lanes x hardware-queue sweep of the whole-cascade throughput.
set -u
cd "$(dirname "$0")/../.."
R=$PWD
OUT=$R/gpurun_out/r03_n
mkdir -p $OUT
T="timeout 300 python tools/step_time.py"
$T --tag "lanes 6" --lanes 6 2>$OUT/step.err | tee -a $OUT/step_times.jsonl
$T --tag "lanes 8" --lanes 8 2>>$OUT/step.err | tee -a $OUT/step_times.jsonl
$T --tag "lanes 12" --lanes 12 2>>$OUT/step.err | tee -a $OUT/step_times.jsonl
GPU_MAX_HW_QUEUES=8 $T --tag "lanes 8, 8 hw queues" --lanes 8 2>>$OUT/step.err | tee -a $OUT/step_times.jsonl
GPU_MAX_HW_QUEUES=12 $T --tag "lanes 12, 12 hw queues" --lanes 12 2>>$OUT/step.err | tee -a $OUT/step_times.jsonl
GPU_MAX_HW_QUEUES=2 $T --tag "lanes 6, 2 hw queues" --lanes 6 2>>$OUT/step.err | tee -a $OUT/step_times.jsonl
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/gp -- python $R/tools/graph_profile.py run --steps 12 --dup-igemm --plan-out /tmp/plan.json > /tmp/gp.log 2>&1
tail -2 /tmp/gp.log
f=$(find /tmp/gp -name "*kernel_trace.csv" | head -1)
python $R/tools/graph_profile.py analyze $f /tmp/plan.json --top 40 --csv $OUT/graph_profile_dup > $OUT/graph_profile_dup.txt 2>&1
grep -E "block# \[(192|384|256|128)->" $OUT/graph_profile_dup.txt | head -24 | cut -c1-130
