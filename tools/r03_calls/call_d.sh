#!/bin/bash
# (historical: the IMAGEN_TAIL_* / IMAGEN_LN_STATS_FUSED switches this call used were folded into module constants of engine.py after the measurement)
# Round-3 GPU call D: lanes beyond 6; 64-pixel tiles for the all-DMA conv family (more workgroups per CU); calibration beside each.
set -u
cd "$(dirname "$0")/../.."
R=$PWD
OUT=$R/gpurun_out/r03_d
mkdir -p $OUT
T="timeout 240 python tools/step_time.py"
$T --tag "default, lanes 6" --lanes 6 2>$OUT/step.err | tee -a $OUT/step_times.jsonl
$T --tag "lanes 8" --lanes 8 --reps 1 2>>$OUT/step.err | tee -a $OUT/step_times.jsonl
$T --tag "lanes 10" --lanes 10 --reps 1 2>>$OUT/step.err | tee -a $OUT/step_times.jsonl
IMAGEN_DMA_PREFER_64=1 $T --tag "conv_dma: 64-pixel tiles" --lanes 6 2>>$OUT/step.err | tee -a $OUT/step_times.jsonl
IMAGEN_TAIL_FUSED=0 $T --tag "no fused tails" 2>>$OUT/step.err | tee -a $OUT/step_times.jsonl
$T --tag "default again" 2>>$OUT/step.err | tee -a $OUT/step_times.jsonl
python - <<'PY'
import torch, sys
sys.path.insert(0, '.')
import bench
print("calibration:", bench.calibration_leg(torch.device("cuda:0")))
PY
cd /tmp && export TMPDIR=/tmp
timeout 240 rocprofv3 --kernel-trace --output-format csv -d /tmp/gp -- python $R/tools/graph_profile.py run --steps 12 --plan-out /tmp/plan.json > /tmp/gp.log 2>&1
f=$(find /tmp/gp -name "*kernel_trace.csv" | head -1)
python $R/tools/graph_profile.py analyze $f /tmp/plan.json --top 60 --csv $OUT/graph_profile > $OUT/graph_profile.txt 2>&1
grep -A 12 "===" $OUT/graph_profile.txt | cut -c1-120
