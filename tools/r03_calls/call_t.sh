#!/bin/bash
# Round-3 GPU call T: would ONE batch-8 request run faster as two concurrent half-batches (the lanes mechanism inside a request)?
# step_time at batch 4 with 2 lanes vs batch 8 with 1 / 2 lanes, same box.
set -u
cd "$(dirname "$0")/../.."
R=$PWD
OUT=$R/gpurun_out/r03_t
mkdir -p $OUT
T="timeout 240 python tools/step_time.py"
$T --tag "batch 8, one request (+ 2 lanes)" --lanes 2 2>$OUT/step.err | tee -a $OUT/step_times.jsonl
$T --tag "batch 4 x 2 lanes" --batch 4 --lanes 2 2>>$OUT/step.err | tee -a $OUT/step_times.jsonl
$T --tag "batch 2 x 4 lanes" --batch 2 --lanes 4 2>>$OUT/step.err | tee -a $OUT/step_times.jsonl
