#!/bin/bash
# Round-3 GPU call G: timing ablations of the sampling loop (results are garbage, only the step times mean anything): what the consumers'
# MFMA work, the output stores and the producers' prologue arithmetic cost in the step.
set -u
cd "$(dirname "$0")/../.."
OUT=$PWD/gpurun_out/r03_g
mkdir -p $OUT
T="timeout 240 python tools/step_time.py"
$T --tag "default" 2>$OUT/step.err | tee -a $OUT/step_times.jsonl
IMAGEN_ABLATE_IGEMM_DBG=2 $T --tag "family-0 igemm without consumer compute (no weight loads, LDS reads, MFMAs)" 2>>$OUT/step.err | tee -a $OUT/step_times.jsonl
IMAGEN_ABLATE_IGEMM_DBG=8 $T --tag "family-0 igemm without output stores" 2>>$OUT/step.err | tee -a $OUT/step_times.jsonl
IMAGEN_ABLATE_PROLOGUE=1 $T --tag "3x3 convs without prologue arithmetic (raw inputs everywhere)" 2>>$OUT/step.err | tee -a $OUT/step_times.jsonl
IMAGEN_ABLATE_PROLOGUE=1 IMAGEN_CONV_DMA=0 IMAGEN_CONV_STREAM=0 $T --tag "... and everything on the wave-specialised family" 2>>$OUT/step.err | tee -a $OUT/step_times.jsonl
IMAGEN_CONV_DMA=0 $T --tag "no all-DMA family" 2>>$OUT/step.err | tee -a $OUT/step_times.jsonl
$T --tag "default again" 2>>$OUT/step.err | tee -a $OUT/step_times.jsonl
