#!/bin/bash
# Round-3 GPU call I: timing ablations of the wave-specialised kernel's two roles (results are garbage; only times mean anything).
set -u
cd "$(dirname "$0")/../.."
OUT=$PWD/gpurun_out/r03_i
mkdir -p $OUT
T="timeout 240 python tools/step_time.py --reps 2"
$T --tag "default" 2>$OUT/step.err | tee -a $OUT/step_times.jsonl
IMAGEN_ABLATE_IGEMM_DBG=17 $T --tag "producers idle (no loads, no transform, no LDS writes)" 2>>$OUT/step.err | tee -a $OUT/step_times.jsonl
IMAGEN_ABLATE_IGEMM_DBG=16 $T --tag "producers without global loads" 2>>$OUT/step.err | tee -a $OUT/step_times.jsonl
IMAGEN_ABLATE_IGEMM_DBG=1 $T --tag "producers without transform + LDS writes" 2>>$OUT/step.err | tee -a $OUT/step_times.jsonl
IMAGEN_ABLATE_IGEMM_DBG=19 $T --tag "barriers + epilogues only (no producer work, no consumer compute)" 2>>$OUT/step.err | tee -a $OUT/step_times.jsonl
IMAGEN_ABLATE_IGEMM_DBG=27 $T --tag "... and no stores" 2>>$OUT/step.err | tee -a $OUT/step_times.jsonl
IMAGEN_ABLATE_IGEMM_DBG=2 $T --tag "no consumer compute" 2>>$OUT/step.err | tee -a $OUT/step_times.jsonl
