#!/bin/bash
# Round-3 GPU call U: the per-request time table (the timestep-only conditioning chain of every step evaluated for all steps at once, one
# STEP_SLICE copy per step): sampler parity (reference fixtures, table vs chain, lanes / pipelined, checkpoints), step pair A/B.
set -u
cd "$(dirname "$0")/../.."
R=$PWD
OUT=$R/gpurun_out/r03_u
mkdir -p $OUT
timeout 600 python -m pytest tests/test_model_gpu.py -m gpu -q --tb=short -p no:cacheprovider -k "sample or time_table or sampling or handle" > $OUT/pytest.log 2>&1
tail -n 8 $OUT/pytest.log | cut -c1-300
T="timeout 240 python tools/step_time.py"
$T --tag "time table (product)" --lanes 6 2>$OUT/step.err | tee -a $OUT/step_times.jsonl
IMAGEN_TIME_TABLE=0 $T --tag "per-step chain" --lanes 6 2>>$OUT/step.err | tee -a $OUT/step_times.jsonl
$T --tag "time table again" 2>>$OUT/step.err | tee -a $OUT/step_times.jsonl
tail -n 2 $OUT/step.err | cut -c1-300
