#!/bin/bash
# Round-3 GPU call X: under CFG the init conv of the 256^2 stage runs on the B distinct images only (output + statistics copied to the null
# rows): whole-Unet parity at 256^2, a sampler test, step pair A/B, smoke.
set -u
cd "$(dirname "$0")/../.."
R=$PWD
OUT=$R/gpurun_out/r03_x
mkdir -p $OUT
timeout 400 python -m pytest tests/test_model_gpu.py -m gpu -q --tb=short -p no:cacheprovider -k "(unet_forward_vs_oracle and 256) or sample_vs_reference or time_table" > $OUT/pytest.log 2>&1
tail -n 6 $OUT/pytest.log | cut -c1-300
T="timeout 240 python tools/step_time.py"
$T --tag "init conv on the distinct images (product)" 2>$OUT/step.err | tee -a $OUT/step_times.jsonl
IMAGEN_INIT_CONV_SHARED=0 $T --tag "init conv on all CFG rows" 2>>$OUT/step.err | tee -a $OUT/step_times.jsonl
$T --tag "product again" 2>>$OUT/step.err | tee -a $OUT/step_times.jsonl
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
