#!/bin/bash
# Round-3 GPU call P: the big-tile all-DMA family (conv_big.hip, family 5) meets hardware: parity of every configuration, kernel-level A/B on
# the benchmark's C >= 128 shapes, the step pair with / without it (ACT_PREP + conv_big for the prologue Blocks).
set -u
cd "$(dirname "$0")/../.."
R=$PWD
OUT=$R/gpurun_out/r03_p
mkdir -p $OUT
timeout 300 python -m pytest tests/test_igemm_cfgs_gpu.py -m gpu -q --tb=short -p no:cacheprovider -k "conv_big" > $OUT/pytest.log 2>&1
tail -n 6 $OUT/pytest.log | cut -c1-400
timeout 200 python tools/conv_bench.py --out $OUT/conv_bench.jsonl 2>$OUT/bench.err | cut -c1-220
timeout 100 python tools/conv_bench.py --gca --shapes 128:128:64 --out $OUT/conv_bench.jsonl 2>>$OUT/bench.err | cut -c1-220
T="timeout 240 python tools/step_time.py"
$T --tag "conv_big on" 2>$OUT/step.err | tee -a $OUT/step_times.jsonl
IMAGEN_CONV_BIG=0 $T --tag "conv_big off" 2>>$OUT/step.err | tee -a $OUT/step_times.jsonl
IMAGEN_BIG_PREP=0 $T --tag "conv_big on, no prep (raw launches only)" 2>>$OUT/step.err | tee -a $OUT/step_times.jsonl
tail -n 3 $OUT/step.err | cut -c1-300
