#!/bin/bash
# Round-3 GPU call W: the final conv of the 256^2 stage on the streaming family (low-res image packed once per request as a 32-channel
# tensor): whole-Unet parity at 256^2, the sampler tests, step pair A/B.
set -u
cd "$(dirname "$0")/../.."
R=$PWD
OUT=$R/gpurun_out/r03_w
mkdir -p $OUT
timeout 600 python -m pytest tests/test_model_gpu.py tests/test_bench_shapes_gpu.py -m gpu -q --tb=short -p no:cacheprovider -k "(unet_forward_vs_oracle and 256) or sample_vs_reference or bench or pipelined" > $OUT/pytest.log 2>&1
tail -n 8 $OUT/pytest.log | cut -c1-300
T="timeout 240 python tools/step_time.py"
$T --tag "final conv streaming (product)" 2>$OUT/step.err | tee -a $OUT/step_times.jsonl
IMAGEN_FINAL_CONV_STREAM=0 $T --tag "final conv wave-specialised (8-channel second input)" 2>>$OUT/step.err | tee -a $OUT/step_times.jsonl
$T --tag "product again" 2>>$OUT/step.err | tee -a $OUT/step_times.jsonl
