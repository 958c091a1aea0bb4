#!/bin/bash
# Round-3 GPU call R: conv_big with LDS-staged full-line stores, preloaded epilogue operands and the counted prologue wait: parity, kernel A/B,
# the three fixed-cost ablations again.
set -u
cd "$(dirname "$0")/../.."
R=$PWD
OUT=$R/gpurun_out/r03_r
mkdir -p $OUT
timeout 300 python -m pytest tests/test_igemm_cfgs_gpu.py -m gpu -q --tb=short -p no:cacheprovider -k "conv_big" > $OUT/pytest.log 2>&1
tail -n 6 $OUT/pytest.log | cut -c1-400
timeout 200 python tools/conv_bench.py --out $OUT/conv_bench.jsonl 2>$OUT/bench.err | cut -c1-200
timeout 100 python tools/conv_bench.py --gca --shapes 128:128:64 --out $OUT/conv_bench.jsonl 2>>$OUT/bench.err | cut -c1-200
IMAGEN_LIB_PATH=$R/imagen-pytorch_amd/libimagen_hip_ablate.so timeout 200 python tools/conv_bench.py --shapes 192:128:64 384:256:32 --cands big:0 big:1 --ablate 4 8 15 16 31 --out $OUT/ablate.jsonl 2>>$OUT/bench.err | cut -c1-160
