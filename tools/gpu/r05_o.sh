#!/bin/bash
# Round 5, call O: conv_small.hip's 1x1 mode (the res_conv / upsample GEMMs of the 8^2 / 16^2 maps, IMAGEN_CONV_SMALL=2) against the 3x3 convs only (=1):
# tests, bench shapes, the step.
#   gpurun --timeout 900 -- 'bash tools/gpu/r05_o.sh'
set -u
cd "$(dirname "$0")/../.."
R=$PWD
OUT=$R/gpurun_out/r05_o
mkdir -p $OUT
echo "=== tests"
timeout 400 python -m pytest tests/test_igemm_cfgs_gpu.py tests/test_bench_shapes_gpu.py -m gpu -q --tb=short -p no:cacheprovider -x > $OUT/pytest.txt 2>&1; tail -n 5 $OUT/pytest.txt | cut -c1-220
echo "=== step A/B"
for v in 2 1 0 2 1 0; do
  IMAGEN_CONV_SMALL=$v timeout 300 python tools/step_time.py --steps 60 --reps 3 --tag small$v 2>/dev/null | tail -n 1 | tee -a $OUT/step_ab.jsonl
done
