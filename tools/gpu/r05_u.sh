#!/bin/bash
# Round-5 call U (session 2, the last GPU seconds): the fp32 timestep-conditioning chain (IMAGEN_OP_LINEAR_F32, engine.TIME_CHAIN_F32) on
# hardware — its kernel test, then the whole-model file (whole-Unet parity on three draws, samplers, time table).
#   gpurun --timeout 165 -- 'bash tools/gpu/r05_u.sh'
set -u
cd "$(dirname "$0")/../.."
R=$PWD
OUT=$R/gpurun_out/r05_u
mkdir -p $OUT
timeout 40 python -m pytest tests/test_kernels_gpu.py -m gpu -q --tb=short -p no:cacheprovider -k "linear_f32 or time_embed_scale_shift" > $OUT/pytest_kernels.txt 2>&1; tail -n 3 $OUT/pytest_kernels.txt | cut -c1-200
timeout 125 python -m pytest tests/test_model_gpu.py -m gpu -q --tb=short -p no:cacheprovider --durations=6 > $OUT/pytest_model.txt 2>&1
grep -E "unet_forward_vs_oracle|passed|failed|Error|time_table" $OUT/pytest_model.txt | cut -c1-230 | tail -n 30
