#!/bin/bash
# Round 6, call A: write-through (sc1) output stores against plain stores (libimagen_hip_plainst.so = -DIMAGEN_WT_STORES=0): the step, sequential
# and six lanes, interleaved on one box; then the whole-denoiser parity of the image unets on three seeds and of C5 on three seeds.
#   gpurun --timeout 1500 -- 'bash tools/gpu/r06_a.sh'
set -u
cd "$(dirname "$0")/../.."
R=$PWD
OUT=$R/gpurun_out/r06_a
mkdir -p $OUT
rocm-smi --showclocks 2>/dev/null | grep -i "sclk\|socclk\|fclk\|mclk" | head -8 > $OUT/clocks_idle.txt
echo "=== step A/B: write-through vs plain stores (sequential + 6 lanes)"
for lib in libimagen_hip.so libimagen_hip_plainst.so libimagen_hip.so libimagen_hip_plainst.so; do
  IMAGEN_LIB_PATH=$R/imagen-pytorch_amd/$lib timeout 400 python tools/step_time.py --steps 60 --reps 3 --lanes 6 --tag $lib 2>/dev/null | tail -n 1 | tee -a $OUT/step_ab.jsonl
done
echo "=== kernel + fusion tests on the write-through library"
timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_fusion_gpu.py tests/test_rowchain_gpu.py -m gpu -q --tb=short -p no:cacheprovider -x > $OUT/pytest_kernels.txt 2>&1; tail -n 3 $OUT/pytest_kernels.txt | cut -c1-220
echo "=== whole-denoiser parity (image unets incl. seeds 1 / 2; C5 seeds 0 / 1 / 2)"
timeout 1200 python -m pytest tests/test_model_gpu.py "tests/test_video_gpu.py::test_unet3d_forward_vs_oracle_c5" -m gpu -q --tb=short -p no:cacheprovider > $OUT/pytest_parity.txt 2>&1; tail -n 40 $OUT/pytest_parity.txt | cut -c1-220
cp $R/gpurun_out/parity_measured.json $OUT/parity_measured.json 2>/dev/null
