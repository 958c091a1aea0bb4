#!/bin/bash
# First GPU call of round 3 (prepared at the end of round 2, when the GPU budget was spent): everything that was written or found
# without hardware, in one call.  Build both libraries BEFORE the call, in the build container (the .so files travel with the snapshot):
#     python -c "import __graft_entry__ as g; g.build()" && bash tools/build_remat_lib.sh && \
#         bash tools/build_variant_lib.sh onewg -DIGEMM_ONE_WG -DIGEMM_LA1=12 -DIGEMM_LA2=10 && \
#         bash tools/build_variant_lib.sh remat8 -DIGEMM_EPI_REMAT -DIGEMM_LA2=8
#     gpurun --timeout 1500 -- 'bash tools/r03_calls/first_call.sh'
# Output: gpurun_out/r03_first/*.log|json.  Budget: ~25 GPU-minutes.
set -u
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r03_first
mkdir -p $OUT
REMAT=$PWD/imagen-pytorch_amd/libimagen_hip_remat.so
REMAT8=$PWD/imagen-pytorch_amd/libimagen_hip_remat8.so    # remat + the 8-deep weight ring for the 2-MFMA-per-step tilings (fits the 128-VGPR budget once the epilogue constants are gone)
ONEWG=$PWD/imagen-pytorch_amd/libimagen_hip_onewg.so      # 256-VGPR budget (one workgroup per CU), weight rings 12 / 10 deep: spill-free everywhere
B="python bench.py --steps 8 --warmup 4 --no-roofline --no-cpu-baseline"

# 1. the tests that have not met hardware (ElucidatedImagen sampler options, upsample combiner)
IMAGEN_UNVERIFIED_GPU_TESTS=1 timeout 300 python -m pytest tests/test_model_gpu.py -q -k "elucidated_sample_options or upsample_combiner" -s > $OUT/unverified_tests.log 2>&1
tail -n 3 $OUT/unverified_tests.log
IMAGEN_UNVERIFIED_GPU_TESTS=1 timeout 120 python -m pytest tests/test_video_gpu.py -q -k "cond_images" -s > $OUT/unverified_video_tests.log 2>&1
tail -n 2 $OUT/unverified_video_tests.log

# 2. parity of the -DIGEMM_EPI_REMAT library: every tile configuration x k-step path, and every distinct igemm launch of the benchmark's plans
if [ -f "$REMAT" ]; then
  IMAGEN_LIB_PATH=$REMAT timeout 400 python -m pytest tests/test_igemm_cfgs_gpu.py tests/test_bench_shapes_gpu.py tests/test_fusion_gpu.py -q -x > $OUT/remat_parity.log 2>&1
  tail -n 3 $OUT/remat_parity.log
else
  echo "no $REMAT: run tools/build_remat_lib.sh before the call" | tee $OUT/remat_parity.log
fi

if [ -f "$ONEWG" ]; then
  IMAGEN_LIB_PATH=$ONEWG timeout 300 python -m pytest tests/test_igemm_cfgs_gpu.py tests/test_bench_shapes_gpu.py -q -x > $OUT/onewg_parity.log 2>&1
  tail -n 3 $OUT/onewg_parity.log
fi

# 3. bench A/B, same box, same call (boxes of the pool differ by +-20 %): product | remat | persistent grids below the resident slot count
timeout 240 $B > $OUT/bench_default.json 2> $OUT/bench_default.err
[ -f "$REMAT" ] && IMAGEN_LIB_PATH=$REMAT timeout 240 $B > $OUT/bench_remat.json 2> $OUT/bench_remat.err
[ -f "$REMAT8" ] && IMAGEN_LIB_PATH=$REMAT8 timeout 240 $B > $OUT/bench_remat8.json 2> $OUT/bench_remat8.err   # (its parity rides on the remat library's: same code, one constant)
for pct in 94 88 80; do
  IMAGEN_GRID_PCT=$pct timeout 240 $B > $OUT/bench_grid$pct.json 2> $OUT/bench_grid$pct.err
done
[ -f "$ONEWG" ] && IMAGEN_LIB_PATH=$ONEWG timeout 240 $B > $OUT/bench_onewg.json 2> $OUT/bench_onewg.err
# ... and with the 128 px x 128 co tiles for the C_out >= 128 layers (half the weight bytes per MFMA; a loss of 3 % in the model under the product build)
[ -f "$ONEWG" ] && IMAGEN_LIB_PATH=$ONEWG IMAGEN_PICK_128=1 timeout 240 $B > $OUT/bench_onewg_pick128.json 2> $OUT/bench_onewg_pick128.err
IMAGEN_IGEMM_DBG=32 timeout 240 $B > $OUT/bench_one_tile_per_wg.json 2> $OUT/bench_one_tile_per_wg.err   # igemm grids non-persistent: other lanes' launches interleave as slots free up
[ -f "$REMAT" ] && IMAGEN_LIB_PATH=$REMAT IMAGEN_GRID_PCT=88 timeout 240 $B > $OUT/bench_remat_grid88.json 2> $OUT/bench_remat_grid88.err
# more hardware queues for the lanes' streams (the HIP runtime's default is 4 per process: lanes beyond that share a queue and serialise —
# a candidate reading of "three to four lanes, then flat"), alone and with room left on the chip
GPU_MAX_HW_QUEUES=8 timeout 300 $B --lanes 6 --steps 12 > $OUT/bench_hwq8_lanes6.json 2> $OUT/bench_hwq8_lanes6.err
GPU_MAX_HW_QUEUES=8 IMAGEN_GRID_PCT=88 timeout 300 $B --lanes 6 --steps 12 > $OUT/bench_hwq8_lanes6_grid88.json 2> $OUT/bench_hwq8_lanes6_grid88.err
timeout 240 $B > $OUT/bench_default_again.json 2> $OUT/bench_default_again.err      # drift of the box over the call
python - <<'PY'
import glob, json, os
for f in sorted(glob.glob("gpurun_out/r03_first/bench_*.json")):
    try:
        r = json.loads(open(f).read().strip().splitlines()[-1])
        print(f"{os.path.basename(f):32s} {r['value']:.4f} images/s  (sequential {r.get('sequential', {}).get('value')})")
    except Exception as e:   # noqa: BLE001
        print(f"{os.path.basename(f):32s} failed: {e}")
PY
