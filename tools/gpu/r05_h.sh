#!/bin/bash
# Round 5, call H: conv_small.hip (igemm family 8, the small-map 3x3 convs with the K loop split over the waves of a workgroup) on hardware:
# its tests, the whole-Unet parity with it routed, the step A/B (off | 8^2 + 16^2 | + 32^2), C2 / C4 with and without it, per-kernel durations.
#   gpurun --timeout 1500 -- 'bash tools/gpu/r05_h.sh'
# (IMAGEN_CONV_SMALL_ROWS was an environment knob of ops.SMALL_MAX_ROWS at the commit this call ran on; the limit is a module constant since.)
set -u
cd "$(dirname "$0")/../.."
R=$PWD
OUT=$R/gpurun_out/r05_h
mkdir -p $OUT
rm -f $R/gpurun_out/parity_measured.json
echo "=== conv_small tests + every cfg"
timeout 500 python -m pytest tests/test_igemm_cfgs_gpu.py -m gpu -q --tb=short -p no:cacheprovider -x > $OUT/pytest_cfgs.txt 2>&1; tail -n 12 $OUT/pytest_cfgs.txt | cut -c1-220
echo "=== bench shapes + whole-Unet parity"
timeout 600 python -m pytest tests/test_bench_shapes_gpu.py tests/test_model_gpu.py -m gpu -q --tb=short -p no:cacheprovider > $OUT/pytest_model.txt 2>&1; tail -n 40 $OUT/pytest_model.txt | cut -c1-220
cp $R/gpurun_out/parity_measured.json $OUT/parity.json 2>/dev/null
echo "=== step A/B"
for v in "0 4096" "1 4096" "1 16384" "0 4096" "1 4096"; do
  set -- $v
  IMAGEN_CONV_SMALL=$1 IMAGEN_CONV_SMALL_ROWS=$2 timeout 200 python tools/step_time.py --steps 60 --reps 3 --tag small$1_rows$2 2>/dev/null | tail -n 1 | tee -a $OUT/step_ab.jsonl
done
IMAGEN_LIB_PATH=$R/imagen-pytorch_amd/libimagen_hip_cs18.so timeout 200 python tools/step_time.py --steps 60 --reps 3 --tag small1_rows4096_one_wg_per_cu 2>/dev/null | tail -n 1 | tee -a $OUT/step_ab.jsonl
IMAGEN_CONV_SMALL_ROWS=16384 IMAGEN_LIB_PATH=$R/imagen-pytorch_amd/libimagen_hip_cs18.so timeout 200 python tools/step_time.py --steps 60 --reps 3 --tag small1_rows16384_one_wg_per_cu 2>/dev/null | tail -n 1 | tee -a $OUT/step_ab.jsonl
echo "=== C2 / C4 with and without"
for c in c2 c4; do for v in 0 1; do
  IMAGEN_CONV_SMALL=$v timeout 400 python bench.py --config $c --steps 2 --config-steps 50 2>$OUT/bench_${c}_$v.err | tail -n 1 > $OUT/bench_${c}_small$v.json; cut -c1-400 $OUT/bench_${c}_small$v.json; echo
done; done
cd /tmp && export TMPDIR=/tmp
echo "=== per-kernel durations inside the loop"
for v in "1 4096" "1 16384"; do
  set -- $v
  rm -rf /tmp/kt
  IMAGEN_CONV_SMALL=$1 IMAGEN_CONV_SMALL_ROWS=$2 timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -- python $R/tools/graph_profile.py run --steps 12 --plan-out /tmp/plan.json > /tmp/kt.log 2>&1
  f=$(find /tmp/kt -name "*kernel_trace.csv" | head -1)
  python $R/tools/graph_profile.py analyze $f /tmp/plan.json --top 60 --csv $OUT/graph_profile_rows$2 > $OUT/graph_profile_rows$2.txt 2>&1
  grep -A 6 "===" $OUT/graph_profile_rows$2.txt | cut -c1-140
done
