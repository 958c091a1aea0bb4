#!/bin/bash
# Round 6, call G: the gate derivation with per-wave shuffle sums + an unrolled 16-wave final sum in its two matrix-vector stages and the instruction
# warm-up at kernel entry (libimagen_hip.so) against the serial column walk of rounds 2-5 (libimagen_hip_gcaold.so = -DIMAGEN_GCA_AB_OLD): the phase
# timeline of both (-DGCA_TRACE twins), the GlobalContext / tail tests, the step interleaved on one box.
#   gpurun --timeout 1200 -- 'bash tools/gpu/r06_g.sh'
set -u
cd "$(dirname "$0")/../.."
R=$PWD
OUT=$R/gpurun_out/r06_g
mkdir -p $OUT
echo "=== tests"
timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_fusion_gpu.py -m gpu -q --tb=short -p no:cacheprovider -x > $OUT/pytest_kernels.txt 2>&1; tail -n 3 $OUT/pytest_kernels.txt | cut -c1-220
for v in gcatrace gcatrace_old; do
  echo "=== phase timeline: $v"
  IMAGEN_LIB_PATH=$R/imagen-pytorch_amd/libimagen_hip_$v.so timeout 300 python tools/gca_bench.py --trace --tag $v > $OUT/$v.json 2> $OUT/$v.err
  python - <<PY
import json
d = json.load(open("$OUT/$v.json"))
for k, v in d["trace"].items():
    print(f"{k:44s} wgs={v['wgs']:5d} per_wg={v['per_wg_ticks']:9.0f}  phases={v['phase_ticks']}")
PY
done
echo "=== step A/B (sequential + 6 lanes)"
for lib in libimagen_hip.so libimagen_hip_gcaold.so libimagen_hip.so libimagen_hip_gcaold.so; do
  IMAGEN_LIB_PATH=$R/imagen-pytorch_amd/$lib timeout 400 python tools/step_time.py --steps 60 --reps 3 --lanes 6 --tag $lib 2>/dev/null | tail -n 1 | tee -a $OUT/step_ab.jsonl
done
echo "=== whole-denoiser parity on the bench's own plans + samplers"
timeout 900 python -m pytest tests/test_model_gpu.py -m gpu -q --tb=short -p no:cacheprovider -k "rows16 or sample_vs_reference or graph" > $OUT/pytest_parity.txt 2>&1; tail -n 6 $OUT/pytest_parity.txt | cut -c1-220
