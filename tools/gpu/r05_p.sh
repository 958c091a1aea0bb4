#!/bin/bash
# Round 5, call P: two more variants of conv_small.hip against the product — eight walk phases (csph8); 16 weight fragments in flight per wave with one
# workgroup per CU (csr16): per launch and on the step.
#   gpurun --timeout 900 -- 'bash tools/gpu/r05_p.sh'
set -u
cd "$(dirname "$0")/../.."
R=$PWD
OUT=$R/gpurun_out/r05_p
mkdir -p $OUT
echo "=== step A/B"
for lib in libimagen_hip.so libimagen_hip_csph8.so libimagen_hip_csr16.so libimagen_hip.so libimagen_hip_csph8.so libimagen_hip_csr16.so; do
  IMAGEN_LIB_PATH=$R/imagen-pytorch_amd/$lib timeout 300 python tools/step_time.py --steps 60 --reps 3 --tag $lib 2>/dev/null | tail -n 1 | tee -a $OUT/step_ab.jsonl | cut -c1-120
done
cd /tmp && export TMPDIR=/tmp
echo "=== per launch"
for lib in libimagen_hip.so libimagen_hip_csph8.so libimagen_hip_csr16.so; do
  rm -rf /tmp/sb
  IMAGEN_LIB_PATH=$R/imagen-pytorch_amd/$lib timeout 250 rocprofv3 --kernel-trace --output-format csv -d /tmp/sb -- python $R/tools/small_bench.py --tag $lib --list /tmp/small_cases.json > /tmp/sb.log 2>&1
  python $R/tools/small_bench.py --parse /tmp/sb /tmp/small_cases.json | tee -a $OUT/small_bench.jsonl | head -10 | cut -c1-160
done
