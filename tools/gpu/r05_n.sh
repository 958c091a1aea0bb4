#!/bin/bash
# Round 5, call N: the phase-shifted K walk in rowchain.hip (row tiles that stream the same weights start at different quarters of their slice)
# against the same kernels walking from the head (libimagen_hip_rcph1.so): the chain tests, per-launch durations, the step.
#   gpurun --timeout 900 -- 'bash tools/gpu/r05_n.sh'
set -u
cd "$(dirname "$0")/../.."
R=$PWD
OUT=$R/gpurun_out/r05_n
mkdir -p $OUT
echo "=== rowchain tests"
timeout 300 python -m pytest tests/test_rowchain_gpu.py -m gpu -q --tb=short -p no:cacheprovider > $OUT/pytest_rowchain.txt 2>&1; tail -n 3 $OUT/pytest_rowchain.txt | cut -c1-220
echo "=== step A/B"
for lib in libimagen_hip.so libimagen_hip_rcph1.so libimagen_hip.so libimagen_hip_rcph1.so; do
  IMAGEN_LIB_PATH=$R/imagen-pytorch_amd/$lib timeout 300 python tools/step_time.py --steps 60 --reps 3 --tag $lib 2>/dev/null | tail -n 1 | tee -a $OUT/step_ab.jsonl
done
cd /tmp && export TMPDIR=/tmp
echo "=== per launch"
for lib in libimagen_hip.so libimagen_hip_rcph1.so; do
  rm -rf /tmp/cb
  IMAGEN_LIB_PATH=$R/imagen-pytorch_amd/$lib timeout 250 rocprofv3 --kernel-trace --output-format csv -d /tmp/cb -- python $R/tools/chain_bench.py --tag $lib --list /tmp/chain_cases.json > /tmp/cb.log 2>&1
  tail -n 1 /tmp/cb.log
  python $R/tools/chain_bench.py --parse /tmp/cb /tmp/chain_cases.json | tee -a $OUT/chain_bench.jsonl | cut -c1-1500
done
