#!/bin/bash
# Round 5, call E: ROWCHAIN v3 (K^ / V^T through the wave's LDS staging, gains in LDS, batched row copies): tests, per-launch times, phase timeline, step.
set -u
cd "$(dirname "$0")/../.."
R=$PWD
OUT=$R/gpurun_out/r05_e
mkdir -p $OUT
timeout 300 python -m pytest tests/test_rowchain_gpu.py -m gpu -q --tb=short -p no:cacheprovider 2>&1 | tail -n 3
echo "=== phase timeline"
IMAGEN_LIB_PATH=$R/imagen-pytorch_amd/libimagen_hip_trace.so timeout 300 python tools/chain_bench.py --trace --tag trace_v3 --reps 8 2>/dev/null | tail -n 1 | tee $OUT/chain_trace.json | cut -c1-3500
echo "=== step"
timeout 200 python tools/step_time.py --steps 60 --reps 3 --tag v3 2>/dev/null | tail -n 1 | tee -a $OUT/step.jsonl
cd /tmp && export TMPDIR=/tmp
echo "=== chain_bench"
rm -rf /tmp/cb_v3
timeout 240 rocprofv3 --kernel-trace --output-format csv -d /tmp/cb_v3 -- python $R/tools/chain_bench.py --tag v3 --list /tmp/cases_v3.json > /tmp/cb_v3.log 2>&1
python $R/tools/chain_bench.py --parse /tmp/cb_v3 /tmp/cases_v3.json | tee -a $OUT/chain_bench.jsonl | cut -c1-2500
