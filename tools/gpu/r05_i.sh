#!/bin/bash
# Round 5, call I: conv_small.hip per launch — every small-map case with the family and with the planner's previous pick (rocprofv3 durations,
# cold operands), and the s_memtime phase timeline of the family's kernel (a -DCS_TRACE variant library).
#   gpurun --timeout 600 -- 'bash tools/gpu/r05_i.sh'
set -u
cd "$(dirname "$0")/../.."
R=$PWD
OUT=$R/gpurun_out/r05_i
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for lib in libimagen_hip.so libimagen_hip_cs18.so; do
  rm -rf /tmp/sb
  IMAGEN_LIB_PATH=$R/imagen-pytorch_amd/$lib timeout 250 rocprofv3 --kernel-trace --output-format csv -d /tmp/sb -- python $R/tools/small_bench.py --tag $lib --list /tmp/small_cases.json > /tmp/sb.log 2>&1
  tail -n 1 /tmp/sb.log
  python $R/tools/small_bench.py --parse /tmp/sb /tmp/small_cases.json | tee -a $OUT/small_bench.jsonl
done
echo "=== phase timeline"
IMAGEN_LIB_PATH=$R/imagen-pytorch_amd/libimagen_hip_cstrace.so timeout 200 python $R/tools/small_bench.py --trace --tag trace 2>&1 | tail -n 1 | tee $OUT/phase_timeline.json | cut -c1-3000
