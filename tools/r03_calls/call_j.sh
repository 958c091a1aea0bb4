#!/bin/bash
# Round-3 GPU call J: per-launch times of the step with the wave-specialised kernel's work ablated (27 = no producer work, no consumer
# compute, no stores: launch + persistent loop + barriers + epilogue operand loads) beside the default, same box.
set -u
cd "$(dirname "$0")/../.."
R=$PWD
OUT=$R/gpurun_out/r03_j
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for v in 0 27 19; do
  rm -rf /tmp/gp$v
  IMAGEN_ABLATE_IGEMM_DBG=$v timeout 240 rocprofv3 --kernel-trace --output-format csv -d /tmp/gp$v -- python $R/tools/graph_profile.py run --steps 10 --plan-out /tmp/plan$v.json > /tmp/gp$v.log 2>&1
  f=$(find /tmp/gp$v -name "*kernel_trace.csv" | head -1)
  python $R/tools/graph_profile.py analyze $f /tmp/plan$v.json --top 40 --csv $OUT/graph_profile_dbg$v > $OUT/graph_profile_dbg$v.txt 2>&1
  grep -A 4 "===" $OUT/graph_profile_dbg$v.txt | cut -c1-120
done
