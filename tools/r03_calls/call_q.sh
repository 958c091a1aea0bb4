#!/bin/bash
# Round-3 GPU call Q: timing ablations of conv_big (bench-only -DCB_ABLATE library) on two shapes + the in-graph profile with the family on.
set -u
cd "$(dirname "$0")/../.."
R=$PWD
OUT=$R/gpurun_out/r03_q
mkdir -p $OUT
IMAGEN_LIB_PATH=$R/imagen-pytorch_amd/libimagen_hip_ablate.so timeout 200 python tools/conv_bench.py --shapes 192:128:64 384:256:32 --cands big:0 big:1 --ablate 1 2 3 4 7 8 12 15 16 31 --out $OUT/ablate.jsonl 2>$OUT/bench.err | cut -c1-200
cd /tmp && export TMPDIR=/tmp
timeout 240 rocprofv3 --kernel-trace --output-format csv -d /tmp/gp -- python $R/tools/graph_profile.py run --steps 12 --plan-out /tmp/plan.json > /tmp/gp.log 2>&1
f=$(find /tmp/gp -name "*kernel_trace.csv" | head -1)
python $R/tools/graph_profile.py analyze $f /tmp/plan.json --top 60 --csv $OUT/graph_profile > $OUT/graph_profile.txt 2>&1
grep -E "=== stage|k3 @64\]|k3 @32\]|prep" $OUT/graph_profile.txt | head -40 | cut -c1-150
