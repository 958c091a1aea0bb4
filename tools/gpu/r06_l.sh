#!/bin/bash
# Call L: phase timeline of conv_big_kernel (-DCB_TRACE variant library, tools/conv_bench.py --trace) on the benchmark's shapes, cold and warm.
#   gpurun --timeout 900 -- 'bash tools/gpu/r06_l.sh'
set -u
cd "$(dirname "$0")/../.."
R=$PWD
OUT=$R/gpurun_out/r06_l
mkdir -p $OUT
T="env IMAGEN_LIB_PATH=$R/imagen-pytorch_amd/libimagen_hip_cbtrace.so timeout 300 python tools/conv_bench.py --trace --iters 20"
$T --tag cb_trace_64 --shapes 192:128:64 128:128:64 --cands big:3 big:0 > $OUT/cb_timeline_64.json 2> $OUT/cb_timeline.err
$T --tag cb_trace_32 --shapes 384:256:32 256:256:32 --cands big:2 big:1 > $OUT/cb_timeline_32.json 2>> $OUT/cb_timeline.err
$T --tag cb_trace_64_gca --gca --shapes 128:128:64 --cands big:3 > $OUT/cb_timeline_64_gca.json 2>> $OUT/cb_timeline.err
tail -n 3 $OUT/cb_timeline.err
python - <<PY
import json,glob
for f in sorted(glob.glob("$OUT/cb_timeline_*.json")):
    d=json.load(open(f))
    for k,v in d["trace"].items(): print(k, json.dumps(v))
PY
echo "=== plain timing of the same shapes (product library)"
timeout 300 python tools/conv_bench.py --iters 40 --shapes 192:128:64 128:128:64 --cands big:3 big:0 --out $OUT/cb_bench.jsonl 2>&1 | tail -n 4
timeout 300 python tools/conv_bench.py --iters 40 --shapes 384:256:32 256:256:32 --cands big:2 big:1 --out $OUT/cb_bench.jsonl 2>&1 | tail -n 4
