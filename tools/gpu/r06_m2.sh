#!/bin/bash
# Call M, second part: the conv_big timeline (cold / warm / eight launches back to back), blocking instruction warm-up against -DCB_WARM_LATE.
set -u
cd "$(dirname "$0")/../.."
R=$PWD
OUT=$R/gpurun_out/r06_m
mkdir -p $OUT
L=$R/imagen-pytorch_amd
for v in cbtrace cbtrace_wl; do
  T="env IMAGEN_LIB_PATH=$L/libimagen_hip_$v.so timeout 300 python tools/conv_bench.py --trace --iters 12"
  $T --tag ${v}_64 --shapes 192:128:64 128:128:64 --cands big:3 > $OUT/${v}_64.json 2> $OUT/trace.err
  $T --tag ${v}_32 --shapes 384:256:32 256:256:32 --cands big:2 > $OUT/${v}_32.json 2>> $OUT/trace.err
done
tail -n 3 $OUT/trace.err
python - <<PY
import json,glob
for f in sorted(glob.glob("$OUT/cbtrace*.json")):
    d=json.load(open(f))
    for k,v in d["trace"].items(): print(d["tag"], k, json.dumps(v))
PY
