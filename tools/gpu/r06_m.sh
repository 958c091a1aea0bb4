#!/bin/bash
# Call M: conv_big between launches.  (1) phase timeline + eight launches back to back with the constant-clock stamps (what the device sees between
# one launch's last workgroup and the next one's first), blocking instruction warm-up (product) against the warm-up as non-blocking loads behind the
# first stage's copies (-DCB_WARM_LATE); (2) step A/B of both libraries and of ops.BIG_PICKS (3, 2) against (0, 1), interleaved on one box.
#   gpurun --timeout 1500 -- 'bash tools/gpu/r06_m.sh'
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
echo "=== kernel-level timing: product / warm-late library"
for lib in libimagen_hip.so libimagen_hip_cbwl.so; do
  IMAGEN_LIB_PATH=$L/$lib timeout 300 python tools/conv_bench.py --iters 40 --shapes 192:128:64 128:128:64 --cands big:3 big:0 2>&1 | grep shape | sed "s/^/$lib /"
  IMAGEN_LIB_PATH=$L/$lib timeout 300 python tools/conv_bench.py --iters 40 --shapes 384:256:32 256:256:32 --cands big:2 big:1 2>&1 | grep shape | sed "s/^/$lib /"
done | tee $OUT/kernel_ab.txt
echo "=== step A/B"
for r in 1 2; do
  for v in base wl picks01 picks01wl; do
    lib=libimagen_hip.so; picks="(3, 2)"
    case $v in wl|picks01wl) lib=libimagen_hip_cbwl.so;; esac
    case $v in picks01|picks01wl) picks="(0, 1)";; esac
    IMAGEN_LIB_PATH=$L/$lib timeout 300 python -c "
import sys
import imagen_pytorch_amd.ops as o
o.BIG_PICKS = $picks
sys.argv = ['step_time.py', '--steps', '60', '--reps', '3', '--tag', '$v']
import runpy
runpy.run_path('tools/step_time.py', run_name='__main__')" 2>>$OUT/step.err | tail -n 1 | tee -a $OUT/step_ab.jsonl | cut -c1-200
  done
done
