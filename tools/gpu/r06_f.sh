#!/bin/bash
# Round 6, call F: phase timeline of the GlobalContext gate derivation (gca_final_fast_body: s_memtime stamps of a -DGCA_TRACE variant library) inside
# gca_final_fast_kernel and gca_tail_kernel on the benchmark's shapes, cold operands.
#   gpurun --timeout 600 -- 'bash tools/gpu/r06_f.sh'
set -u
cd "$(dirname "$0")/../.."
R=$PWD
OUT=$R/gpurun_out/r06_f
mkdir -p $OUT
IMAGEN_LIB_PATH=$R/imagen-pytorch_amd/libimagen_hip_gcatrace.so timeout 400 python tools/gca_bench.py --trace --tag gca_trace > $OUT/gca_trace.json 2> $OUT/gca_trace.err
tail -n 3 $OUT/gca_trace.err
python - <<PY
import json
d = json.load(open("$OUT/gca_trace.json"))
for k, v in d["trace"].items():
    print(f"{k:44s} wgs={v['wgs']:5d} per_wg={v['per_wg_ticks']:9.0f} first_to_last={v['first_to_last_ticks']:9.0f}  phases={v['phase_ticks']}")
PY
