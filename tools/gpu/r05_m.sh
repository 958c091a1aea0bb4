#!/bin/bash
# Round 5, call M: conv_small.hip with the phase-shifted K walk (pixel tiles that share a weight slab start at 0, 1/4, 1/2, 3/4 of their slice)
# against the same kernel walking from the head (libimagen_hip_csph1.so): per launch, phase timeline, the step; and the family in lanes mode.
#   gpurun --timeout 900 -- 'bash tools/gpu/r05_m.sh'
set -u
cd "$(dirname "$0")/../.."
R=$PWD
OUT=$R/gpurun_out/r05_m
mkdir -p $OUT
echo "=== conv_small tests"
timeout 300 python -m pytest tests/test_igemm_cfgs_gpu.py -m gpu -q --tb=short -p no:cacheprovider -x -k "conv_small or every_cfg" > $OUT/pytest_cfgs.txt 2>&1; tail -n 3 $OUT/pytest_cfgs.txt | cut -c1-220
echo "=== step A/B (sequential + 6 lanes)"
for v in "1 libimagen_hip.so" "1 libimagen_hip_csph1.so" "0 libimagen_hip.so" "1 libimagen_hip.so" "1 libimagen_hip_csph1.so" "0 libimagen_hip.so"; do
  set -- $v
  IMAGEN_CONV_SMALL=$1 IMAGEN_LIB_PATH=$R/imagen-pytorch_amd/$2 timeout 300 python tools/step_time.py --steps 60 --reps 3 --lanes 6 --tag small$1_$2 2>/dev/null | tail -n 1 | tee -a $OUT/step_ab.jsonl
done
cd /tmp && export TMPDIR=/tmp
echo "=== per launch"
for lib in libimagen_hip.so libimagen_hip_csph1.so; do
  rm -rf /tmp/sb
  IMAGEN_LIB_PATH=$R/imagen-pytorch_amd/$lib timeout 250 rocprofv3 --kernel-trace --output-format csv -d /tmp/sb -- python $R/tools/small_bench.py --tag $lib --list /tmp/small_cases.json > /tmp/sb.log 2>&1
  tail -n 1 /tmp/sb.log
  python $R/tools/small_bench.py --parse /tmp/sb /tmp/small_cases.json | tee -a $OUT/small_bench.jsonl | cut -c1-200
done
echo "=== phase timeline"
IMAGEN_LIB_PATH=$R/imagen-pytorch_amd/libimagen_hip_cstrace.so timeout 200 python $R/tools/small_bench.py --trace --tag trace_v5 2>&1 | tail -n 1 > $OUT/phase_timeline.json
python - <<'PY'
import json
d=json.load(open("/root/repo/gpurun_out/r05_m/phase_timeline.json"))
for k,v in d["trace"].items(): print(f"{k:28s} {v['tile']} wgs={v['wgs']:5d} {v['phase_cycles']} per_wg={v['per_wg_cycles']}")
PY
