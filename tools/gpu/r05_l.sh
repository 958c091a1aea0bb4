#!/bin/bash
# Round 5, call L: conv_small.hip with the wave's weight slab touched into L2 behind the first ring fill — per launch, phase timeline, step A/B.
#   gpurun --timeout 900 -- 'bash tools/gpu/r05_l.sh'
set -u
cd "$(dirname "$0")/../.."
R=$PWD
OUT=$R/gpurun_out/r05_l
mkdir -p $OUT
echo "=== conv_small tests"
timeout 300 python -m pytest tests/test_igemm_cfgs_gpu.py -m gpu -q --tb=short -p no:cacheprovider -x -k "conv_small or every_cfg" > $OUT/pytest_cfgs.txt 2>&1; tail -n 3 $OUT/pytest_cfgs.txt | cut -c1-220
echo "=== step A/B"
for v in 1 0 1 0; do
  IMAGEN_CONV_SMALL=$v timeout 200 python tools/step_time.py --steps 60 --reps 3 --tag small$v 2>/dev/null | tail -n 1 | tee -a $OUT/step_ab.jsonl
done
cd /tmp && export TMPDIR=/tmp
echo "=== per launch"
rm -rf /tmp/sb
timeout 250 rocprofv3 --kernel-trace --output-format csv -d /tmp/sb -- python $R/tools/small_bench.py --tag v4 --list /tmp/small_cases.json > /tmp/sb.log 2>&1
tail -n 1 /tmp/sb.log
python $R/tools/small_bench.py --parse /tmp/sb /tmp/small_cases.json | tee -a $OUT/small_bench.jsonl
echo "=== phase timeline"
IMAGEN_LIB_PATH=$R/imagen-pytorch_amd/libimagen_hip_cstrace.so timeout 200 python $R/tools/small_bench.py --trace --tag trace_v4 2>&1 | tail -n 1 > $OUT/phase_timeline.json
python - <<'PY'
import json
d=json.load(open("/root/repo/gpurun_out/r05_l/phase_timeline.json"))
for k,v in d["trace"].items(): print(f"{k:28s} {v['tile']} wgs={v['wgs']:5d} {v['phase_cycles']} per_wg={v['per_wg_cycles']}")
PY
