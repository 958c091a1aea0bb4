#!/bin/bash
# Round 6, call C: the whole -m gpu suite at HEAD (after the two reverted experiments), the bench line with the loaded-latency probe and the socclk fields,
# kernel tables of BASELINE's C2 and C4 configurations (rocprofv3 --kernel-trace --stats over their bench legs).
#   gpurun --timeout 2400 -- 'bash tools/gpu/r06_c.sh'
set -u
cd "$(dirname "$0")/../.."
R=$PWD
OUT=$R/gpurun_out/r06_c
mkdir -p $OUT
rm -f $R/gpurun_out/parity_measured.json
echo "=== pytest -m gpu"
timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > $OUT/pytest_gpu.txt 2>&1
tail -n 30 $OUT/pytest_gpu.txt | cut -c1-200
cp $R/gpurun_out/parity_measured.json $OUT/parity.json 2>/dev/null
echo "=== bench (6 lanes, 1000 steps; no PMC passes, no CPU baseline: the calibration block is what this call is about)"
timeout 900 python bench.py --steps 6 --warmup 6 --no-cpu-baseline --no-pmc > $OUT/bench_n1.json 2> $OUT/bench_n1.err
tail -n 4 $OUT/bench_n1.err; python - <<PY
import json
r = json.load(open("$OUT/bench_n1.json"))
print({k: r[k] for k in ("value", "ms_per_step")}, r["sequential"]["value"], r["sequential"].get("box"), r["sequential"].get("in_graph_step_ms"))
print(r["calibration"])
print(r["roofline"])
PY
cd /tmp && export TMPDIR=/tmp
for c in c2 c4; do
  echo "=== kernel table of $c"
  rm -rf /tmp/prof_$c
  timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$c -- python $R/bench.py --config $c --steps 2 --warmup 1 --config-steps 40 > $OUT/bench_${c}_profiled.json 2> $OUT/bench_${c}_profiled.err
  f=$(find /tmp/prof_$c -name '*kernel_stats.csv' | head -n 1); cp $f $OUT/${c}_kernel_stats.csv
  head -n 14 $OUT/${c}_kernel_stats.csv | cut -c1-200
  tail -n 1 $OUT/bench_${c}_profiled.json | cut -c1-500
done
