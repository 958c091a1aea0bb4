#!/bin/bash
# Round-3 GPU call F: -m gpu suite (unrolled 15x15 init conv, ADVICE fixes), step time, in-loop tile-configuration sweep.
set -u
cd "$(dirname "$0")/../.."
R=$PWD
OUT=$R/gpurun_out/r03_f
mkdir -p $OUT
rm -f $R/gpurun_out/parity_measured.json
timeout 420 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > $OUT/pytest.log 2>&1
tail -n 4 $OUT/pytest.log
cp $R/gpurun_out/parity_measured.json $OUT/ 2>/dev/null
timeout 240 python tools/step_time.py --tag "default" --lanes 6 2>$OUT/step.err | tee -a $OUT/step_times.jsonl
timeout 560 python tools/cfg_sweep.py --out $OUT/cfg_sweep.json --budget-s 400 2>$OUT/sweep.err | tee $OUT/cfg_sweep.log
tail -3 $OUT/sweep.err
