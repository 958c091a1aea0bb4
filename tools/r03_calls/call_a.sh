#!/bin/bash
# Round-3 GPU call A: (1) the whole -m gpu suite with the six formerly gated tests ungated and the C5-size Unet3D parity test,
# (2) the four prepared igemm libraries timed with tools/step_time.py on ONE box, (3) the round's baseline in-graph profile.
#     gpurun --timeout 900 -- 'bash tools/r03_calls/call_a.sh'
set -u
cd "$(dirname "$0")/../.."
R=$PWD
OUT=$R/gpurun_out/r03_a
mkdir -p $OUT
rm -f $R/gpurun_out/parity_measured.json
timeout 420 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > $OUT/pytest.log 2>&1
tail -n 45 $OUT/pytest.log
cp $R/gpurun_out/parity_measured.json $OUT/ 2>/dev/null
for v in "" remat remat8 onewg; do
  lib=$R/imagen-pytorch_amd/libimagen_hip${v:+_$v}.so
  [ -f $lib ] && IMAGEN_LIB_PATH=$lib timeout 150 python tools/step_time.py --tag "lib=${v:-default}" 2>$OUT/step_${v:-default}.err | tee -a $OUT/step_times.jsonl
done
timeout 150 python tools/step_time.py --tag "lib=default again" 2>>$OUT/step_default.err | tee -a $OUT/step_times.jsonl
cd /tmp && export TMPDIR=/tmp
timeout 240 rocprofv3 --kernel-trace --output-format csv -d /tmp/gp -- python $R/tools/graph_profile.py run --steps 12 --plan-out /tmp/plan.json > /tmp/gp.log 2>&1
f=$(find /tmp/gp -name "*kernel_trace.csv" | head -1)
python $R/tools/graph_profile.py analyze $f /tmp/plan.json --top 60 --csv $OUT/graph_profile > $OUT/graph_profile.txt 2>&1
grep -A 8 "===" $OUT/graph_profile.txt | cut -c1-120
