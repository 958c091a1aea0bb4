#!/bin/bash
# Call N: the MFMA temporal attention with the next 32-row query block requested while the current one is multiplied (q_scale * scale folded into
# K^: 244 -> 216 registers) against round 5's kernel (libimagen_hip_taold.so = HEAD~'s temporal.hip): video tests on hardware, the C5 leg
# interleaved on one box, kernel table of C5.
#   gpurun --timeout 1800 -- 'bash tools/gpu/r06_n.sh'
set -u
cd "$(dirname "$0")/../.."
R=$PWD
OUT=$R/gpurun_out/r06_n
mkdir -p $OUT
L=$R/imagen-pytorch_amd
timeout 1200 python -m pytest tests/test_video_gpu.py -m gpu -q --tb=short -p no:cacheprovider > $OUT/pytest_video.txt 2>&1; tail -n 12 $OUT/pytest_video.txt | cut -c1-220
for lib in libimagen_hip.so libimagen_hip_taold.so libimagen_hip.so libimagen_hip_taold.so; do
  IMAGEN_LIB_PATH=$L/$lib timeout 400 python bench.py --config c5 --steps 2 --config-steps 50 2>$OUT/c5.err | tail -n 1 | python -c "import sys, json; r = json.loads(sys.stdin.read()); print(json.dumps({'lib': '$lib', 'value': r['value'], 'ms_per_step': r.get('ms_per_sampling_step')}))" | tee -a $OUT/c5_temporal_attention_ab.jsonl
done
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c5 -- python $R/bench.py --config c5 --steps 1 --config-steps 40 > /tmp/prof_c5.log 2>&1
f=$(find /tmp/prof_c5 -name "*kernel_stats.csv" | head -1); cp $f $OUT/c5_kernel_stats.csv; head -8 $f | cut -c1-170
