#!/bin/bash
# Round-5 call R (session 2): the video path after the split-precision output stage (engine3d.SPLIT_OUTPUT_STAGE) — every video test with its
# measured figures (-s), the C5 throughput leg.
#   gpurun --timeout 300 -- 'bash tools/gpu/r05_r.sh'
set -u
cd "$(dirname "$0")/../.."
R=$PWD
OUT=$R/gpurun_out/r05_r
mkdir -p $OUT
echo "=== tests/test_video_gpu.py"
timeout 200 python -m pytest tests/test_video_gpu.py -m gpu -q -s --tb=short -p no:cacheprovider > $OUT/pytest_video.txt 2>&1
grep -E "video options|c5 |edm|errs|passed|failed|Error|assert" $OUT/pytest_video.txt | cut -c1-250 | tail -n 30
echo "=== C5 leg"
timeout 90 python bench.py --config c5 --steps 2 --config-steps 50 2>$OUT/bench_c5.err | tail -n 1 > $OUT/bench_c5.json; cut -c1-400 $OUT/bench_c5.json
