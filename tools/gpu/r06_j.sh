#!/bin/bash
# Round 6, call J: Imagen-Video's causal temporal Conv1d as ONE (3 x 1)-tap igemm launch (ABI 10: pad_x1; engine3d.TEMPORAL_CONV_FUSED) against the three
# accumulating 1x1 GEMMs of rounds 1-5: the video tests on hardware (incl. the C5 whole-denoiser parity on three seeds), the C5 leg interleaved on one box.
#   gpurun --timeout 1800 -- 'bash tools/gpu/r06_j.sh'
set -u
cd "$(dirname "$0")/../.."
R=$PWD
OUT=$R/gpurun_out/r06_j
mkdir -p $OUT
timeout 1200 python -m pytest tests/test_video_gpu.py -m gpu -q --tb=short -p no:cacheprovider > $OUT/pytest_video.txt 2>&1; tail -n 12 $OUT/pytest_video.txt | cut -c1-220
for v in 1 0 1 0; do
  timeout 400 python -c "
import sys
import imagen_pytorch_amd.engine3d as e
e.TEMPORAL_CONV_FUSED = $v
import bench
sys.argv = ['bench.py', '--config', 'c5', '--steps', '2', '--config-steps', '50']
bench.main()" 2>$OUT/c5_$v.err | tail -n 1 | python -c "import sys, json; r = json.loads(sys.stdin.read()); print(json.dumps({'temporal_conv_fused': $v, 'value': r['value'], 'ms_per_step': r.get('ms_per_sampling_step')}))" | tee -a $OUT/c5_temporal_conv_ab.jsonl
done
