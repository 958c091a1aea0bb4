#!/bin/bash
# Call O: the cross-attention chain's softmax weights as fp16 hi + lo pairs (four more MFMAs per key tile) against the bare fp16 P of round 5
# (libimagen_hip_p16.so = -DROWCHAIN_P16): chain tests, whole-denoiser parity (image unets + C5 on three seeds), C3 step and C5 leg interleaved.
#   gpurun --timeout 2400 -- 'bash tools/gpu/r06_o.sh'
set -u
cd "$(dirname "$0")/../.."
R=$PWD
OUT=$R/gpurun_out/r06_o
mkdir -p $OUT
L=$R/imagen-pytorch_amd
rm -f $R/gpurun_out/parity_measured.json
timeout 1500 python -m pytest tests/test_rowchain_gpu.py tests/test_model_gpu.py tests/test_video_gpu.py -m gpu -q --tb=short -p no:cacheprovider -k "rowchain or forward_vs_oracle or chain" > $OUT/pytest_parity.txt 2>&1; tail -n 22 $OUT/pytest_parity.txt | cut -c1-200
cp $R/gpurun_out/parity_measured.json $OUT/parity.json 2>/dev/null
for r in 1 2; do
  for lib in libimagen_hip.so libimagen_hip_p16.so; do
    IMAGEN_LIB_PATH=$L/$lib timeout 300 python tools/step_time.py --steps 60 --reps 3 --tag $lib 2>>$OUT/step.err | tail -n 1 | tee -a $OUT/step_ab.jsonl | cut -c1-160
  done
done
for lib in libimagen_hip.so libimagen_hip_p16.so libimagen_hip.so libimagen_hip_p16.so; do
  IMAGEN_LIB_PATH=$L/$lib timeout 400 python bench.py --config c5 --steps 2 --config-steps 50 2>$OUT/c5.err | tail -n 1 | python -c "import sys, json; r = json.loads(sys.stdin.read()); print(json.dumps({'lib': '$lib', 'value': r['value'], 'ms_per_step': r.get('ms_per_sampling_step')}))" | tee -a $OUT/c5_ab.jsonl
done
