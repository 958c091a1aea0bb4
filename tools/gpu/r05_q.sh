#!/bin/bash
# Round-5 call Q (session 2, 8 GPU-minutes left): where does the C5 denoiser's NULL row differ from its contract on hardware?
#   gpurun --timeout 420 -- 'bash tools/gpu/r05_q.sh'
# The plan interpreter predicts cond 1.01e-3 / null 0.92e-3 for BASELINE C5; the GPU measured 1.05e-3 / 1.06e-3.  The cond difference is the
# fp16 null value of the MFMA temporal attention (fixed in this commit, verified on the emulation); the null row's +12 % is not reproduced by
# the emulated kernels.  tools/op_audit.py runs every launch on the GPU and in the interpreter from identical inputs.
set -u
cd "$(dirname "$0")/../.."
R=$PWD
OUT=$R/gpurun_out/r05_q
mkdir -p $OUT
echo "=== op audit, C5 null row"
timeout 170 python tools/op_audit.py --config c5 --null --top 40 --json $OUT/op_audit_c5_null.json > $OUT/op_audit_c5_null.txt 2>&1; tail -n 60 $OUT/op_audit_c5_null.txt | cut -c1-200
echo "=== video tests that touch the temporal attention, the C5 denoiser"
timeout 170 python -m pytest tests/test_video_gpu.py -m gpu -q --tb=short -p no:cacheprovider -k "temporal_attention or forward_vs_oracle_c5" > $OUT/pytest_video.txt 2>&1; tail -n 8 $OUT/pytest_video.txt | cut -c1-250
echo "=== op audit, C5 cond row"
timeout 120 python tools/op_audit.py --config c5 --top 25 --json $OUT/op_audit_c5_cond.json > $OUT/op_audit_c5_cond.txt 2>&1; tail -n 30 $OUT/op_audit_c5_cond.txt | cut -c1-200
