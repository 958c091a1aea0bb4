#!/bin/bash
# Round 6, call E: C2's 512- / 1024-channel 3x3 Blocks (K = 4.6k - 13.8k on 8^2 / 16^2 maps: 15 % of its step on the wave-specialised kernel's 64 x 64
# tile, profiles/r06_c2_kernel_stats.csv) with their prologue as an ACT_PREP pass in front of an all-DMA conv (engine.ACT_PREP_MIN_COUT = 512 / 256)
# against the fused prologue (0, the default): the C2 leg, interleaved on one box; the C2-sized whole-denoiser parity cases with the winner.
#   gpurun --timeout 1500 -- 'bash tools/gpu/r06_e.sh'
set -u
cd "$(dirname "$0")/../.."
R=$PWD
OUT=$R/gpurun_out/r06_e
mkdir -p $OUT
for v in 0 512 256 0 512 256; do
  timeout 400 python -c "
import sys
import imagen_pytorch_amd.engine as e
e.ACT_PREP_MIN_COUT = $v
import bench
sys.argv = ['bench.py', '--config', 'c2', '--steps', '2', '--config-steps', '50']
bench.main()" 2>$OUT/c2_$v.err | tail -n 1 | python -c "import sys, json; r = json.loads(sys.stdin.read()); print(json.dumps({'act_prep_min_cout': $v, 'value': r['value'], 'ms_per_step': r.get('ms_per_sampling_step')}))" | tee -a $OUT/c2_act_prep_ab.jsonl
done
echo "=== C2-sized whole-denoiser parity with ACT_PREP_MIN_COUT = 512"
timeout 900 python -c "
import sys
import imagen_pytorch_amd.engine as e
e.ACT_PREP_MIN_COUT = 512
import pytest
sys.exit(pytest.main(['tests/test_model_gpu.py', '-m', 'gpu', '-q', '--tb=short', '-p', 'no:cacheprovider', '-k', 'c2-dim128']))" > $OUT/pytest_c2_parity.txt 2>&1; tail -n 8 $OUT/pytest_c2_parity.txt | cut -c1-220
