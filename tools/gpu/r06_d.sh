#!/bin/bash
# Round 6, call D: gca_tail with up to three passes of (h, res) rows staged in LDS by direct-to-LDS copies AHEAD of the gate derivation
# (libimagen_hip.so) against one pass (libimagen_hip_tailpf1.so = -DIMAGEN_TAIL_PF_MAX=1: what rounds 3-5 prefetched): the tail tests on
# hardware, the step interleaved on one box, the in-graph profile.
#   gpurun --timeout 1200 -- 'bash tools/gpu/r06_d.sh'
set -u
cd "$(dirname "$0")/../.."
R=$PWD
OUT=$R/gpurun_out/r06_d
mkdir -p $OUT
echo "=== tail / GlobalContext / fusion tests"
timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_fusion_gpu.py -m gpu -q --tb=short -p no:cacheprovider -x > $OUT/pytest_kernels.txt 2>&1; tail -n 3 $OUT/pytest_kernels.txt | cut -c1-220
echo "=== step A/B (sequential + 6 lanes)"
for lib in libimagen_hip.so libimagen_hip_tailpf1.so libimagen_hip.so libimagen_hip_tailpf1.so; do
  IMAGEN_LIB_PATH=$R/imagen-pytorch_amd/$lib timeout 400 python tools/step_time.py --steps 60 --reps 3 --lanes 6 --tag $lib 2>/dev/null | tail -n 1 | tee -a $OUT/step_ab.jsonl
done
echo "=== whole-denoiser parity on the bench's own plans + samplers"
timeout 900 python -m pytest tests/test_model_gpu.py -m gpu -q --tb=short -p no:cacheprovider -k "rows16 or sample_vs_reference or graph" > $OUT/pytest_parity.txt 2>&1; tail -n 8 $OUT/pytest_parity.txt | cut -c1-220
echo "=== C2: the 100-134 MB 1x1 GEMMs of its 8^2 / 16^2 levels on conv_small (default: SMALL_MAX_STREAM_MB 128) or back on families 0 / 7 (64)"
for mb in 128 64 128 64; do
  timeout 400 python -c "
import sys
import imagen_pytorch_amd.ops as o
o.SMALL_MAX_STREAM_MB = $mb
import bench
sys.argv = ['bench.py', '--config', 'c2', '--steps', '2', '--config-steps', '50']
bench.main()" 2>/dev/null | tail -n 1 | python -c "import sys, json; r = json.loads(sys.stdin.read()); print(json.dumps({'small_max_stream_mb': $mb, 'value': r['value'], 'ms_per_step': r.get('ms_per_sampling_step')}))" | tee -a $OUT/c2_small_stream_ab.jsonl
done
cd /tmp && export TMPDIR=/tmp
echo "=== in-graph per-op profile"
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/gp -- python $R/tools/graph_profile.py run --steps 12 --plan-out /tmp/plan.json > /tmp/gp.log 2>&1
f=$(find /tmp/gp -name "*kernel_trace.csv" | head -1)
python $R/tools/graph_profile.py analyze $f /tmp/plan.json --top 60 --csv $OUT/graph_profile > $OUT/graph_profile.txt 2>&1
grep -A 16 "===" $OUT/graph_profile.txt | cut -c1-120
grep "tail" $OUT/graph_profile.txt | head -12 | cut -c1-140
