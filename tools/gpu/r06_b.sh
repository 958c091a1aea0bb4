#!/bin/bash
# Round 6, call B (NOT KEPT — the implementation is commit 1 of this round's history, reverted by the next): the GlobalContext gate finalised by the image's last tile
# (ticket + in-epilogue merge / squeeze MLP): its tests on hardware, the step with IMAGEN_GCA_EPILOGUE_FINAL = 0 / 1 / 2 interleaved on one box, an in-graph profile.
# Result: 8.12-8.17 / 8.34-8.38 / 8.50-8.53 ms per step pair — the finalisation costs a conv 4-11 us (write-through ack, ticket, acquire, cold loads: four dependent
# round trips at the END of a one-tile-per-CU kernel), more than the 16-workgroup launch it replaces.
#   gpurun --timeout 1500 -- 'bash tools/gpu/r06_b.sh'
set -u
cd "$(dirname "$0")/../.."
R=$PWD
OUT=$R/gpurun_out/r06_b
mkdir -p $OUT
echo "=== tests of the in-launch finalisation"
timeout 600 python -m pytest tests/test_igemm_cfgs_gpu.py tests/test_kernels_gpu.py tests/test_fusion_gpu.py -m gpu -q --tb=short -p no:cacheprovider -x -k "gca or global_context or resnet or every_cfg or conv_small or conv_gemm" > $OUT/pytest_gca.txt 2>&1; tail -n 4 $OUT/pytest_gca.txt | cut -c1-220
echo "=== step A/B (sequential + 6 lanes)"
for v in 2 0 1 2 0 1; do
  IMAGEN_GCA_EPILOGUE_FINAL=$v timeout 400 python tools/step_time.py --steps 60 --reps 3 --lanes 6 --tag gca_epi_final$v 2>/dev/null | tail -n 1 | tee -a $OUT/step_ab.jsonl
done
echo "=== whole-denoiser parity on the bench's own plans"
timeout 900 python -m pytest tests/test_model_gpu.py -m gpu -q --tb=short -p no:cacheprovider -k "rows16 or readme-unet1@64 or sample_vs_reference or graph" > $OUT/pytest_parity.txt 2>&1; tail -n 12 $OUT/pytest_parity.txt | cut -c1-220
cd /tmp && export TMPDIR=/tmp
echo "=== in-graph per-op profile"
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/gp -- python $R/tools/graph_profile.py run --steps 12 --plan-out /tmp/plan.json > /tmp/gp.log 2>&1
f=$(find /tmp/gp -name "*kernel_trace.csv" | head -1)
python $R/tools/graph_profile.py analyze $f /tmp/plan.json --top 60 --csv $OUT/graph_profile > $OUT/graph_profile.txt 2>&1
grep -A 16 "===" $OUT/graph_profile.txt | cut -c1-120
