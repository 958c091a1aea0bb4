#!/bin/bash
# round 4, GPU call L: the tiled pointwise GEMM (family 7, conv_gemm.hip): family test on hardware, step A/B, in-graph per-op profile, whole-Unet parity
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r04_l; mkdir -p $O
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1
timeout 600 python -m pytest tests/test_igemm_cfgs_gpu.py -m gpu -q -x -k "test_conv_gemm_family or test_conv_pw_family or test_conv_stream_family" -p no:cacheprovider > $O/pytest_family.log 2>&1; echo "pytest family rc $?" >> $O/summary.txt
for v in "IMAGEN_CONV_GEMM=0" "IMAGEN_CONV_GEMM=1"; do
  env $v timeout 300 python tools/step_time.py --steps 60 --reps 3 --tag "$v" 2>/dev/null | tail -n 1 >> $O/step_ab.jsonl
done
( cd /tmp && timeout 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/gp -- python $GRAFT_REPO_ROOT/tools/graph_profile.py run --steps 12 --plan-out /tmp/plan.json > $O/graph_profile_run.log 2>&1
  python $GRAFT_REPO_ROOT/tools/graph_profile.py analyze $(find /tmp/gp -name '*kernel_trace.csv' | head -n 1) /tmp/plan.json --top 80 --csv $O/graph_profile > $O/graph_profile.txt 2>&1 )
echo "graph_profile rc $?" >> $O/summary.txt
timeout 900 python -m pytest tests/test_model_gpu.py tests/test_bench_shapes_gpu.py -m gpu -q -x -k "unet_forward_vs_oracle or bench_shapes" -p no:cacheprovider > $O/pytest_unet.log 2>&1; echo "pytest unet rc $?" >> $O/summary.txt
tail -n 3 $O/pytest_family.log; grep "unet_forward_vs_oracle\|passed\|failed" $O/pytest_unet.log | cut -c1-200; cat $O/summary.txt $O/step_ab.jsonl; grep -n "cfg49\|by kind" -A0 $O/graph_profile.txt | head -60
