#!/bin/bash
# round 4, GPU call P: family 7 v3 (weights straight into A-fragment registers): test, step A/B over the three routing modes, per-op profiles
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r04_p; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_igemm_cfgs_gpu.py -m gpu -q -x -k "test_conv_gemm_family or 49" -p no:cacheprovider > $O/pytest_family.log 2>&1; echo "pytest family rc $?" >> $O/summary.txt
for v in "IMAGEN_CONV_GEMM=0" "IMAGEN_CONV_GEMM=1" "IMAGEN_CONV_GEMM=2"; do
  env $v timeout 300 python tools/step_time.py --steps 60 --reps 3 --tag "$v" 2>/dev/null | tail -n 1 >> $O/step_ab.jsonl
done
for v in 0 2; do
( cd /tmp && rm -rf /tmp/gp$v && IMAGEN_CONV_GEMM=$v timeout 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/gp$v -- python $GRAFT_REPO_ROOT/tools/graph_profile.py run --steps 12 --plan-out /tmp/plan$v.json > $O/graph_profile_run$v.log 2>&1
  python $GRAFT_REPO_ROOT/tools/graph_profile.py analyze $(find /tmp/gp$v -name '*kernel_trace.csv' | head -n 1) /tmp/plan$v.json --top 80 --csv $O/graph_profile$v > $O/graph_profile$v.txt 2>&1 )
done
tail -n 2 $O/pytest_family.log; cat $O/summary.txt $O/step_ab.jsonl
