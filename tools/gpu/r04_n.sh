#!/bin/bash
# round 4, GPU call N: family 7 after routing the wide-cout launches to it: step A/B + in-graph per-op profile
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r04_n; mkdir -p $O
export TMPDIR=/tmp
for v in "IMAGEN_CONV_GEMM=0" "IMAGEN_CONV_GEMM=1"; do
  env $v timeout 300 python tools/step_time.py --steps 60 --reps 3 --tag "$v" 2>/dev/null | tail -n 1 >> $O/step_ab.jsonl
done
( cd /tmp && timeout 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/gp -- python $GRAFT_REPO_ROOT/tools/graph_profile.py run --steps 12 --plan-out /tmp/plan.json > $O/graph_profile_run.log 2>&1
  python $GRAFT_REPO_ROOT/tools/graph_profile.py analyze $(find /tmp/gp -name '*kernel_trace.csv' | head -n 1) /tmp/plan.json --top 80 --csv $O/graph_profile > $O/graph_profile.txt 2>&1 )
echo "graph_profile rc $?" >> $O/summary.txt
cat $O/summary.txt $O/step_ab.jsonl; grep "cfg49" $O/graph_profile.txt | head -40
