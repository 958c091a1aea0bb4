#!/bin/bash
# round 4, GPU call B: conv_pro.hip with every unit of tile t + 2 requested as soon as its registers are free (a whole tile period in
# flight); the whole -m gpu suite on ABI 7; kernel-level and step-level A/B; in-graph per-op profile.
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04_b; mkdir -p $O
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1
timeout 300 python tools/stream_bench.py --out $O/stream_bench.jsonl > $O/stream_bench.log 2>&1; echo "stream_bench rc $?" >> $O/summary.txt
for v in "IMAGEN_CONV_PRO=0" "IMAGEN_CONV_PRO=1" "IMAGEN_CONV_PRO=2"; do
  env $v timeout 300 python tools/step_time.py --steps 60 --reps 3 --tag "$v" 2>/dev/null | tail -n 1 >> $O/step_ab.jsonl
done
( cd /tmp && timeout 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/gp -- python $GRAFT_REPO_ROOT/tools/graph_profile.py run --steps 12 --plan-out /tmp/plan.json > $GRAFT_REPO_ROOT/$O/graph_profile_run.log 2>&1
  python $GRAFT_REPO_ROOT/tools/graph_profile.py analyze $(find /tmp/gp -name '*kernel_trace.csv' | head -n 1) /tmp/plan.json --top 60 > $GRAFT_REPO_ROOT/$O/graph_profile.txt 2>&1 )
echo "graph_profile rc $?" >> $O/summary.txt
timeout 1500 python -m pytest tests -m gpu -q -x -p no:cacheprovider > $O/pytest_gpu.log 2>&1; echo "pytest rc $?" >> $O/summary.txt
cp gpurun_out/parity_measured.json $O/parity_measured.json 2>/dev/null
tail -n 5 $O/pytest_gpu.log; cat $O/summary.txt $O/stream_bench.log $O/step_ab.jsonl; head -n 30 $O/graph_profile.txt
