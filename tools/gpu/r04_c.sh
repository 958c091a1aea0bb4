#!/bin/bash
# round 4, GPU call C: conv_pro.hip v3 (two row sets two tiles ahead, 64-cout instantiations for 96 -> 64 / 64 -> 64 @128^2); kernel- and
# step-level A/B; in-graph per-op profile; BASELINE's other configs (C2 / C4 / C5 legs of bench.py); the whole -m gpu suite.
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04_c; mkdir -p $O
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1
timeout 400 python tools/stream_bench.py --out $O/stream_bench.jsonl > $O/stream_bench.log 2>&1; echo "stream_bench rc $?" >> $O/summary.txt
for v in "IMAGEN_CONV_PRO=0" "IMAGEN_CONV_PRO=1" "IMAGEN_CONV_PRO=2"; do
  env $v timeout 300 python tools/step_time.py --steps 60 --reps 3 --tag "$v" 2>/dev/null | tail -n 1 >> $O/step_ab.jsonl
done
( cd /tmp && timeout 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/gp -- python $GRAFT_REPO_ROOT/tools/graph_profile.py run --steps 12 --plan-out /tmp/plan.json > $GRAFT_REPO_ROOT/$O/graph_profile_run.log 2>&1
  python $GRAFT_REPO_ROOT/tools/graph_profile.py analyze $(find /tmp/gp -name '*kernel_trace.csv' | head -n 1) /tmp/plan.json --top 60 > $GRAFT_REPO_ROOT/$O/graph_profile.txt 2>&1 )
echo "graph_profile rc $?" >> $O/summary.txt
for c in c2 c4 c5; do timeout 400 python bench.py --config $c --steps 2 --config-steps 50 2>$O/bench_$c.err | tail -n 1 > $O/bench_$c.json; echo "bench $c rc $?" >> $O/summary.txt; done
timeout 1500 python -m pytest tests -m gpu -q -x -p no:cacheprovider > $O/pytest_gpu.log 2>&1; echo "pytest rc $?" >> $O/summary.txt
cp gpurun_out/parity_measured.json $O/parity_measured.json 2>/dev/null
tail -n 5 $O/pytest_gpu.log; cat $O/summary.txt $O/stream_bench.log $O/step_ab.jsonl $O/bench_c*.json; head -n 40 $O/graph_profile.txt
