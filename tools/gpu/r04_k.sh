#!/bin/bash
# round 4, GPU call K: the ping-pong attention kernel: tests, kernel-level timing (mode 0 = online softmax as the same-box reference), counters, step A/B
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r04_k; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "test_attention" -p no:cacheprovider > $O/pytest_kernels.log 2>&1; echo "pytest kernels rc $?" >> $O/summary.txt
timeout 300 python tools/attn_bench.py --iters 20 --out $O/attn_bench.jsonl > $O/attn_bench.log 2>&1; echo "attn_bench rc $?" >> $O/summary.txt
cd /tmp
rm -rf /tmp/pmcA; timeout 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY --output-format csv -d /tmp/pmcA -- python $GRAFT_REPO_ROOT/tools/attn_bench.py --iters 2 --sites self-1024 > $O/pmc_attn.log 2>&1
f=$(find /tmp/pmcA -name '*counter_collection.csv' | head -n 1); [ -n "$f" ] && python $GRAFT_REPO_ROOT/tools/pmc_summary.py $f $O/pmc_attention.json > /dev/null 2>&1
rm -rf /tmp/ktA; timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ktA -- python $GRAFT_REPO_ROOT/tools/attn_bench.py --iters 5 --sites self-1024 self-256 > $O/trace_attn.log 2>&1
cp $(find /tmp/ktA -name '*kernel_stats.csv' | head -n 1) $O/attn_kernel_stats.csv 2>/dev/null
cd $GRAFT_REPO_ROOT
for v in "IMAGEN_ATTN_BOUNDED=0" "IMAGEN_ATTN_BOUNDED=1"; do
  env $v timeout 300 python tools/step_time.py --steps 60 --reps 3 --tag "$v" 2>/dev/null | tail -n 1 >> $O/step_ab.jsonl
done
tail -n 2 $O/pytest_kernels.log; cat $O/summary.txt $O/attn_bench.log $O/step_ab.jsonl; head -5 $O/attn_kernel_stats.csv | cut -c1-150
