#!/bin/bash
# round 4, GPU call D: hardware counters of conv_pro.hip / conv_stream.hip on the 64 -> 32 @256^2 launch (raw and with the prologue):
# where do the cycles go?  Separate rocprofv3 --pmc passes (kernel-trace only beside them), summaries per kernel symbol.
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r04_d; mkdir -p $O
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1
cd /tmp
ARGS="--iters 2 --shapes 256:32:32:32 128:64:32:64 --variants pro+post raw+ssq --cands stream pro fam0"
i=0
for pass in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" \
            "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU" \
            "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_MFMA SQ_INSTS_VALU_TRANS SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_WAVES" \
            "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
  i=$((i+1))
  rm -rf /tmp/pmc$i
  timeout 200 rocprofv3 --pmc $pass --output-format csv -d /tmp/pmc$i -- python $GRAFT_REPO_ROOT/tools/stream_bench.py $ARGS > $O/pass$i.log 2>&1
  f=$(find /tmp/pmc$i -name '*counter_collection.csv' | head -n 1)
  if [ -n "$f" ]; then python $GRAFT_REPO_ROOT/tools/pmc_summary.py $f $O/pmc_pass$i.json > /dev/null 2>&1; echo "pass $i ok: $pass" >> $O/summary.txt; else echo "pass $i FAILED: $pass" >> $O/summary.txt; tail -n 3 $O/pass$i.log >> $O/summary.txt; fi
done
# which launch is which: kernel trace of the same command (names + durations in dispatch order)
rm -rf /tmp/kt; timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -- python $GRAFT_REPO_ROOT/tools/stream_bench.py $ARGS > $O/trace.log 2>&1
cp $(find /tmp/kt -name '*kernel_stats.csv' | head -n 1) $O/kernel_stats.csv 2>/dev/null
cat $O/summary.txt
