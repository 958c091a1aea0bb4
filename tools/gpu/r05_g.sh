#!/bin/bash
# Round 5, call G: the bench line again with its PMC passes (the join of call `measure` tripped over ~2500 upload copies in front of the first replay:
# tools/graph_profile.py find_period), the PMC passes joined with the plan, a driver-sized run (--steps 20).
set -u
cd "$(dirname "$0")/../.."
R=$PWD
OUT=$R/gpurun_out/r05_g
mkdir -p $OUT
echo "=== bench (driver-sized: --steps 20 --warmup 6)"
timeout 1200 python bench.py --steps 20 --warmup 6 > $OUT/bench_n1.json 2> $OUT/bench_n1.err
tail -n 8 $OUT/bench_n1.err | cut -c1-200; cut -c1-600 $OUT/bench_n1.json
cd /tmp && export TMPDIR=/tmp
echo "=== PMC passes joined with the plan"
for c in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU GRBM_GUI_ACTIVE"; do
  tag=$(echo $c | cut -d' ' -f1)
  rm -rf /tmp/pmc_$tag
  timeout 300 rocprofv3 --pmc $c --output-format csv -d /tmp/pmc_$tag -- python $R/tools/graph_profile.py run --steps 5 --plan-out /tmp/plan_$tag.json > /tmp/pmc_$tag.log 2>&1
  f=$(find /tmp/pmc_$tag -name "*counter_collection.csv" | head -1)
  python $R/tools/graph_profile.py pmc $f /tmp/plan_$tag.json $OUT/pmc_$tag.json | cut -c1-300
done
