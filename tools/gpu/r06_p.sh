#!/bin/bash
# Call P: probe of a fork / join inside the step graph (res_conv beside block1 -> block2 -> GlobalContext), tools/probe/branch_overlap.py.
set -u
cd "$(dirname "$0")/../.."
OUT=$PWD/gpurun_out/r06_p
mkdir -p $OUT
timeout 600 python tools/probe/branch_overlap.py > $OUT/branch_overlap.json 2> $OUT/err.txt; tail -n 3 $OUT/err.txt
python - <<PY
import json
d=json.load(open("$OUT/branch_overlap.json"))
for k,v in d.items():
    print(k, "seq", v["seq_us"], "fork/join", v["fork_join_us"])
    for r in v["blocks"]: print("   ", r)
PY
