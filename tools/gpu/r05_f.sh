#!/bin/bash
# Round 5, call F: why the PMC join fails (keep the counter CSV + the plan).
set -u
cd "$(dirname "$0")/../.."
R=$PWD
OUT=$R/gpurun_out/r05_f
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc_f
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pmc_f -- python $R/tools/graph_profile.py run --steps 5 --plan-out $OUT/plan.json > $OUT/pmc.log 2>&1
f=$(find /tmp/pmc_f -name "*counter_collection.csv" | head -1)
cut -d, -f1-12 $f | head -3
python - "$f" <<'PY' > $OUT/names.txt
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
print(len(rows), list(rows[0].keys()))
seen = []
for r in rows:
    seen.append((int(r.get("Dispatch_Id", 0)), r["Kernel_Name"][:90]))
seen = sorted(set(seen))
for d, n in seen: print(d, n)
PY
head -2 $OUT/names.txt | cut -c1-300; wc -l $OUT/names.txt
