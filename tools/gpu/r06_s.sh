#!/bin/bash
# Call S: the per-op in-graph profile of a batch-24 request (a merged batch) beside the batch-8 one: which launches scale worse than their work?
set -u
cd "$(dirname "$0")/../.."
R=$PWD
OUT=$R/gpurun_out/r06_s
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for b in 8 24; do
  rm -rf /tmp/gp$b
  timeout 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/gp$b -- python $R/tools/graph_profile.py run --batch $b --steps 10 --plan-out /tmp/plan$b.json > /tmp/gp$b.log 2>&1
  f=$(find /tmp/gp$b -name "*kernel_trace.csv" | head -1)
  python $R/tools/graph_profile.py analyze $f /tmp/plan$b.json --top 5 --csv $OUT/b$b > $OUT/b$b.txt 2>&1
  grep "replay period" $OUT/b$b.txt | cut -c1-140
done
python - <<PY
import csv
for st in (1, 2):
    a = list(csv.DictReader(open("$OUT/b8.stage%d.csv" % st))); b = list(csv.DictReader(open("$OUT/b24.stage%d.csv" % st)))
    print("stage", st, len(a), len(b), "sum8", round(sum(float(r["kernel_us"]) for r in a)), "sum24", round(sum(float(r["kernel_us"]) for r in b)))
    if len(a) != len(b):
        la = [r["label"] for r in a]; lb = [r["label"] for r in b]
        print("  only in 8:", [l for l in la if l not in lb][:12]); print("  only in 24:", [l for l in lb if l not in la][:12])
    db = {r["label"]: r for r in b}
    rows = []
    for r in a:
        q = db.get(r["label"])
        if q: rows.append((float(q["kernel_us"]) - 3 * float(r["kernel_us"]), float(r["kernel_us"]), float(q["kernel_us"]), r["label"], r["desc"][:60], q["desc"][:60]))
    rows.sort(reverse=True)
    for x in rows[:14]: print("  excess %.1f us: %.1f -> %.1f  %s | %s | %s" % x)
PY
