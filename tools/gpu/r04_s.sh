#!/bin/bash
# round 4, GPU call S: requests in flight (lanes) A/B at 250 steps per stage (same kernels, shorter schedule)
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r04_s; mkdir -p $O
for L in 6 9 12; do
  timeout 400 python bench.py --lanes $L --steps $L --warmup $L --timesteps 250 --no-cpu-baseline --no-roofline 2>/dev/null | tail -n 1 > $O/lanes_$L.json
  python - $O/lanes_$L.json $L <<'PY' | tee -a $O/lanes_ab.jsonl
import json, sys
d = json.loads(open(sys.argv[1]).read())
print(json.dumps(dict(lanes=int(sys.argv[2]), value=d["value"], ms_per_step=d["ms_per_step"], sequential=d["sequential"]["value"])))
PY
done
