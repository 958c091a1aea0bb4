#!/bin/bash
# round 4, last GPU call: does routing EVERYTHING the GEMM family accepts (IMAGEN_CONV_GEMM=2) pay on C2 / C5 (many deep 1x1 layers)?
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r04_zz; mkdir -p $O
for c in c2 c5; do for m in 1 2; do
  IMAGEN_CONV_GEMM=$m timeout 200 python bench.py --config $c --steps 1 --warmup 1 --config-steps 30 2>/dev/null | tail -n 1 > $O/b.json
  python - $O/b.json $c $m <<'PY' | tee -a $O/gemm_mode_ab.jsonl
import json, sys
d = json.loads(open(sys.argv[1]).read())
print(json.dumps(dict(config=sys.argv[2], conv_gemm=int(sys.argv[3]), value=d["value"], ms_per_sampling_step=d["ms_per_sampling_step"])))
PY
done; done
