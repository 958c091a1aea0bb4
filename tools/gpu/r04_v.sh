#!/bin/bash
# round 4, GPU call V: two-phase GlobalContext finalisation of the wide blocks (C2's 512- / 1024-channel levels): kernel tests, C2 parity, C2 A/B
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r04_v; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1
timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py -m gpu -q -x -k "test_global_context or test_gca_tail or c2 or clamp_the_step" -p no:cacheprovider > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/summary.txt
for v in 0 1; do
  IMAGEN_GCA_FINAL_SPLIT=$v timeout 400 python bench.py --config c2 --steps 2 --config-steps 50 2>/dev/null | tail -n 1 > $O/bench_c2_split$v.json
  python - $O/bench_c2_split$v.json $v <<'PY' | tee -a $O/c2_ab.jsonl
import json, sys
d = json.loads(open(sys.argv[1]).read())
print(json.dumps(dict(gca_final_split=int(sys.argv[2]), images_per_s=d["value"], ms_per_sampling_step=d["ms_per_sampling_step"], launches=d["config"]["launches_per_step"])))
PY
done
grep "unet_forward\|passed\|failed" $O/pytest.log | cut -c1-200; cat $O/summary.txt
