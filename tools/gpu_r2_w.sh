cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_model_gpu.py -m gpu -x -q 2>&1 | tail -4
python - <<'PY'
import json
d=json.load(open('gpurun_out/r02_parity_model.json'))
for k,v in d.items(): print(k, {kk:(round(vv,5) if isinstance(vv,float) else vv) for kk,vv in v.items() if kk not in ('taps','igemm_cfgs')})
PY
