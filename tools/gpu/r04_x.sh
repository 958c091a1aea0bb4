#!/bin/bash
# round 4, GPU call X: temporal attention on the matrix pipe: video tests, C5 against the previous commit's kernel (throw-away library, same box)
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r04_x; mkdir -p $O
timeout 900 python -m pytest tests/test_video_gpu.py -m gpu -q -p no:cacheprovider > $O/pytest_video.log 2>&1; echo "pytest video rc $?" >> $O/summary.txt
for lib in abl_oldtemporal hip; do
  IMAGEN_LIB_PATH=$GRAFT_REPO_ROOT/imagen-pytorch_amd/libimagen_$lib.so timeout 400 python bench.py --config c5 --steps 2 --config-steps 50 2>/dev/null | tail -n 1 > $O/bench_c5_$lib.json
  python - $O/bench_c5_$lib.json $lib <<'PY' | tee -a $O/c5_ab.jsonl
import json, sys
d = json.loads(open(sys.argv[1]).read())
print(json.dumps(dict(lib=sys.argv[2], clips_per_s=d["value"], ms_per_sampling_step=d["ms_per_sampling_step"], path_frac_of_mfma_peak=d.get("path_frac_of_mfma_peak"))))
PY
done
grep "unet3d\|passed\|failed" $O/pytest_video.log | cut -c1-220; cat $O/summary.txt
