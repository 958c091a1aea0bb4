#!/bin/bash
# round 4, GPU call J: bounded attention with 4-wave workgroups (two per CU) against the 8-wave build of the previous commit (same box)
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r04_j; mkdir -p $O
timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "test_attention" -p no:cacheprovider > $O/pytest_kernels.log 2>&1; echo "pytest kernels rc $?" >> $O/summary.txt
for t in abl_nw8 hip; do
  echo "== libimagen_$t.so" >> $O/attn_ab.log
  IMAGEN_LIB_PATH=$GRAFT_REPO_ROOT/imagen-pytorch_amd/libimagen_$t.so timeout 200 python tools/attn_bench.py --modes 1 --iters 20 2>&1 | grep site >> $O/attn_ab.log
done
timeout 200 python tools/attn_bench.py --sites self-1024 --iters 20 2>&1 | grep site >> $O/attn_ab.log
for v in "IMAGEN_ATTN_BOUNDED=0" "IMAGEN_ATTN_BOUNDED=1"; do
  env $v timeout 300 python tools/step_time.py --steps 60 --reps 3 --tag "$v" 2>/dev/null | tail -n 1 >> $O/step_ab.jsonl
done
tail -n 2 $O/pytest_kernels.log; cat $O/summary.txt $O/attn_ab.log $O/step_ab.jsonl
