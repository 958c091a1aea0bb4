#!/bin/bash
# round 4, GPU call A: conv_pro.hip meets hardware (family test, kernel-level A/B against conv_stream.hip, step-level A/B), whole-Unet parity
# figures with the split-precision weights (UNET_TOL 1.0e-3, the 16-row bench-plan cases), bench-shape replay.
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04_a; mkdir -p $O
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1
timeout 600 python -m pytest tests/test_igemm_cfgs_gpu.py -m gpu -q -k "conv_pro or conv_stream" -p no:cacheprovider > $O/pytest_families.log 2>&1; echo "families rc $?" >> $O/summary.txt
timeout 300 python tools/stream_bench.py --out $O/stream_bench.jsonl > $O/stream_bench.log 2>&1; echo "stream_bench rc $?" >> $O/summary.txt
for v in "IMAGEN_CONV_PRO=0" "IMAGEN_CONV_PRO=1" "IMAGEN_CONV_PRO=2" "IMAGEN_CONV_PRO=1 IMAGEN_SPLIT_SMALL=0" "IMAGEN_CONV_PRO=1 IMAGEN_SPLIT_SMALL=0 IMAGEN_SPLIT_STATIC=0"; do
  env $v timeout 300 python tools/step_time.py --steps 60 --reps 3 --tag "$v" 2>/dev/null | tail -1 >> $O/step_ab.jsonl
done
timeout 900 python -m pytest tests/test_model_gpu.py -m gpu -q -k "unet_forward_vs_oracle" -p no:cacheprovider > $O/pytest_parity.log 2>&1; echo "parity rc $?" >> $O/summary.txt
cp gpurun_out/parity_measured.json $O/parity_measured.json 2>/dev/null
timeout 600 python -m pytest tests/test_bench_shapes_gpu.py -m gpu -q -p no:cacheprovider > $O/pytest_bench_shapes.log 2>&1; echo "bench_shapes rc $?" >> $O/summary.txt
tail -3 $O/pytest_families.log $O/pytest_parity.log $O/pytest_bench_shapes.log; cat $O/summary.txt $O/stream_bench.log $O/step_ab.jsonl
