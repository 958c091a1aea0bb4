#!/bin/bash
# Round 5, call C: per-launch A/B of the ROWCHAIN kernel variants on the benchmark's shapes (tools/chain_bench.py under rocprofv3), RESPREP on hardware
# (tests + step A/B), the model tests that cover the changed plans.
#   gpurun --timeout 1500 -- 'bash tools/gpu/r05_c.sh'
set -u
cd "$(dirname "$0")/../.."
R=$PWD
OUT=$R/gpurun_out/r05_c
mkdir -p $OUT
python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1; tail -n 1 $OUT/build.log
echo "=== rowchain tests"
timeout 300 python -m pytest tests/test_rowchain_gpu.py -m gpu -q --tb=short -p no:cacheprovider > $OUT/pytest_rowchain.txt 2>&1; tail -n 12 $OUT/pytest_rowchain.txt | cut -c1-220
echo "=== model tests on the changed plans"
timeout 500 python -m pytest tests/test_model_gpu.py -m gpu -q --tb=short -p no:cacheprovider -k "forward_vs_oracle or sample_vs_reference or time_table" > $OUT/pytest_model.txt 2>&1; tail -n 16 $OUT/pytest_model.txt | cut -c1-250
echo "=== step A/B: RESPREP"
for rc in 2 1 0; do IMAGEN_ROWCHAIN=$rc timeout 200 python tools/step_time.py --steps 60 --reps 3 --tag rowchain$rc 2>/dev/null | tail -n 1 | tee -a $OUT/step_ab.jsonl; done
cd /tmp && export TMPDIR=/tmp
echo "=== chain_bench: kernel variants"
for tag in product rc1 r8 r4w4 r8w4; do
  lib=$R/imagen-pytorch_amd/libimagen_hip.so; [ $tag != product ] && lib=$R/imagen-pytorch_amd/libimagen_hip_$tag.so
  extra=""; [ $tag = rc1 ] && extra="CHAIN_BENCH_NO_RESPREP=1"
  rm -rf /tmp/cb_$tag
  env IMAGEN_LIB_PATH=$lib $extra timeout 240 rocprofv3 --kernel-trace --output-format csv -d /tmp/cb_$tag -- python $R/tools/chain_bench.py --tag $tag --list /tmp/cases_$tag.json > /tmp/cb_$tag.log 2>&1
  tail -n 1 /tmp/cb_$tag.log | cut -c1-200
  python $R/tools/chain_bench.py --parse /tmp/cb_$tag /tmp/cases_$tag.json | tee -a $OUT/chain_bench.jsonl | cut -c1-2500
done
