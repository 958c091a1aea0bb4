#!/bin/bash
# round 4, GPU call F: bounded attention v2 (fragments read one half step ahead, no shift) + GlobalContext partials from conv_stream's epilogue:
# kernel tests on hardware, kernel-level attention A/B, step-level A/B of both switches, whole-Unet parity.
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r04_f; mkdir -p $O
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1
timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_igemm_cfgs_gpu.py -m gpu -q -x -k "test_attention or test_conv_stream_family or gca or global_context" -p no:cacheprovider > $O/pytest_kernels.log 2>&1; echo "pytest kernels rc $?" >> $O/summary.txt
timeout 300 python tools/attn_bench.py --out $O/attn_bench.jsonl > $O/attn_bench.log 2>&1; echo "attn_bench rc $?" >> $O/summary.txt
cd /tmp
rm -rf /tmp/pmcA; timeout 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY --output-format csv -d /tmp/pmcA -- python $GRAFT_REPO_ROOT/tools/attn_bench.py --iters 2 --sites self-1024 > $O/pmc_attn.log 2>&1
f=$(find /tmp/pmcA -name '*counter_collection.csv' | head -n 1); [ -n "$f" ] && python $GRAFT_REPO_ROOT/tools/pmc_summary.py $f $O/pmc_attention.json > /dev/null 2>&1
rm -rf /tmp/pmcB; timeout 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_MFMA SQ_INSTS_VALU_TRANS SQ_INST_LEVEL_LDS --output-format csv -d /tmp/pmcB -- python $GRAFT_REPO_ROOT/tools/attn_bench.py --iters 2 --sites self-1024 > $O/pmc_attn2.log 2>&1
f=$(find /tmp/pmcB -name '*counter_collection.csv' | head -n 1); [ -n "$f" ] && python $GRAFT_REPO_ROOT/tools/pmc_summary.py $f $O/pmc_attention2.json > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
for v in "IMAGEN_ATTN_BOUNDED=0 IMAGEN_STREAM_GCA=0" "IMAGEN_ATTN_BOUNDED=1 IMAGEN_STREAM_GCA=0" "IMAGEN_ATTN_BOUNDED=1 IMAGEN_STREAM_GCA=1"; do
  env $v timeout 300 python tools/step_time.py --steps 60 --reps 3 --tag "$v" 2>/dev/null | tail -n 1 >> $O/step_ab.jsonl
done
timeout 900 python -m pytest tests/test_model_gpu.py -m gpu -q -x -k "unet_forward_vs_oracle" -p no:cacheprovider > $O/pytest_unet.log 2>&1; echo "pytest unet rc $?" >> $O/summary.txt
cp gpurun_out/parity_measured.json $O/parity_measured.json 2>/dev/null
tail -n 3 $O/pytest_kernels.log $O/pytest_unet.log; cat $O/summary.txt $O/attn_bench.log $O/step_ab.jsonl
