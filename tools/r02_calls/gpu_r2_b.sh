# round-2 GPU call B: ablation of the LDS-staged conv kernel + SQ counters
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export IMAGEN_LIB_PATH=$GRAFT_REPO_ROOT/imagen-pytorch_amd/libimagen_hip_probe.so
timeout 600 python -m pytest tests/test_igemm_cfgs_gpu.py -m gpu -q --tb=line -k "test_conv3x3_block_every_cfg_and_tile_shape or test_1x1_every_cfg" 2>&1 | tail -8
timeout 900 python tools/conv_probe.py > gpurun_out/r02_conv_probe_b.txt 2>&1; cat gpurun_out/r02_conv_probe_b.txt | cut -c1-330
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -i -E "MFMA|SQ_BUSY_CY|SQ_WAVE_CYCLES|SQ_WAIT_INST_ANY|SQ_WAIT_ANY|SQ_ACTIVE_INST_ANY|LDS_BANK|LDS_IDX|SQ_INSTS_VALU |SQ_WAVES |TCC_HIT_sum|TCC_MISS_sum|TCP_PENDING|TCC_EA0_RDREQ_sum|TCC_REQ_sum" | cut -c1-150 | head -40
for shape in "384->256" "64->64 @64"; do
CONV_PROBE_ONLY_FULL=1 timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d /tmp/pmc_sq -- python $GRAFT_REPO_ROOT/tools/conv_probe.py "$shape" > /tmp/pmc_sq.log 2>&1
tail -3 /tmp/pmc_sq.log
f=$(find /tmp/pmc_sq -name "*counter_collection.csv" | head -1); python $GRAFT_REPO_ROOT/tools/pmc_summary.py $f "$GRAFT_REPO_ROOT/gpurun_out/r02_pmc_sq_$(echo $shape | tr -c 'a-z0-9' '_').json" | cut -c1-400
rm -rf /tmp/pmc_sq
done
