set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" 2>&1 | tail -1
timeout 600 python tools/step_profile.py --reps 3 --top 70 > gpurun_out/r02_step_profile_dma.txt 2>&1
IMAGEN_CONV_DMA=0 timeout 600 python tools/step_profile.py --reps 3 --top 70 > gpurun_out/r02_step_profile_nodma.txt 2>&1
grep -A 16 "=== stage" gpurun_out/r02_step_profile_dma.txt | cut -c1-150
grep -A 16 "=== stage" gpurun_out/r02_step_profile_nodma.txt | cut -c1-150
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -E "Counter_Name" | grep -E "SQ_.*LDS|SQ_WAIT|SQ_INST_CYCLES|SQ_ACTIVE_INST|SQ_IFETCH|SQ_INSTS_(LDS|SALU|VALU|SMEM|VMEM)|TCP_.*STALL|TCP_PENDING|TA_BUSY|TCC_BUSY" | tr -s '\t ' ' ' | tr '\n' ';' | cut -c1-3000
echo
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d /tmp/pmc_sq -- python $GRAFT_REPO_ROOT/tools/igemm_probe.py --sweep-raw "384->256 3x3 @32,192->128,u2.L2 64->64" > /tmp/pmc_sq.log 2>&1
tail -2 /tmp/pmc_sq.log | cut -c1-300
f=$(find /tmp/pmc_sq -name "*counter_collection.csv" | head -1); python $GRAFT_REPO_ROOT/tools/pmc_summary.py $f $GRAFT_REPO_ROOT/gpurun_out/r02_pmc_sq_dma.json > /dev/null
python - <<'PY'
import json,os
d=json.load(open(os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/r02_pmc_sq_dma.json'))
for k,v in sorted(d.items()):
    if 'conv_dma' in k or 'igemm_kernel' in k:
        wc=v['SQ_WAVE_CYCLES']
        print(k[28:75], 'launches',v['launches'],'wavecyc(M)',round(wc/1e6,1),'wait_any',round(v['SQ_WAIT_ANY']/wc,2),'wait_inst',round(v['SQ_WAIT_INST_ANY']/wc,2),'wait_lds',round(v['SQ_WAIT_INST_LDS']/wc,2),'active',round(v['SQ_ACTIVE_INST_ANY']/wc,2),'lds_act(M)',round(v['SQ_LDS_IDX_ACTIVE']/1e6,2),'bank_conf',round(v['SQ_LDS_BANK_CONFLICT']/max(v['SQ_LDS_IDX_ACTIVE'],1),2),'busy(M)',round(v['SQ_BUSY_CYCLES']/1e6,2))
PY
