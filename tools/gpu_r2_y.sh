cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for k in 2 4; do
rm -rf /tmp/pmc_att
IMAGEN_ATTN_KERNEL=$k timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --output-format csv -d /tmp/pmc_att -- python $R/tools/latency_probe.py "attn self 1024" > /tmp/pmc_att.log 2>&1
f=$(find /tmp/pmc_att -name "*counter_collection.csv" | head -1)
python $R/tools/pmc_summary.py $f /tmp/att_$k.json > /dev/null
python - <<PY
import json
d=json.load(open('/tmp/att_$k.json'))
for k,v in d.items():
    if 'attention' in k:
        wc=v['SQ_WAVE_CYCLES']
        print(k[:60], {c:round(x/wc,3) if c!='launches' else x for c,x in v.items()}, 'wave_cycles(M)', round(wc/1e6,2), 'busy_cycles', round(v['SQ_BUSY_CYCLES']))
PY
done
rocprofv3 -L 2>/dev/null | grep -E "SQ_ACTIVE_INST_|SQ_INST_CYCLES|SQ_WAIT_INST_LDS|SQ_INSTS_(LDS|VALU|MFMA|SALU)|SQ_VALU_MFMA" | awk '{print $0}' | cut -c1-160 | head -30
