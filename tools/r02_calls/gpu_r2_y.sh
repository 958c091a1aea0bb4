cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export IMAGEN_STREAM_MIN_TILES=1
for form in 1 2; do
rm -rf /tmp/pmc_cs
IMAGEN_STREAM_FORM=$form timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_INST_LDS --output-format csv -d /tmp/pmc_cs -- python $R/tools/stream_probe.py > /tmp/pmc_cs.log 2>&1
f=$(find /tmp/pmc_cs -name "*counter_collection.csv" | head -1)
python - <<PY
import csv, collections
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open("$f")):
    if 'conv_stream' in r['Kernel_Name'] and r['Grid_Size'] in ('262144','131072'):
        agg[(r['Kernel_Name'][22:75], r['Grid_Size'])][r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in agg.items():
    m={c:sum(x)/len(x) for c,x in v.items()}
    wc=m['SQ_WAVE_CYCLES']
    print("form $form", k, 'n', len(v['SQ_WAVE_CYCLES']), {c:round(x/wc,3) for c,x in m.items() if c!='SQ_WAVE_CYCLES'}, 'wave_cyc(M)', round(wc/1e6,1), 'busy/32', round(m['SQ_BUSY_CYCLES']/32))
PY
done
