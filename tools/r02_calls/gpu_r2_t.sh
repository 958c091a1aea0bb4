cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
smi() { for i in 1 2 3 4 5 6 7 8 9 10; do /opt/rocm/bin/rocm-smi --showclocks --showpower --showuse --json 2>/dev/null | python -c "
import json,sys
try:
    d=json.load(sys.stdin)
    for k,v in d.items():
        print({kk:vv for kk,vv in v.items() if any(s in kk.lower() for s in ('sclk','power','use','fclk','mclk'))})
except Exception as e: print('smi parse failed', e)
"; sleep 0.7; done; }
echo "=== idle"; /opt/rocm/bin/rocm-smi --showclocks --showpower --showperflevel 2>/dev/null | head -30
echo "=== sequential bench running"
(IMAGEN_BENCH_MODE=sequential IMAGEN_CONV_DMA=0 IMAGEN_GCA_IN_EPILOGUE=0 python bench.py --timesteps 300 --steps 6 --warmup 2 --no-cpu-baseline --no-roofline 2>/dev/null > gpurun_out/t_seq.json) &
sleep 14; smi; wait; python -c "import json; r=json.load(open('gpurun_out/t_seq.json')); print('sequential', r['ms_per_step'])"
echo "=== lanes 3 bench running"
(IMAGEN_CONV_DMA=0 IMAGEN_GCA_IN_EPILOGUE=0 python bench.py --mode lanes --lanes 3 --timesteps 300 --steps 9 --warmup 3 --no-cpu-baseline --no-roofline 2>/dev/null > gpurun_out/t_l3.json) &
sleep 16; smi; wait; python -c "import json; r=json.load(open('gpurun_out/t_l3.json')); print('lanes3', r['ms_per_step'])"
echo "=== latency probe (hot chains, mixed chain)"
timeout 600 python tools/latency_probe.py "" 2>&1 | tail -8
