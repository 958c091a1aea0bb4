cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
run() { printf "%-50s" "$1"; shift; env "$@" timeout 600 python bench.py --timesteps 200 --steps 4 --warmup 2 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read()); print(r['ms_per_step'], r['value'])"; }
(cd _r01 && run "r01 tree (a72d774)" X=1)
run "HEAD sequential dma0" IMAGEN_BENCH_MODE=sequential IMAGEN_CONV_DMA=0 IMAGEN_GCA_IN_EPILOGUE=0
(cd _r01 && run "r01 tree (a72d774) again" X=1)
run "HEAD sequential defaults" IMAGEN_BENCH_MODE=sequential
timeout 600 python tools/latency_probe.py > gpurun_out/r02_latency_probe.txt 2>&1
IMAGEN_GCA_FINAL_SLOW=1 timeout 300 python tools/latency_probe.py gca >> gpurun_out/r02_latency_probe.txt 2>&1
cat gpurun_out/r02_latency_probe.txt
IMAGEN_LIB_PATH=imagen-pytorch_amd/libimagen_hip_trace.so timeout 300 python tools/igemm_probe.py --timeline "u1.L3 128->128 3x3 @8,u1.L2 64->64 3x3 @16,to_q 256->512,u1.L0 32->32 3x3 @64" > gpurun_out/r02_timeline_small.txt 2>&1
cat gpurun_out/r02_timeline_small.txt
