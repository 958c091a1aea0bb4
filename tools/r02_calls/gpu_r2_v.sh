cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_fusion_gpu.py -m gpu -x -q 2>&1 | tail -3
run() { printf "%-50s" "$1"; shift; env "$@" timeout 600 python bench.py --timesteps 200 --steps 4 --warmup 2 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read()); print(r['ms_per_step'], r['value'])"; }
export IMAGEN_CONV_DMA=0 IMAGEN_GCA_IN_EPILOGUE=0 IMAGEN_BENCH_MODE=sequential
run "code warm off" IMAGEN_CODE_WARM=0
run "code warm on" IMAGEN_CODE_WARM=1
run "code warm off" IMAGEN_CODE_WARM=0
run "code warm on" IMAGEN_CODE_WARM=1
run "code warm on, lanes 3" IMAGEN_CODE_WARM=1 IMAGEN_BENCH_MODE=lanes
