cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q --tb=short 2>&1 | tail -8 > gpurun_out/r02_p_pytest.txt
cat gpurun_out/r02_p_pytest.txt
run() { printf "%-50s" "$1"; shift; env "$@" timeout 600 python bench.py --timesteps 200 --steps 4 --warmup 2 --no-cpu-baseline --no-roofline 2>gpurun_out/bench_p.err | python -c "import json,sys; r=json.loads(sys.stdin.read()); print(r['ms_per_step'], r['value'])"; }
export IMAGEN_CONV_DMA=0 IMAGEN_GCA_IN_EPILOGUE=0
run "r01path sequential" IMAGEN_BENCH_MODE=sequential
run "r01path sequential slow-gca-final" IMAGEN_BENCH_MODE=sequential IMAGEN_GCA_FINAL_SLOW=1
run "r01path pipeline" IMAGEN_BENCH_MODE=pipeline
run "r01path lanes" IMAGEN_BENCH_MODE=lanes
export IMAGEN_CONV_DMA=1 IMAGEN_GCA_IN_EPILOGUE=1 IMAGEN_ACT_PREP_MIN_COUT=0
run "dma sequential" IMAGEN_BENCH_MODE=sequential
run "dma pipeline" IMAGEN_BENCH_MODE=pipeline
tail -3 gpurun_out/bench_p.err
