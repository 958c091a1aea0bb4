cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_model_gpu.py -m gpu -x -q --tb=long -k pipelined 2>&1 | tail -60 > gpurun_out/r02_q_pytest.txt
grep -n "Error\|assert\|passed\|failed" gpurun_out/r02_q_pytest.txt | head
run() { printf "%-50s" "$1"; shift; env "$@" timeout 600 python bench.py --timesteps 200 --steps 6 --warmup 3 --no-cpu-baseline --no-roofline 2>gpurun_out/bench_q.err | python -c "import json,sys; r=json.loads(sys.stdin.read()); print(r['ms_per_step'], r['value'])"; }
export IMAGEN_CONV_DMA=0 IMAGEN_GCA_IN_EPILOGUE=0
run "r01path sequential" IMAGEN_BENCH_MODE=sequential
run "r01path lanes 2" IMAGEN_BENCH_MODE=lanes
python bench.py --timesteps 200 --steps 6 --warmup 3 --no-cpu-baseline --no-roofline --mode lanes --lanes 3 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('lanes 3', r['ms_per_step'], r['value'])"
python bench.py --timesteps 200 --steps 8 --warmup 4 --no-cpu-baseline --no-roofline --mode lanes --lanes 4 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('lanes 4', r['ms_per_step'], r['value'])"
python bench.py --timesteps 1000 --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --mode sequential 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('seq 1000', r['ms_per_step'], r['value'])"
