set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" 2>&1 | tail -1
timeout 900 python -m pytest tests/test_igemm_cfgs_gpu.py tests/test_bench_shapes_gpu.py -m gpu -q --tb=short 2>&1 | tail -8
timeout 600 python tools/step_profile.py --reps 3 --top 70 > gpurun_out/r02_step_profile_dma2.txt 2>&1
grep -A 18 "=== stage" gpurun_out/r02_step_profile_dma2.txt | cut -c1-150
grep -n "block# \[384->256 k3 @32\]\|block# \[256->256 k3 @32\]\|block# \[192->128 k3 @64\]\|block#.prep\|\[128->128 k3 @64\]\|\[128->128 k3 @32\]\|\[64->64 k3 @64\]\|k3 @8\]\|k3 @16\]" gpurun_out/r02_step_profile_dma2.txt | cut -c1-160
timeout 600 python bench.py --timesteps 50 --steps 1 --warmup 1 --no-cpu-baseline --no-roofline 2> gpurun_out/bench_e.err | cut -c1-300
