set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" 2>&1 | tail -1
timeout 900 python -m pytest tests/test_igemm_cfgs_gpu.py -m gpu -q --tb=short -k "test_conv_dma_every_cfg or test_conv3x3_raw_post" 2>&1 | tail -12
timeout 900 python -m pytest tests/test_model_gpu.py tests/test_bench_shapes_gpu.py -m gpu -q --tb=short -x 2>&1 | tail -8
for v in "1 1" "1 0" "0 0"; do set -- $v
IMAGEN_CONV_DMA=$1 IMAGEN_GCA_IN_EPILOGUE=$2 timeout 600 python bench.py --timesteps 100 --steps 1 --warmup 1 --no-cpu-baseline --no-roofline 2> gpurun_out/bench_f.err | cut -c1-240
done
timeout 600 python tools/step_profile.py --reps 3 --top 50 > gpurun_out/r02_step_profile_f.txt 2>&1
grep -A 18 "=== stage" gpurun_out/r02_step_profile_f.txt | cut -c1-150
