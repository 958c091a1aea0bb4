set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" 2>&1 | tail -1
timeout 900 python -m pytest tests/test_igemm_cfgs_gpu.py -m gpu -q --tb=short -k "test_conv_dma_every_cfg or test_conv3x3_raw_post" 2>&1 | tail -6
for v in "1 1 128" "1 0 128" "0 0 128" "1 1 256" "1 1 64"; do set -- $v
IMAGEN_CONV_DMA=$1 IMAGEN_GCA_IN_EPILOGUE=$2 IMAGEN_ACT_PREP_MIN_COUT=$3 timeout 600 python bench.py --timesteps 100 --steps 1 --warmup 1 --no-cpu-baseline --no-roofline 2> gpurun_out/bench_g.err | cut -c95-240
done
timeout 600 python tools/step_profile.py --reps 3 --top 60 > gpurun_out/r02_step_profile_g.txt 2>&1
sed -n '/=== stage (1/,$p' gpurun_out/r02_step_profile_g.txt | sed -n 1,70p | cut -c1-150
