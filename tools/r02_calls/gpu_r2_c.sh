# round-2 GPU call C: all-DMA conv family: parity, raw sweep of all families, model-level parity, short bench
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_igemm_cfgs_gpu.py -m gpu -q --tb=short -k "test_conv_dma_every_cfg or test_act_prep" 2>&1 | tail -30
timeout 900 python tools/igemm_probe.py --sweep-raw > gpurun_out/r02_sweep_raw_c.txt 2>&1; tail -40 gpurun_out/r02_sweep_raw_c.txt | cut -c1-420
timeout 900 python -m pytest tests/test_bench_shapes_gpu.py tests/test_model_gpu.py -m gpu -q --tb=short -x 2>&1 | tail -15
timeout 600 python bench.py --timesteps 50 --steps 1 --warmup 1 --no-cpu-baseline 2> gpurun_out/bench_c.err | cut -c1-1500; tail -3 gpurun_out/bench_c.err
IMAGEN_CONV_DMA=0 timeout 600 python bench.py --timesteps 50 --steps 1 --warmup 1 --no-cpu-baseline --no-roofline 2> gpurun_out/bench_c0.err | cut -c1-400
