# round-2 GPU call A: first run of the LDS-staged conv family (parity per cfg, bench-shape replay, tile sweep) + first run of the video path
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" 2>&1 | tail -1
for t in test_conv3x3_block_every_cfg_and_tile_shape test_conv3x3_raw_post_and_ssq_prologue_every_cfg test_1x1_every_cfg_and_epilogue test_strided_and_wide_kernels_every_cfg; do
  timeout 600 python -m pytest tests/test_igemm_cfgs_gpu.py -m gpu -q --tb=line -k $t 2>&1 | tail -30
done
timeout 900 python -m pytest tests/test_bench_shapes_gpu.py -m gpu -q --tb=short 2>&1 | tail -30
timeout 900 python tools/igemm_probe.py --sweep > gpurun_out/r02_sweep_a.txt 2>&1; tail -80 gpurun_out/r02_sweep_a.txt | cut -c1-400
bash tools/run_video_gpu.sh 2>&1 | tail -60
