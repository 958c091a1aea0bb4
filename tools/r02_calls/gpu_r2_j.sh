set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" 2>&1 | tail -1
timeout 900 python -m pytest tests/test_igemm_cfgs_gpu.py -m gpu -q --tb=short -k "test_conv_dma_every_cfg" 2>&1 | tail -6
timeout 900 python tools/igemm_probe.py --sweep-raw "384->256 3x3 @32,256->256 3x3 @32,u2.L3 128->128,192->128 3x3 @64,u2.L2 64->64,u2.L2 128->128,u2.L1 96,u2.L1 64->64" > gpurun_out/r02_sweep_raw_j.txt 2>&1; cat gpurun_out/r02_sweep_raw_j.txt | cut -c1-520
