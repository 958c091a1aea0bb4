cd $GRAFT_REPO_ROOT
export IMAGEN_STREAM_MIN_TILES=1 IMAGEN_STREAM_CONCAT_PRO=1
timeout 120 python -m pytest tests/test_igemm_cfgs_gpu.py -m gpu -x -q -k "conv_stream" 2>&1 | tail -5
echo "--- form 4 (lean DMA)"; timeout 300 python tools/stream_probe.py 2>&1 | grep "@256\|@128" | cut -c1-24,84-130
