# First supervised GPU run of the Imagen-Video path (opt-in tests, then the C5 timing).  Usage (from the build container):
#   gpurun --timeout 600 -- 'bash tools/run_video_gpu.sh'
set -x
cd $GRAFT_REPO_ROOT
export IMAGEN_VIDEO_GPU_TESTS=1
# kernels first, one test per process: a faulting kernel then only takes its own process down
for t in test_igemm_view_patterns test_temporal_peg_kernel test_temporal_attention_kernel test_unet3d_forward_vs_reference_fixture test_video_cascade_sample_vs_reference_fixture test_video_elucidated_sample_vs_reference_fixture; do
  timeout 120 python -m pytest tests/test_video_gpu.py -m gpu -q -x -k $t 2>&1 | tail -15
done
timeout 300 python tools/time_c5.py 25 2>&1 | tail -5
