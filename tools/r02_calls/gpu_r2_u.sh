cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export IMAGEN_CONV_DMA=0 IMAGEN_GCA_IN_EPILOGUE=0
timeout 600 python tools/latency_probe.py --separate > gpurun_out/r02_latency_separate.txt 2>&1
cat gpurun_out/r02_latency_separate.txt
