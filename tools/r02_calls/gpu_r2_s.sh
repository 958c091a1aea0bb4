cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export IMAGEN_LIB_PATH=imagen-pytorch_amd/libimagen_hip_trace.so IMAGEN_CONV_DMA=0 IMAGEN_GCA_IN_EPILOGUE=0
timeout 600 python tools/insitu_trace.py "to_time_cond,to_time_tokens,time_mlps,ff.lin1,ff.lin2,downs.3.1.block1,downs.3.2.0.block,mid_block1.block,downs.2.4,ctx.dyn.self,qkv,to_out" > gpurun_out/r02_insitu_trace.txt 2>&1
cat gpurun_out/r02_insitu_trace.txt | cut -c1-400
