cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
prof() { tag=$1; shift
  env "$@" timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/gp_$tag -- python $R/tools/graph_profile.py run --steps 12 --plan-out /tmp/plan_$tag.json > /tmp/gp_$tag.log 2>&1
  tail -1 /tmp/gp_$tag.log
  f=$(find /tmp/gp_$tag -name "*kernel_trace.csv" | head -1)
  cp $f $R/gpurun_out/r02_trace_$tag.csv; cp /tmp/plan_$tag.json $R/gpurun_out/r02_plan_$tag.json
  python $R/tools/graph_profile.py analyze $f /tmp/plan_$tag.json --top 70 --csv $R/gpurun_out/r02_graph_profile_$tag > $R/gpurun_out/r02_graph_profile_$tag.txt 2>&1
  grep "===" $R/gpurun_out/r02_graph_profile_$tag.txt
}
prof r01path IMAGEN_CONV_DMA=0 IMAGEN_GCA_IN_EPILOGUE=0
prof dma IMAGEN_CONV_DMA=1 IMAGEN_ACT_PREP_MIN_COUT=0 IMAGEN_GCA_IN_EPILOGUE=1
