#!/bin/bash
# A/B libraries of csrc/conv_small.hip (the product objects with another conv_small.o; built here, shipped in-tree to the GPU box).
#   bash tools/gpu/build_small_variants.sh            -> imagen-pytorch_amd/libimagen_hip_<tag>.so for every variant below
set -e
cd "$(dirname "$0")/../.."
P=imagen-pytorch_amd
TL=$(python -c "import torch,os;print(os.path.join(os.path.dirname(torch.__file__),'lib'))")
OBJS=$(ls $P/build/*.o | grep -v "conv_small.o")
build() {   # tag flags...
  tag=$1; shift 1
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -I$P/csrc "$@" -c $P/csrc/conv_small.hip -o /tmp/conv_small_$tag.o
  hipcc --offload-arch=gfx950 -shared -fPIC -o $P/libimagen_hip_$tag.so $OBJS /tmp/conv_small_$tag.o -L$TL -Wl,-rpath,$TL -Wl,-rpath,/opt/rocm/lib
  echo "built $P/libimagen_hip_$tag.so"
}
#build cs18 -DCS_MINW=1 -DCS_BATCH=8      # one workgroup per CU, 8 staged pieces in flight
build cstrace -DCS_TRACE                 # s_memtime stamps at the phase boundaries (tools/small_bench.py --trace)
build csph1 -DCS_PHASES=1                # every pixel tile walks its K slice from the head (the A/B of the phase-shifted walk)
