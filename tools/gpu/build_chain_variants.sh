#!/bin/bash
# A/B libraries of csrc/rowchain.hip for tools/chain_bench.py: the product objects with another rowchain.o (built here, shipped in-tree to the GPU box).
#   bash tools/gpu/build_chain_variants.sh            -> imagen-pytorch_amd/libimagen_hip_<tag>.so for every variant below
set -e
cd "$(dirname "$0")/../.."
P=imagen-pytorch_amd
TL=$(python -c "import torch,os;print(os.path.join(os.path.dirname(torch.__file__),'lib'))")
OBJS=$(ls $P/build/*.o | grep -v "rowchain.o")
build() {   # tag source flags...
  tag=$1; src=$2; shift 2
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -I$P/csrc "$@" -c $src -o /tmp/rowchain_$tag.o
  hipcc --offload-arch=gfx950 -shared -fPIC -o $P/libimagen_hip_$tag.so $OBJS /tmp/rowchain_$tag.o -L$TL -Wl,-rpath,$TL -Wl,-rpath,/opt/rocm/lib
  echo "built $P/libimagen_hip_$tag.so"
}
git show 785fb26:imagen-pytorch_amd/csrc/rowchain.hip > /tmp/rowchain_v1.hip      # round 5's first version (call A): 4-step ring, loads at use
#build rc1 /tmp/rowchain_v1.hip
#build r8 $P/csrc/rowchain.hip -DROWCHAIN_RING=8
#build r4w4 $P/csrc/rowchain.hip -DROWCHAIN_RING=4 -DROWCHAIN_MINW=4
#build r8w4 $P/csrc/rowchain.hip -DROWCHAIN_RING=8 -DROWCHAIN_MINW=4
build trace $P/csrc/rowchain.hip -DROWCHAIN_TRACE
