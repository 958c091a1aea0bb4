#!/bin/bash
# A/B libraries of csrc/conv_small.hip (the product objects with another conv_small.o; built here, shipped in-tree to the GPU box).
#   bash tools/gpu/build_small_variants.sh [tag ...]     -> imagen-pytorch_amd/libimagen_hip_<tag>.so for the named variants (default: cstrace)
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
for tag in ${@:-cstrace}; do
  case $tag in
    cs18) build cs18 -DCS_MINW=1 -DCS_BATCH=8 ;;                      # one workgroup per CU, 8 staged pieces in flight (call I)
    cstrace) build cstrace -DCS_TRACE ;;                              # s_memtime stamps at the phase boundaries (tools/small_bench.py --trace)
    csph1) build csph1 -DCS_PHASES=1 ;;                               # every pixel tile walks its K slice from the head (call M)
    csph8) build csph8 -DCS_PHASES=8 ;;                               # eight phases (call P)
    csr16) build csr16 -DCS_RING=16 -DCS_MINW=1 -DCS_BATCH=8 ;;       # 16 weight fragments in flight per wave, one workgroup per CU (call P)
    *) echo "unknown variant $tag"; exit 1 ;;
  esac
done
