#!/bin/bash
# Builds the CPU functional emulation of csrc/igemm.hip, conv_dma.hip and conv_stream.hip: imagen-pytorch_amd/libimagen_emul.so, and with "remat" as the first argument
# libimagen_emul_remat.so (-DIGEMM_EPI_REMAT); `NAME -Dflags...` builds libimagen_emul_NAME.so with those flags.  Host clang (the ROCm toolchain's), no GPU code; skipped when the library is newer than
# its sources.  See tools/emul/README.md.
set -e
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
P=$ROOT/imagen-pytorch_amd
CL=/opt/rocm/lib/llvm/bin/clang++
OUT=$P/build/emul
mkdir -p $OUT
TAG=""; DEFS=""
if [ "${1:-}" = "remat" ]; then TAG="_remat"; DEFS="-DIGEMM_EPI_REMAT";
elif [ -n "${1:-}" ]; then TAG="_$1"; shift; DEFS="$*"; fi       # any other name: the remaining arguments are the -D flags of that variant
LIB=$P/libimagen_emul$TAG.so
SRCS="$P/csrc/igemm.hip $P/csrc/conv_dma.hip $P/csrc/conv_stream.hip $P/csrc/conv_epilogue.h $P/csrc/gca_device.h $P/csrc/common.h $ROOT/include/imagen_hip.h $ROOT/tools/emul/emul_runtime.cpp $ROOT/tools/emul/hip/hip_runtime.h $ROOT/tools/emul/build_emul_lib.sh"
if [ -f "$LIB" ]; then
  fresh=1
  for s in $SRCS; do [ "$s" -nt "$LIB" ] && fresh=0; done
  if [ $fresh = 1 ]; then echo "up to date: $LIB"; exit 0; fi
fi
FLAGS="-x c++ -std=c++17 -O1 -fPIC -w -DIMAGEN_EMUL $DEFS -I$ROOT/tools/emul -I$ROOT/include -I$P/csrc"
$CL $FLAGS -c $P/csrc/igemm.hip -o $OUT/igemm$TAG.o &
$CL $FLAGS -c $P/csrc/conv_dma.hip -o $OUT/conv_dma$TAG.o &
$CL $FLAGS -c $P/csrc/conv_stream.hip -o $OUT/conv_stream$TAG.o &
$CL $FLAGS -c $ROOT/tools/emul/emul_runtime.cpp -o $OUT/runtime$TAG.o &
wait
$CL -shared -fPIC -o $LIB $OUT/igemm$TAG.o $OUT/conv_dma$TAG.o $OUT/conv_stream$TAG.o $OUT/runtime$TAG.o
echo built $LIB
