#!/bin/bash
# A/B builds of the kernel library: csrc/igemm.hip compiled with extra -D flags, everything else the product's objects.
#     bash tools/build_variant_lib.sh remat  -DIGEMM_EPI_REMAT
#     bash tools/build_variant_lib.sh onewg  -DIGEMM_ONE_WG -DIGEMM_LA1=12 -DIGEMM_LA2=10
# -> imagen-pytorch_amd/libimagen_hip_<name>.so, selected with IMAGEN_LIB_PATH.  The same flags under tools/emul/build_emul_lib.sh give the
# CPU-executable twin (functional check before any GPU minute); tools/scratch_report.py the spill placement.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
P=$ROOT/imagen-pytorch_amd
NAME=$1; shift
python -c "import sys; sys.path.insert(0, '$ROOT'); import __graft_entry__ as g; g.build()" > /dev/null
TL=$(python -c "import torch, os; print(os.path.join(os.path.dirname(torch.__file__), 'lib'))")
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC "$@" -I$ROOT/include -I$P/csrc -c $P/csrc/igemm.hip -o $P/build/igemm_$NAME.o
hipcc --offload-arch=gfx950 -shared -fPIC -o $P/libimagen_hip_$NAME.so $P/build/igemm_$NAME.o $P/build/conv_lds.o $P/build/conv_dma.o $P/build/conv_stream.o $P/build/attention.o $P/build/elementwise.o $P/build/sampler.o $P/build/temporal.o $P/build/codesize.o $P/build/capi.o -L$TL -Wl,-rpath,$TL -Wl,-rpath,/opt/rocm/lib
echo built $P/libimagen_hip_$NAME.so
