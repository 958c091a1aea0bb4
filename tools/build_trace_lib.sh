#!/bin/bash
# Builds imagen-pytorch_amd/libimagen_hip_trace.so: the product sources with -DIGEMM_TRACE (s_memtime stamps in the igemm
# roles, see csrc/igemm.hip).  Used only by `IMAGEN_LIB_PATH=... python tools/igemm_probe.py --timeline`.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
P=$ROOT/imagen-pytorch_amd
python -c "import sys; sys.path.insert(0, '$ROOT'); import __graft_entry__ as g; g.build()" > /dev/null
TL=$(python -c "import torch, os; print(os.path.join(os.path.dirname(torch.__file__), 'lib'))")
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DIGEMM_TRACE -I$ROOT/include -I$P/csrc -c $P/csrc/igemm.hip -o $P/build/igemm_trace.o
hipcc --offload-arch=gfx950 -shared -fPIC -o $P/libimagen_hip_trace.so $P/build/igemm_trace.o $P/build/conv_lds.o $P/build/conv_dma.o $P/build/attention.o $P/build/elementwise.o $P/build/sampler.o $P/build/temporal.o $P/build/conv_stream.o $P/build/codesize.o $P/build/capi.o -L$TL -Wl,-rpath,$TL -Wl,-rpath,/opt/rocm/lib
echo built $P/libimagen_hip_trace.so
