#!/bin/bash
# Builds imagen-pytorch_amd/libimagen_hip_probe.so: the product sources with -DCL_PROBE / -DCD_PROBE (run-time ablation switches in the
# LDS-staged, all-DMA and streaming conv kernels, see CL_DBG in csrc/conv_epilogue.h).  Used only by `IMAGEN_LIB_PATH=... python tools/*_probe.py`.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
P=$ROOT/imagen-pytorch_amd
python -c "import sys; sys.path.insert(0, '$ROOT'); import __graft_entry__ as g; g.build()" > /dev/null
TL=$(python -c "import torch, os; print(os.path.join(os.path.dirname(torch.__file__), 'lib'))")
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DCL_PROBE -I$ROOT/include -I$P/csrc -c $P/csrc/conv_lds.hip -o $P/build/conv_lds_probe.o &
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DCD_PROBE -I$ROOT/include -I$P/csrc -c $P/csrc/conv_dma.hip -o $P/build/conv_dma_probe.o &
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DCL_PROBE -I$ROOT/include -I$P/csrc -c $P/csrc/conv_stream.hip -o $P/build/conv_stream_probe.o &
wait
hipcc --offload-arch=gfx950 -shared -fPIC -o $P/libimagen_hip_probe.so $P/build/conv_lds_probe.o $P/build/conv_dma_probe.o $P/build/conv_stream_probe.o $P/build/igemm.o $P/build/attention.o $P/build/elementwise.o $P/build/sampler.o $P/build/temporal.o $P/build/codesize.o $P/build/capi.o -L$TL -Wl,-rpath,$TL -Wl,-rpath,/opt/rocm/lib
echo built $P/libimagen_hip_probe.so
