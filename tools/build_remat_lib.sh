#!/bin/bash
# Builds imagen-pytorch_amd/libimagen_hip_remat.so: the product sources with igemm.hip compiled -DIGEMM_EPI_REMAT (per-lane epilogue
# constants rematerialised inside the tile loop instead of spilled across the k loop).  A/B it against the product library
# with IMAGEN_LIB_PATH=imagen-pytorch_amd/libimagen_hip_remat.so (parity: tests/test_igemm_cfgs_gpu.py tests/test_bench_shapes_gpu.py;
# time: bench.py --no-roofline --no-cpu-baseline).  tools/scratch_report.py compares the spill placement of the two builds.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
P=$ROOT/imagen-pytorch_amd
python -c "import sys; sys.path.insert(0, '$ROOT'); import __graft_entry__ as g; g.build()" > /dev/null
TL=$(python -c "import torch, os; print(os.path.join(os.path.dirname(torch.__file__), 'lib'))")
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DIGEMM_EPI_REMAT -I$ROOT/include -I$P/csrc -c $P/csrc/igemm.hip -o $P/build/igemm_remat.o
hipcc --offload-arch=gfx950 -shared -fPIC -o $P/libimagen_hip_remat.so $P/build/igemm_remat.o $P/build/conv_lds.o $P/build/conv_dma.o $P/build/conv_stream.o $P/build/attention.o $P/build/elementwise.o $P/build/sampler.o $P/build/temporal.o $P/build/codesize.o $P/build/capi.o -L$TL -Wl,-rpath,$TL -Wl,-rpath,/opt/rocm/lib
echo built $P/libimagen_hip_remat.so
