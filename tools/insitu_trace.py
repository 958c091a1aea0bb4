#!/usr/bin/env python
"""In-situ timelines of selected igemm launches INSIDE the real denoiser step (cold caches, real neighbours): s_memtime stamps of
workgroup 0's first consumer / producer wave (IGEMM_TRACE build of the library, tools/build_trace_lib.sh).

    IMAGEN_LIB_PATH=imagen-pytorch_amd/libimagen_hip_trace.so IMAGEN_CONV_DMA=0 IMAGEN_GCA_IN_EPILOGUE=0 python tools/insitu_trace.py label[,label...]

Only launches whose `gate` argument is unused can be traced (the stamps are written through it).
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from imagen_pytorch_amd import _abi

dev = torch.device("cuda", 0)
want = sys.argv[1].split(",") if len(sys.argv) > 1 else ["to_time_cond", "ff.lin2", "downs.3.1.block1"]
imagen = bench.build_imagen(1000, dev)
te = torch.randn(8, 256, 768, device=dev)
imagen.sample(text_embeds=te, cond_scale=3.0, use_tqdm=False, seed=1, max_steps=2, use_graph=False)
K_IGEMM = _abi.ENUMS["IMAGEN_OP_IGEMM"]
traces = []
for key, st in imagen._stages.items():
    for kind, p, label in st["plan"].ops:
        if kind == K_IGEMM and any(w in label for w in want) and not p.gate:
            t = torch.zeros(128, dtype=torch.int64, device=dev)
            p.gate = t.data_ptr()
            p.dbg = 128
            traces.append((key[0], label, p, t))
imagen.sample(text_embeds=te, cond_scale=3.0, use_tqdm=False, seed=1, max_steps=3, use_graph=False)
torch.cuda.synchronize()
for stage, label, p, t in traces:
    v = t.cpu().tolist()
    nz = [x for x in v if x]
    if not nz:
        print(f"stage {stage} {label}: no stamps")
        continue
    t0 = min(nz)
    cons = [x - t0 for x in v[:64] if x]
    prod = [x - t0 for x in v[64:] if x]
    print(f"stage {stage} {label}: {p.C1 + p.C2}->{p.Cout} k{p.KH} @{p.H}x{p.W} B{p.B} cfg{p.cfg} t{p.TH}x{p.TW}")
    print("   consumer:", cons[:40])
    print("   producer:", prod[:40], flush=True)
