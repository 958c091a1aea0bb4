"""BASELINE config C5 on one GPU: `Unet3D(dim = 64, dim_mults = (1, 2, 4, 8))` (README.md:587), one clip of 16 x 64 x 64, 250 DDPM
steps, CFG 3.  NOT YET RUN (the video path was written after round 1's GPU budget was spent): first run the opt-in parity tests,

    IMAGEN_VIDEO_GPU_TESTS=1 timeout 300 python -m pytest tests/test_video_gpu.py -m gpu -q

then this script (optionally with fewer steps: `python tools/time_c5.py 25`).  Prints seconds per clip and the per-step time; the
reference-count FLOPs are 326.7 GF per forward, i.e. 163 TFLOP per clip with CFG."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imagen_pytorch_amd import Imagen, Unet3D  # noqa: E402

dev = torch.device("cuda:0")
T = int(sys.argv[1]) if len(sys.argv) > 1 else 250
torch.manual_seed(0)
unet = Unet3D(dim=64, dim_mults=(1, 2, 4, 8))
imagen = Imagen((unet,), image_sizes=(64,), timesteps=T, cond_drop_prob=0.1)
for u in imagen.unets:
    torch.nn.init.normal_(u.final_conv.weight, std=0.05)
imagen = imagen.to(dev).eval()
te = torch.randn(1, 256, 768, device=dev)
os.environ.setdefault("IMAGEN_TIMING", "1")
imagen.sample(text_embeds=te, video_frames=16, cond_scale=3., use_tqdm=False, seed=1, max_steps=2)   # packs weights, captures the graph
torch.cuda.synchronize()
t0 = time.perf_counter()
clip = imagen.sample(text_embeds=te, video_frames=16, cond_scale=3., use_tqdm=False, seed=2)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
assert clip.shape == (1, 3, 16, 64, 64) and torch.isfinite(clip).all()
flops = 2 * T * 326.7e9
print(f"C5: {T} steps, CFG 3: {dt:.2f} s per clip ({dt / T * 1e3:.2f} ms per step), {flops / dt / 1e12:.1f} TFLOP/s by the reference count")
