"""BASELINE config C4 on ONE GPU's shard: ElucidatedImagen, README unet1 + unet2, 64 -> 256, 32 Karras steps (63 denoiser evaluations
per stage with the Heun correction), CFG 3, 4 images (= 32 / 8 GPUs).  Prints seconds per call and images/s; informational
(the bench contract's metric is C3)."""
import sys
import time

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import bench  # noqa: E402
from imagen_pytorch_amd import ElucidatedImagen, Unet  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)
u1, u2 = Unet(**bench.README_U1), Unet(**bench.README_U2)
model = ElucidatedImagen((u1, u2), image_sizes=(64, 256), num_sample_steps=32, cond_drop_prob=0.1)
for u in model.unets:
    torch.nn.init.normal_(u.final_conv.weight, std=0.05)
model = model.to(dev).eval()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
te = torch.randn(B, 256, 768, device=dev)
model.sample(text_embeds=te, cond_scale=3., use_tqdm=False, seed=1)      # warm-up: packs weights, captures the graphs
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 3
for i in range(n):
    img = model.sample(text_embeds=te, cond_scale=3., use_tqdm=False, seed=2 + i)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
assert img.shape == (B, 3, 256, 256) and torch.isfinite(img).all()
print(f"C4 shard: batch {B}, 32 steps/stage: {dt:.3f} s per call, {B / dt:.2f} images/s on one MI355X")
