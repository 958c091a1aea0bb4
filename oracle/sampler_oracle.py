"""CPU oracle for the cascaded-DDPM sampling loop (TEST INFRASTRUCTURE — see oracle/unet_oracle.py header).

Restates, in fp32 torch on CPU, the continuous-time Gaussian diffusion helpers
(ip.py:212-318) and the ancestral sampler `Imagen.p_mean_variance / p_sample /
p_sample_loop / sample` (ip.py:2042-2498) for the options the BASELINE configs use:
noise-prediction objective, dynamic thresholding, classifier-free guidance, low-res
noise-conditioning augmentation, plus the p_sample_loop options init_images, skip_steps
and inpainting with resampling (ip.py:2205-2206, 2228-2229, 2237-2286), `cond_images`,
self-conditioning unets, and for video stages `cond_video_frames` / `post_cond_video_frames`
(ip.py:2417-2434) and inpaint_videos / init videos (resize over frames too, ip.py:2196-2220).

All Gaussian noise is drawn through an injectable `noise_fn(tag, shape)` so the HIP
path and the reference can be fed identical tensors (CPU and GPU RNG streams differ,
SURVEY §7.3-8).  Tags: ("init", stage), ("lowres", stage), ("step", stage, step_index);
with inpainting ("inpaint" | "step" | "renoise", stage, step_index, resample_index).

Parity status: pinned against the live reference in tests/test_oracle_vs_reference.py
(container only) and tests/golden/sample_*.pt (travels; sample_tiny_options.pt holds the
init_images/skip_steps and inpainting runs).
"""
from __future__ import annotations

import math
from typing import Callable, List, Optional, Sequence

import torch
import torch.nn.functional as F

from .unet3d_oracle import unet3d_forward_with_cond_scale
from .unet_oracle import unet_forward_with_cond_scale

Tensor = torch.Tensor


# ---------------------------------------------------------------- schedules (ip.py:212-221)

def log_snr_linear(t: Tensor) -> Tensor:
    """ip.py:212-214."""
    return -torch.log(torch.special.expm1(1e-4 + 10 * (t ** 2)))


def log_snr_cosine(t: Tensor, s: float = 0.008) -> Tensor:
    """ip.py:216-218 (log() there clamps at eps=1e-5)."""
    return -torch.log(((torch.cos((t + s) / (1 + s) * math.pi * 0.5) ** -2) - 1).clamp(min=1e-5))


SCHEDULES = {"linear": log_snr_linear, "cosine": log_snr_cosine}


def alpha_sigma(log_snr: Tensor):
    """ip.py:220-221."""
    return torch.sqrt(torch.sigmoid(log_snr)), torch.sqrt(torch.sigmoid(-log_snr))


def sampling_time_pairs(num_timesteps: int) -> List[tuple]:
    """ip.py:245-250 — consecutive pairs of linspace(1, 0, T+1) as python floats held in fp32."""
    times = torch.linspace(1.0, 0.0, num_timesteps + 1)
    return [(times[i], times[i + 1]) for i in range(num_timesteps)]


def default_noise_schedules(num_unets: int, given="cosine") -> List[str]:
    """ip.py:1853-1855 — pad to ('cosine','cosine') then 'linear' for the rest."""
    g = list(given) if isinstance(given, (list, tuple)) else [given]
    while len(g) < 2:
        g.append("cosine")
    while len(g) < num_unets:
        g.append("linear")
    return g[:num_unets] if len(g) > num_unets else g


# ---------------------------------------------------------------- one DDPM step (ip.py:2042-2165)

def dynamic_threshold(x0: Tensor, percentile: float = 0.95) -> Tensor:
    """ip.py:2094-2105."""
    s = torch.quantile(x0.reshape(x0.shape[0], -1).abs(), percentile, dim=-1).clamp(min=1.0)
    s = s.reshape(-1, *([1] * (x0.ndim - 1)))
    return x0.clamp(-s, s) / s


def ddpm_step(x: Tensor, pred_noise: Tensor, t: Tensor, t_next: Tensor, noise: Tensor, schedule: str,
              dynamic_thresholding: bool = True, percentile: float = 0.95):
    """x_t, eps_hat -> x_{t_next}.  t, t_next: (B,) fp32.  ip.py:314-318, 2094-2109, 252-270, 2160-2164."""
    fn = SCHEDULES[schedule]
    pad = lambda v: v.reshape(-1, *([1] * (x.ndim - 1)))
    log_snr, log_snr_next = pad(fn(t)), pad(fn(t_next))
    alpha, sigma = alpha_sigma(log_snr)
    alpha_next, sigma_next = alpha_sigma(log_snr_next)
    x0 = (x - sigma * pred_noise) / alpha.clamp(min=1e-8)
    x0 = dynamic_threshold(x0, percentile) if dynamic_thresholding else x0.clamp(-1.0, 1.0)
    c = -torch.special.expm1(log_snr - log_snr_next)
    mean = alpha_next * (x * (1 - c) / alpha + c * x0)
    var = (sigma_next ** 2) * c
    log_var = torch.log(var.clamp(min=1e-20))
    nonzero = pad(1.0 - (t_next == 0).float())
    return mean + nonzero * (0.5 * log_var).exp() * noise, x0


# ---------------------------------------------------------------- loops (ip.py:2167-2289, 2291-2498)

def q_sample_from_to(x: Tensor, t_from: Tensor, t_to: Tensor, noise: Tensor, schedule: str) -> Tensor:
    """ip.py:286-307."""
    fn = SCHEDULES[schedule]
    pad = lambda v: v.reshape(-1, *([1] * (x.ndim - 1)))
    alpha, sigma = alpha_sigma(pad(fn(t_from)))
    alpha_to, sigma_to = alpha_sigma(pad(fn(t_to)))
    return x * (alpha_to / alpha) + noise * (sigma_to * alpha - sigma * alpha_to) / alpha


def p_sample_loop(denoise: Callable[[Tensor, Tensor], Tensor], shape, *, schedule: str, num_timesteps: int,
                  noise_fn: Callable, stage: int, dynamic_thresholding: bool = True, percentile: float = 0.95,
                  max_steps: Optional[int] = None, trace: Optional[list] = None, init_images: Optional[Tensor] = None,
                  skip_steps: Optional[int] = None, inpaint_images: Optional[Tensor] = None, inpaint_masks: Optional[Tensor] = None,
                  inpaint_resample_times: int = 5, self_cond: bool = False, resize_mode: str = "nearest") -> Tensor:
    """denoise(x_t, log_snr(t)) -> guided eps_hat; with `self_cond` the call is denoise(x_t, log_snr(t), x0_prev) where x0_prev is the
    thresholded x0 estimate of the previous call (None before the first one; ip.py:2208-2210, 2249-2251).  Returns the un-normalised image in [0, 1] (ip.py:2167-2289).

    init_images / inpaint_images arrive NORMALISED to [-1, 1] and at any resolution (resized here, ip.py:2219-2220, 2457);
    inpaint_masks: (B, H, W) bool, True = keep the known pixel.  With inpainting every timestep is run `inpaint_resample_times`
    times (RePaint); the draws are tagged ("inpaint", stage, i, r), ("step", stage, i, r), ("renoise", stage, i, r) in the
    reference's call order; without it the step draw keeps its 3-tuple tag ("step", stage, i)."""
    b, size = shape[0], shape[-1]
    if len(shape) == 5:     # video: resize_video_to with target_frames = this stage's frame count (ip.py:2198-2200, iv.py:134-156)
        resize = lambda im: im if tuple(im.shape[-3:]) == tuple(shape[-3:]) else F.interpolate(im, tuple(shape[-3:]), mode="nearest")
    else:
        resize = lambda im: im if im.shape[-1] == size else F.interpolate(im, size, mode=resize_mode)   # ip.py:152-168, 1924
    img = noise_fn(("init", stage), shape)
    if init_images is not None:
        img = img + resize(init_images)                                     # ip.py:2205-2206 (+ the resize of :2457)
    inpainting = inpaint_images is not None and inpaint_masks is not None
    R = inpaint_resample_times if inpainting else 1
    if inpainting:
        known = resize(inpaint_images)
        mask = resize(inpaint_masks[:, None].float()).bool()               # ip.py:2220
    fn = SCHEDULES[schedule]
    pairs = sampling_time_pairs(num_timesteps)
    x_start = None
    first = skip_steps or 0                                                 # ip.py:2228-2229
    if max_steps is not None:
        pairs = pairs[:first + max_steps]
    for i, (tt, tn) in list(enumerate(pairs))[first:]:
        t = tt.expand(b).clone()
        t_next = tn.expand(b).clone()
        last_t = bool((t_next == 0).all())
        for r in reversed(range(R)):
            if inpainting:
                a, s_ = alpha_sigma(fn(t).reshape(-1, *([1] * (len(shape) - 1))))
                noised = a * known + s_ * noise_fn(("inpaint", stage, i, r), shape)       # ip.py:2244-2246
                img = img * ~mask + noised * mask
            pred = denoise(img, fn(t), x_start) if self_cond else denoise(img, fn(t))
            tag = ("step", stage, i, r) if inpainting else ("step", stage, i)
            img, x_start = ddpm_step(img, pred, t, t_next, noise_fn(tag, shape), schedule, dynamic_thresholding, percentile)
            if inpainting and not (r == 0 or last_t):
                img = q_sample_from_to(img, t_next, t, noise_fn(("renoise", stage, i, r), shape), schedule)   # ip.py:2268-2275
            if trace is not None:
                trace.append(img.clone())
    img = img.clamp(-1.0, 1.0)
    if inpainting:
        img = img * ~mask + known * mask                                    # ip.py:2283-2286
    return (img + 1) * 0.5


def imagen_sample(
    unets: Sequence[tuple],           # [(state_dict, ctor_kwargs), ...] — kwargs already carry lowres_cond etc.
    image_sizes: Sequence[int],
    text_embeds: Tensor,
    *,
    timesteps=1000,
    cond_scale=1.0,
    noise_schedules="cosine",
    lowres_noise_schedule: str = "linear",
    lowres_sample_noise_level: float = 0.2,
    dynamic_thresholding: bool = True,
    percentile: float = 0.95,
    channels: int = 3,
    text_masks: Optional[Tensor] = None,
    noise_fn: Optional[Callable] = None,
    max_steps: Optional[int] = None,
    return_all: bool = False,
    init_images=None,                 # one [0, 1] image batch (or None) per unet, or a single batch for all   (ip.py:2390-2393)
    skip_steps=None,                  # int (or None) per unet
    inpaint_images: Optional[Tensor] = None,   # [0, 1] images, same for every stage
    inpaint_masks: Optional[Tensor] = None,    # (B, H, W) bool; videos: (B, F, H, W), or (B, H, W) repeated over the frames (ip.py:2376)
    inpaint_resample_times: int = 5,
    video_frames: Optional[int] = None,        # Imagen-Video: the unets are Unet3D state_dicts, samples are (b, c, f, h, w)
    temporal_downsample_factor=1,              # per stage: stage i samples video_frames // factor[i] frames (ip.py:170-183, 1928-1935)
    cond_images: Optional[Tensor] = None,      # (B, cond_images_channels, h, w) in [0, 1], handed to every unet as is (ip.py:2324, 2465)
    cond_video_frames: Optional[Tensor] = None,        # (b, c, f', h, w) prompt frames, handed to every Unet3D as they are — not
    post_cond_video_frames: Optional[Tensor] = None,   # normalised — after the per-stage temporal resize (ip.py:2417-2434)
    resize_cond_video_frames: bool = True,
    resize_mode: str = "nearest",              # Imagen(resize_mode=...): every image resize of the cascade (ip.py:1924); images only here
):
    """ip.py:2291-2498 for text_embeds-conditioned sampling (no self-conditioning)."""
    n = len(unets)
    timesteps = timesteps if isinstance(timesteps, (list, tuple)) else (timesteps,) * n
    cond_scale = cond_scale if isinstance(cond_scale, (list, tuple)) else (cond_scale,) * n
    schedules = default_noise_schedules(n, noise_schedules)
    if noise_fn is None:
        noise_fn = lambda tag, shape: torch.randn(shape)
    if text_masks is None:
        text_masks = torch.any(text_embeds != 0.0, dim=-1)  # ip.py:2337
    b = text_embeds.shape[0]
    video = video_frames is not None          # Imagen-Video: every unet is a Unet3D, samples are (b, c, f, h, w)  (ip.py:1918, 2381-2383)
    tds = temporal_downsample_factor if isinstance(temporal_downsample_factor, (list, tuple)) else (temporal_downsample_factor,) * n
    as_tuple = lambda v: tuple(v) if isinstance(v, (list, tuple)) else (v,) * n
    init_images = [None if im is None else im * 2 - 1 for im in as_tuple(init_images)]   # normalize_img, ip.py:2391
    skip_steps = as_tuple(skip_steps)
    known = None if inpaint_images is None else inpaint_images * 2 - 1                   # ip.py:2218
    if video and inpaint_masks is not None and inpaint_masks.ndim == 3:                  # ip.py:2376-2377
        inpaint_masks = inpaint_masks[:, None].expand(-1, video_frames, -1, -1)
    outputs, img = [], None
    for stage, ((sd, kw), size, T, cs, sched) in enumerate(zip(unets, image_sizes, timesteps, cond_scale, schedules)):
        lowres_img = lowres_times = None
        if kw.get("lowres_cond", False):
            lowres_times = torch.full((b,), lowres_sample_noise_level, dtype=torch.float32)
            if video:   # resize_video_to (iv.py:134-156): nearest over (f, h, w) to this stage's frame count and size
                target = (video_frames // tds[stage], size, size)
                up = img if tuple(img.shape[-3:]) == target else F.interpolate(img, target, mode="nearest")
            else:
                up = img if img.shape[-1] == size else F.interpolate(img, size, mode=resize_mode)  # ip.py:152-168
            up = up * 2 - 1
            a, s = alpha_sigma(SCHEDULES[lowres_noise_schedule](lowres_times).reshape(-1, *([1] * (up.ndim - 1))))
            lowres_img = a * up + s * noise_fn(("lowres", stage), up.shape)       # ip.py:272-284, 2449
            lowres_logsnr = SCHEDULES[lowres_noise_schedule](lowres_times)       # ip.py:2081

        fwd = unet3d_forward_with_cond_scale if video else unet_forward_with_cond_scale
        video_kw = {}
        for name, v in (("cond_video_frames", cond_video_frames), ("post_cond_video_frames", post_cond_video_frames)):
            if video and v is not None:
                if resize_cond_video_frames and tds[stage] != 1:   # scale_video_time (iv.py:158-178): nearest over the frame axis
                    assert v.shape[2] % tds[stage] == 0
                    v = F.interpolate(v, (v.shape[2] // tds[stage], v.shape[-2], v.shape[-1]), mode="nearest")
                video_kw[name] = v

        def denoise(x, log_snr, x0_prev=None, _sd=sd, _kw=kw, _cs=cs, _li=lowres_img, _lt=(lowres_logsnr if lowres_img is not None else None),
                    _vk=video_kw):
            extra = dict(_vk) if cond_images is None else dict(_vk, cond_images=cond_images)
            if _kw.get("self_cond", False):
                extra["self_cond"] = x0_prev
            return fwd(_sd, _kw, x, log_snr, cond_scale=_cs, text_embeds=text_embeds, text_mask=text_masks, lowres_cond_img=_li,
                       lowres_noise_times=_lt, **extra)

        shape = (b, channels, video_frames // tds[stage], size, size) if video else (b, channels, size, size)
        img = p_sample_loop(denoise, shape, schedule=sched, num_timesteps=T, noise_fn=noise_fn,
                            stage=stage, dynamic_thresholding=dynamic_thresholding, percentile=percentile,
                            max_steps=max_steps, init_images=init_images[stage], skip_steps=skip_steps[stage], inpaint_images=known,
                            inpaint_masks=inpaint_masks, inpaint_resample_times=inpaint_resample_times,
                            self_cond=bool(kw.get("self_cond", False)), resize_mode=resize_mode)
        outputs.append(img)
    return outputs if return_all else outputs[-1]
