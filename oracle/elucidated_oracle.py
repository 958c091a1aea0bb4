"""CPU oracle for the ElucidatedImagen sampler (TEST INFRASTRUCTURE — see oracle/unet_oracle.py header).

Restates, in fp32 torch on CPU, `ElucidatedImagen.sample / one_unet_sample / preconditioned_network_forward /
threshold_x_start / sample_schedule` (el.py = imagen_pytorch/elucidated_imagen.py :309-392, :393-545, :547-745) for
text_embeds-conditioned image sampling: Karras sigma schedule, stochastic churn, preconditioned denoiser
(c_in / c_noise / c_skip / c_out, Table 1 of Karras et al.), dynamic thresholding, second-order (Heun) correction,
low-res noise-conditioning augmentation, video stages (Unet3D) with prompt frames, and the one_unet_sample options init_images,
skip_steps and inpainting with resampling (el.py:446-452, 459-470, 477-479, 497-498, 532-545).  Self-conditioning is out of scope.

Gaussian noise is drawn through an injectable `noise_fn(tag, shape)` in the reference's call order per stage:
("lowres", stage) [el.py:705], ("init", stage) [el.py:442], ("step", stage, i) [el.py:489]; with inpainting every timestep runs
`inpaint_resample_times` times and the draws are ("step", stage, i, r) and ("renoise", stage, i, r) [el.py:489, 534].

Parity status: pinned against the live reference in tests/test_oracle_vs_reference.py (container only) and
tests/golden/sample_tiny_elucidated.pt (travels).
"""
from __future__ import annotations

import math
from typing import Callable, Optional, Sequence

import torch
import torch.nn.functional as F

from .sampler_oracle import SCHEDULES, alpha_sigma
from .unet3d_oracle import unet3d_forward_with_cond_scale
from .unet_oracle import unet_forward_with_cond_scale

Tensor = torch.Tensor

DEFAULT_HPARAMS = dict(num_sample_steps=32, sigma_min=0.002, sigma_max=80.0, sigma_data=0.5, rho=7.0, S_churn=80.0, S_tmin=0.05,
                       S_tmax=50.0, S_noise=1.003)   # el.py:100-110


def sample_schedule(num_sample_steps: int, rho: float, sigma_min: float, sigma_max: float) -> Tensor:
    """el.py:373-391 — Karras eq. (5) in fp32, with a trailing sigma = 0."""
    N = num_sample_steps
    inv_rho = 1 / rho
    steps = torch.arange(N, dtype=torch.float32)
    sigmas = (sigma_max ** inv_rho + steps / (N - 1) * (sigma_min ** inv_rho - sigma_max ** inv_rho)) ** rho
    return F.pad(sigmas, (0, 1), value=0.0)


def step_table(hp: dict):
    """el.py:428-436, 484: the per-step python floats (sigma, sigma_next, gamma)."""
    sigmas = sample_schedule(hp["num_sample_steps"], hp["rho"], hp["sigma_min"], hp["sigma_max"])
    gammas = torch.where((sigmas >= hp["S_tmin"]) & (sigmas <= hp["S_tmax"]),
                         min(hp["S_churn"] / hp["num_sample_steps"], math.sqrt(2) - 1), 0.0)
    return [(s.item(), sn.item(), g.item()) for s, sn, g in zip(sigmas[:-1], sigmas[1:], gammas[:-1])], sigmas[0].item()


def threshold_x_start(x_start: Tensor, dynamic_threshold: bool = True, percentile: float = 0.95) -> Tensor:
    """el.py:309-321."""
    if not dynamic_threshold:
        return x_start.clamp(-1.0, 1.0)
    s = torch.quantile(x_start.flatten(1).abs(), percentile, dim=-1)
    s.clamp_(min=1.0)
    s = s.view(-1, *([1] * (x_start.ndim - 1)))
    return x_start.clamp(-s, s) / s


def preconditioned_forward(net: Callable[[Tensor, Tensor], Tensor], noised: Tensor, sigma: float, sigma_data: float, *, clamp: bool,
                           dynamic_threshold: bool, percentile: float) -> Tensor:
    """el.py:340-369 — sigma is a python float broadcast to the batch (fp32)."""
    b = noised.shape[0]
    sig = torch.full((b,), sigma, dtype=torch.float32)
    ps = sig.view(-1, *([1] * (noised.ndim - 1)))
    c_in = 1 * (ps ** 2 + sigma_data ** 2) ** -0.5
    c_noise = torch.log(sig.clamp(min=1e-20)) * 0.25
    c_skip = (sigma_data ** 2) / (ps ** 2 + sigma_data ** 2)
    c_out = ps * sigma_data * (sigma_data ** 2 + ps ** 2) ** -0.5
    out = c_skip * noised + c_out * net(c_in * noised, c_noise)
    return threshold_x_start(out, dynamic_threshold, percentile) if clamp else out


def one_unet_sample(net: Callable[[Tensor, Tensor], Tensor], shape, hp: dict, *, noise_fn: Callable, stage: int, clamp: bool = True,
                    dynamic_threshold: bool = True, percentile: float = 0.95, max_steps: Optional[int] = None,
                    init_images: Optional[Tensor] = None, skip_steps: Optional[int] = None, inpaint_images: Optional[Tensor] = None,
                    inpaint_masks: Optional[Tensor] = None, inpaint_resample_times: int = 5) -> Tensor:
    """el.py:393-545 without self-conditioning.  Returns the UNNORMALISED [0, 1] image.  init_images / inpaint_images arrive
    normalised to [-1, 1] at any resolution (resized here like el.py:466-467 and the caller's resize of init images, el.py:717);
    inpaint_masks (B, [F,] H, W) bool, True = keep the known pixel."""
    table, init_sigma = step_table(hp)
    if len(shape) == 5:   # resize_video_to with target_frames (el.py:415-417)
        resize = lambda im: im if tuple(im.shape[-3:]) == tuple(shape[-3:]) else F.interpolate(im, tuple(shape[-3:]), mode="nearest")
    else:
        resize = lambda im: im if im.shape[-1] == shape[-1] else F.interpolate(im, shape[-1], mode="nearest")
    images = init_sigma * noise_fn(("init", stage), shape)             # el.py:440-442: always sigmas[0], also when steps are skipped
    if init_images is not None:
        images = images + resize(init_images)                           # el.py:446-447
    inpainting = inpaint_images is not None and inpaint_masks is not None
    R = inpaint_resample_times if inpainting else 1
    if inpainting:
        known = resize(inpaint_images)
        mask = resize(inpaint_masks[:, None].float()).bool()
    first = skip_steps or 0                                             # el.py:477-479
    table = list(enumerate(table))[first:]
    for n_done, (ind, (sigma, sigma_next, gamma)) in enumerate(table):
        if max_steps is not None and n_done >= max_steps:
            break
        is_last = n_done == len(table) - 1
        for r in reversed(range(R)):
            eps = hp["S_noise"] * noise_fn(("step", stage, ind, r) if inpainting else ("step", stage, ind), shape)
            sigma_hat = sigma + gamma * sigma
            added_noise = math.sqrt(sigma_hat ** 2 - sigma ** 2) * eps
            images_hat = images + added_noise
            if inpainting:
                images_hat = images_hat * ~mask + (known + added_noise) * mask      # el.py:497-498
            kw = dict(clamp=clamp, dynamic_threshold=dynamic_threshold, percentile=percentile)
            model_output = preconditioned_forward(net, images_hat, sigma_hat, hp["sigma_data"], **kw)
            denoised_over_sigma = (images_hat - model_output) / sigma_hat
            images_next = images_hat + (sigma_next - sigma_hat) * denoised_over_sigma
            if sigma_next != 0:   # second-order correction
                model_output_next = preconditioned_forward(net, images_next, sigma_next, hp["sigma_data"], **kw)
                denoised_prime_over_sigma = (images_next - model_output_next) / sigma_next
                images_next = images_hat + 0.5 * (sigma_next - sigma_hat) * (denoised_over_sigma + denoised_prime_over_sigma)
            images = images_next
            if inpainting and not (r == 0 or is_last):                  # el.py:532-535
                images = images + (sigma - sigma_next) * noise_fn(("renoise", stage, ind, r), shape)
    images = images.clamp(-1.0, 1.0)
    if inpainting:
        images = images * ~mask + known * mask                          # el.py:542-543
    return (images + 1) * 0.5


def elucidated_sample(unets: Sequence[tuple], image_sizes: Sequence[int], text_embeds: Tensor, *, hparams: Optional[dict] = None,
                      cond_scale=1.0, lowres_noise_schedule: str = "linear", lowres_sample_noise_level: float = 0.2,
                      dynamic_thresholding: bool = True, percentile: float = 0.95, channels: int = 3, text_masks: Optional[Tensor] = None,
                      noise_fn: Optional[Callable] = None, max_steps: Optional[int] = None, return_all: bool = False,
                      video_frames: Optional[int] = None, cond_video_frames: Optional[Tensor] = None,
                      post_cond_video_frames: Optional[Tensor] = None, init_images=None, skip_steps=None,
                      inpaint_images: Optional[Tensor] = None, inpaint_masks: Optional[Tensor] = None, inpaint_resample_times: int = 5,
                      sigma_min=None, sigma_max=None):
    """el.py:547-745.  sigma_min / sigma_max: per-call overrides of the schedule's end points, one value or one per unet (el.py:647-648,
    425-426).    `unets`: [(state_dict, ctor_kwargs), ...]; `hparams`: overrides of DEFAULT_HPARAMS (same for every stage).
    video_frames: the unets are Unet3D state_dicts and every stage samples (b, c, video_frames, h, w) clips; the prompt frames
    (el.py:679-695) are handed to every stage as they are (temporal_downsample_factor 1 only here)."""
    n = len(unets)
    hp = dict(DEFAULT_HPARAMS, **(hparams or {}))
    cond_scale = cond_scale if isinstance(cond_scale, (list, tuple)) else (cond_scale,) * n
    if noise_fn is None:
        noise_fn = lambda tag, shape: torch.randn(shape)
    if text_masks is None:
        text_masks = torch.any(text_embeds != 0.0, dim=-1)   # el.py:591
    b = text_embeds.shape[0]
    as_tuple = lambda v: tuple(v) if isinstance(v, (list, tuple)) else (v,) * n
    init_images = [None if im is None else im * 2 - 1 for im in as_tuple(init_images)]   # normalize_img, el.py:651
    skip_steps = as_tuple(skip_steps)
    known = None if inpaint_images is None else inpaint_images * 2 - 1                   # el.py:466
    if video_frames is not None and inpaint_masks is not None and inpaint_masks.ndim == 3:   # el.py:634-635
        inpaint_masks = inpaint_masks[:, None].expand(-1, video_frames, -1, -1)
    outputs, img = [], None
    for stage, ((sd, kw), size, cs) in enumerate(zip(unets, image_sizes, cond_scale)):
        lowres_img = lowres_times = None
        if kw.get("lowres_cond", False):
            lowres_times = torch.full((b,), lowres_sample_noise_level, dtype=torch.float32)   # el.py:700 — passed on RAW (:728)
            if video_frames is not None:
                up = img if img.shape[-1] == size else F.interpolate(img, (img.shape[2], size, size), mode="nearest")
            else:
                up = img if img.shape[-1] == size else F.interpolate(img, size, mode="nearest")
            up = up * 2 - 1
            a, s = alpha_sigma(SCHEDULES[lowres_noise_schedule](lowres_times).reshape(-1, *([1] * (up.ndim - 1))))
            lowres_img = a * up + s * noise_fn(("lowres", stage), up.shape)                  # el.py:705

        fwd = unet3d_forward_with_cond_scale if video_frames is not None else unet_forward_with_cond_scale

        video_kw = {k: v for k, v in (("cond_video_frames", cond_video_frames), ("post_cond_video_frames", post_cond_video_frames))
                    if v is not None and video_frames is not None}

        def net(x, c_noise, _sd=sd, _kw=kw, _cs=cs, _li=lowres_img, _lt=lowres_times):
            return fwd(_sd, _kw, x, c_noise, cond_scale=_cs, text_embeds=text_embeds, text_mask=text_masks, lowres_cond_img=_li,
                       lowres_noise_times=_lt, **video_kw)

        shape = (b, channels, video_frames, size, size) if video_frames is not None else (b, channels, size, size)
        hp_s = dict(hp)
        for name, v in (("sigma_min", sigma_min), ("sigma_max", sigma_max)):
            v = as_tuple(v)[stage]
            if v is not None:
                hp_s[name] = v
        img = one_unet_sample(net, shape, hp_s, noise_fn=noise_fn, stage=stage, dynamic_threshold=dynamic_thresholding,
                              percentile=percentile, max_steps=max_steps, init_images=init_images[stage], skip_steps=skip_steps[stage],
                              inpaint_images=known, inpaint_masks=inpaint_masks, inpaint_resample_times=inpaint_resample_times)
        outputs.append(img)
    return outputs if return_all else outputs[-1]
