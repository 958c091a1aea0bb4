"""Drop-in `ElucidatedImagen` (Karras et al. EDM sampler over the same cascaded unets): constructor and `.sample()` signature of the
reference (imagen_pytorch/elucidated_imagen.py:76-745 = "el.py"), sampling half only.  SURVEY.md §8(f) NEXT-1 / BASELINE config C4.

A sampling step is [x_hat = x + churn noise, c_in * x_hat] -> denoiser plan (both CFG branches as one 2B-row batch) -> [CFG
combine + c_skip / c_out preconditioning -> exact 0.95-quantile -> Euler step, c_in * x_next] -> denoiser plan again ->
[preconditioning -> quantile -> Heun combination], all HIP kernels (LINCOMB / CFG_X0 / QUANTILE ops), captured ONCE into a
hipGraph and replayed for every step but the last (which has no second-order correction and its own graph).  The per-evaluation
scalars (c_skip, c_out, c_in, c_noise, step weights) live in device tables indexed by a device counter, so replay needs no host
patching.  Preconditioning is expressed through the existing CFG_X0 "noise" form: c_skip*x + c_out*F = (x - sigma'*F) / alpha'
with alpha' = 1/c_skip, sigma' = -c_out/c_skip.

Training (`forward`), T5 text encoding, video, inpainting, init_images / skip_steps and per-call sigma_min / sigma_max overrides are
outside the hot-path scope and raise.
"""
from __future__ import annotations

import math
from collections import namedtuple
from typing import Callable, List, Optional

import torch

from . import ops
from .unet3d import Unet3D
from .imagen import DEFAULT_T5_NAME, TAG_INIT, Imagen, _cast_tuple, _out_of_scope
from .ops import Plan

Hparams_fields = ['num_sample_steps', 'sigma_min', 'sigma_max', 'sigma_data', 'rho', 'P_mean', 'P_std', 'S_churn', 'S_tmin', 'S_tmax', 'S_noise']
Hparams = namedtuple('Hparams', Hparams_fields)


class ElucidatedImagen(Imagen):
    def __init__(
        self,
        unets,
        *,
        image_sizes,
        text_encoder_name=DEFAULT_T5_NAME,
        text_embed_dim=None,
        channels=3,
        cond_drop_prob=0.1,
        random_crop_sizes=None,
        resize_mode='nearest',
        temporal_downsample_factor=1,
        resize_cond_video_frames=True,
        lowres_sample_noise_level=0.2,
        per_sample_random_aug_noise_level=False,
        condition_on_text=True,
        auto_normalize_img=True,
        dynamic_thresholding=True,
        dynamic_thresholding_percentile=0.95,
        only_train_unet_number=None,
        lowres_noise_schedule='linear',
        num_sample_steps=32,
        sigma_min=0.002,
        sigma_max=80,
        sigma_data=0.5,
        rho=7,
        P_mean=-1.2,
        P_std=1.2,
        S_churn=80,
        S_tmin=0.05,
        S_tmax=50,
        S_noise=1.003,
    ):
        num_unets = len(_cast_tuple(unets))
        steps = _cast_tuple(num_sample_steps, num_unets)
        super().__init__(unets, image_sizes=image_sizes, text_encoder_name=text_encoder_name, text_embed_dim=text_embed_dim, channels=channels,
                         timesteps=steps, cond_drop_prob=cond_drop_prob, random_crop_sizes=random_crop_sizes,
                         lowres_noise_schedule=lowres_noise_schedule, lowres_sample_noise_level=lowres_sample_noise_level,
                         per_sample_random_aug_noise_level=per_sample_random_aug_noise_level, condition_on_text=condition_on_text,
                         auto_normalize_img=auto_normalize_img, dynamic_thresholding=dynamic_thresholding,
                         dynamic_thresholding_percentile=dynamic_thresholding_percentile, only_train_unet_number=only_train_unet_number,
                         temporal_downsample_factor=temporal_downsample_factor, resize_cond_video_frames=resize_cond_video_frames,
                         resize_mode=resize_mode)
        hparams = [num_sample_steps, sigma_min, sigma_max, sigma_data, rho, P_mean, P_std, S_churn, S_tmin, S_tmax, S_noise]
        hparams = [_cast_tuple(hp, num_unets) for hp in hparams]
        self.hparams = [Hparams(*unet_hp) for unet_hp in zip(*hparams)]    # el.py:233-236
        self._lowres_time_raw = True    # el.py:700, 728: the raw augmentation level conditions the unet at sample time

    # ---- schedule (el.py:373-391, 428-436) -------------------------------------------------------------------------------
    @staticmethod
    def sample_schedule(num_sample_steps, rho, sigma_min, sigma_max):
        N = num_sample_steps
        inv_rho = 1 / rho
        steps = torch.arange(N, dtype=torch.float32)
        sigmas = (sigma_max ** inv_rho + steps / (N - 1) * (sigma_min ** inv_rho - sigma_max ** inv_rho)) ** rho
        return torch.nn.functional.pad(sigmas, (0, 1), value=0.)

    def _tables(self, hp: Hparams):
        """Per-evaluation device tables (row 2i: first evaluation of step i at sigma_hat, row 2i+1: second at sigma_next), fp32
        [2N, 8] each: `coef` (CFG_X0 / time embedding: 1/c_skip, -c_out/c_skip, ..., col 6 = c_noise), `w_hat` (x_hat op),
        `w_euler`, `w_heun` (LINCOMB weights w0..w5)."""
        sigmas = self.sample_schedule(hp.num_sample_steps, hp.rho, hp.sigma_min, hp.sigma_max)
        gammas = torch.where((sigmas >= hp.S_tmin) & (sigmas <= hp.S_tmax), min(hp.S_churn / hp.num_sample_steps, math.sqrt(2) - 1), 0.)
        N = hp.num_sample_steps
        sd = hp.sigma_data
        coef = torch.zeros(2 * N, 8, dtype=torch.float64)
        w_hat, w_euler, w_heun = (torch.zeros(2 * N, 8, dtype=torch.float64) for _ in range(3))

        def precond(row, sigma):   # el.py:323-336
            c_skip = sd ** 2 / (sigma ** 2 + sd ** 2)
            c_out = sigma * sd * (sd ** 2 + sigma ** 2) ** -0.5
            coef[row, 0], coef[row, 1], coef[row, 6] = 1.0 / c_skip, -c_out / c_skip, math.log(max(sigma, 1e-20)) * 0.25

        for i in range(N):
            sigma, sigma_next, gamma = sigmas[i].item(), sigmas[i + 1].item(), gammas[i].item()
            sigma_hat = sigma + gamma * sigma
            precond(2 * i, sigma_hat)
            w_hat[2 * i, 0] = 1.0
            w_hat[2 * i, 4] = math.sqrt(sigma_hat ** 2 - sigma ** 2) * hp.S_noise          # el.py:489-492
            w_hat[2 * i, 5] = (sigma_hat ** 2 + sd ** 2) ** -0.5                              # c_in(sigma_hat)
            r = sigma_next / sigma_hat
            w_euler[2 * i, 0], w_euler[2 * i, 1] = r, 1.0 - r                                 # x_hat + (s_n - s_h)(x_hat - x0)/s_h
            if sigma_next != 0:
                precond(2 * i + 1, sigma_next)
                w_euler[2 * i, 5] = (sigma_next ** 2 + sd ** 2) ** -0.5                       # c_in(sigma_next)
                d = 0.5 * (sigma_next - sigma_hat)
                w_heun[2 * i + 1, 0], w_heun[2 * i + 1, 1] = 1.0 + d / sigma_hat, -d / sigma_hat
                w_heun[2 * i + 1, 2], w_heun[2 * i + 1, 3] = d / sigma_next, -d / sigma_next  # el.py:528-529
        return sigmas[0].item(), [t.float().contiguous() for t in (coef, w_hat, w_euler, w_heun)]

    # ---- per-stage plans --------------------------------------------------------------------------------------------------
    def _build_stage(self, idx: int, B: int, device, *, cond_scale: float, with_text: bool, inject_noise: bool, sample_offset: int,
                     resample_times: int = 0, frames: int = 0, prompt_frames: tuple = (0, 0)):
        unet = self.unets[idx]
        if getattr(unet, 'self_cond', False):
            from .imagen import _out_of_scope
            _out_of_scope("ElucidatedImagen sampling with self-conditioning unets (el.py:496, 518)")
        S = self.image_sizes[idx]
        hp = self.hparams[idx]
        cfg = cond_scale != 1.
        key = ("edm", idx, B, S, str(device), float(cond_scale), with_text, inject_noise, sample_offset, self.dynamic_thresholding[idx],
               self.dynamic_thresholding_percentile, tuple(hp), frames, prompt_frames, self._lane)
        st = self._stages.get(key)
        if st is not None and not st['eng'].stale():
            return st
        rows = 2 * B if cfg else B
        video = isinstance(unet, Unet3D)
        if video:      # as in Imagen._stage: the state is the engine's frame-major clip, every update below is elementwise per sample
            assert frames > 0, 'video_frames must be passed in on sample time if training on video'
            from . import engine3d
            eng = engine3d.UnetEngine3D(unet, rows, B, frames, S, device, with_text=with_text, pre_frames=prompt_frames[0],
                                        post_frames=prompt_frames[1])
        else:
            from . import engine
            eng = engine.UnetEngine(unet, rows, B, S, device, with_text=with_text)
        n = eng.x_in[0].numel()
        dev = device
        init_sigma, (coef, w_hat, w_euler, w_heun) = self._tables(hp)
        coef, w_hat, w_euler, w_heun = (t.to(dev) for t in (coef, w_hat, w_euler, w_heun))
        if inject_noise:   # the churn noise comes in through t1 (weight col 1) instead of the in-kernel Philox stream (col 4)
            w_hat[:, 1] = w_hat[:, 4]
            w_hat[:, 4] = 0
        step_ptr = torch.zeros(1, dtype=torch.int32, device=dev)
        seed_dev = torch.zeros(2, dtype=torch.int32, device=dev)
        eng.bind_step_counter(coef, step_ptr)
        mk = lambda: torch.empty_like(eng.x_in)
        x, xhat, xnext, x0a, x0b, absx0, final = mk(), mk(), mk(), mk(), mk(), mk(), mk()
        qa, qb = torch.empty(B, device=dev), torch.empty(B, device=dev)
        W = ops.ENUMS["IMAGEN_QUANTILE_SCRATCH_WORDS"]
        scr_a, scr_b = (torch.empty(B * W, dtype=torch.int32, device=dev) for _ in range(2))
        noise = mk() if inject_noise else None
        dyn = bool(self.dynamic_thresholding[idx])
        thr = 1 if dyn else 2                      # clamp=True (el.py:399): dynamic threshold, else clamp to [-1, 1]
        q = float(self.dynamic_thresholding_percentile)
        kw = dict(B=B, n_per_sample=n, stream_id=idx, sample_offset=sample_offset, seed_ptr=seed_dev)

        def first_eval(plan):
            ops.lincomb(plan, x, xhat, w_hat, step_ptr, t1=noise, out2=eng.x_in, label="edm.x_hat", **kw)
            plan.extend(eng.step_plan)
            ops.cfg_x0(plan, xhat, eng.out, coef, step_ptr, x0a, absx0, B=B, n_per_sample=n, cfg=cfg, cond_scale=float(cond_scale),
                       objective="noise", label="edm.precond")
            if dyn:
                ops.quantile(plan, absx0, qa, scr_a, B=B, n=n, q=q)

        full = Plan(f"edm-stage{idx}-step")
        first_eval(full)
        ops.lincomb(full, xhat, xnext, w_euler, step_ptr, t1=x0a, q1=qa if dyn else None, out2=eng.x_in, thr_mode=thr, advance=True,
                    label="edm.euler", **kw)
        full.extend(eng.step_plan)
        ops.cfg_x0(full, xnext, eng.out, coef, step_ptr, x0b, absx0, B=B, n_per_sample=n, cfg=cfg, cond_scale=float(cond_scale),
                   objective="noise", label="edm.precond2")
        if dyn:
            ops.quantile(full, absx0, qb, scr_b, B=B, n=n, q=q)
        ops.lincomb(full, xhat, x, w_heun, step_ptr, t1=x0a, t2=xnext, t3=x0b, q1=qa if dyn else None, q3=qb if dyn else None,
                    thr_mode=thr, advance=True, label="edm.heun", **kw)

        last = Plan(f"edm-stage{idx}-last")       # sigma_next = 0: Euler step only, then clamp + unnormalise (el.py:515, 540-545)
        first_eval(last)
        ops.lincomb(last, xhat, x, w_euler, step_ptr, t1=x0a, q1=qa if dyn else None, thr_mode=thr, final=True, final_out=final,
                    advance=True, label="edm.euler.final", **kw)

        w_init = torch.zeros(1, 8, device=dev)
        w_init[0, 0] = init_sigma
        zero_ptr = torch.zeros(1, dtype=torch.int32, device=dev)
        st = dict(eng=eng, plan=full, last=last, graph=None, graph_last=None, coef=coef, step_ptr=step_ptr, seed_dev=seed_dev, noise=noise,
                  final=final, T=hp.num_sample_steps, S=S, x=x, w_init=w_init, zero_ptr=zero_ptr, tables=(w_hat, w_euler, w_heun),
                  video=video, frames=frames,
                  bufs=(xhat, xnext, x0a, x0b, absx0, qa, qb, scr_a, scr_b))
        self._stages[key] = st
        return st

    @torch.no_grad()
    def _run_stage(self, st, *, noise_fn: Optional[Callable], stage: int, seed: int, use_graph: bool = True, use_tqdm: bool = False,
                      max_steps: Optional[int] = None, trace: Optional[list] = None, init_images=None, skip_steps=None):
        """el.py:393-545 for one stage (init_images / skip_steps are rejected by sample())."""
        assert init_images is None and not skip_steps
        eng, T, x = st['eng'], st['T'], st['x']
        stream = torch.cuda.current_stream()
        B = eng.src_batch
        n = x[0].numel()

        def draw(tag, like):   # videos are drawn in the reference's (b, c, f, h, w) layout, the state is frame-major
            if not st.get('video', False):
                return noise_fn(tag, tuple(like.shape))
            b, f, c, h, w = like.shape
            return noise_fn(tag, (b, c, f, h, w)).permute(0, 2, 1, 3, 4)

        def init_state():
            if noise_fn is not None:
                x.copy_(draw(("init", stage), x))
            else:
                pl = Plan("edm-init-noise")
                ops.randn(pl, x, seed=seed, stream_id=stage, tag=TAG_INIT, sample_offset=st.get('sample_offset', 0))
                pl.run()
            pl = Plan("edm-init-scale")                      # images = init_sigma * randn (el.py:440-442)
            ops.lincomb(pl, x, x, st['w_init'], st['zero_ptr'], B=B, n_per_sample=n)
            pl.run()
            st['step_ptr'].zero_()

        st['seed_dev'].copy_(torch.tensor([seed & 0x7FFFFFFF, (seed >> 31) & 0x7FFFFFFF], dtype=torch.int32))
        steps = T if max_steps is None else min(T, max_steps)
        if use_graph and st['graph'] is None:
            if noise_fn is not None:
                st['noise'].zero_()
            init_state()
            st['plan'].run()                                 # warm-up outside capture (kernel attributes), then rewind
            st['last'].run()
            torch.cuda.synchronize()
            st['graph'] = ops.Graph(st['plan'], stream)
            st['graph_last'] = ops.Graph(st['last'], stream)
        init_state()
        for i in range(steps):
            if noise_fn is not None:
                st['noise'].copy_(draw(("step", stage, i), st['noise']))
            is_last = i == T - 1
            if use_graph:
                (st['graph_last'] if is_last else st['graph']).launch()
            else:
                (st['last'] if is_last else st['plan']).run()
            if trace is not None:
                trace.append(x.clone())
        if steps == T:
            return st['final']
        return (x.clamp(-1., 1.) + 1) * 0.5                   # truncated loop (tests)

    # ---- public sampling API (el.py:547-745) ---------------------------------------------------------------------------------
    @torch.no_grad()
    def sample(
        self,
        texts: Optional[List[str]] = None,
        text_masks=None,
        text_embeds=None,
        cond_images=None,
        cond_video_frames=None,
        post_cond_video_frames=None,
        inpaint_videos=None,
        inpaint_images=None,
        inpaint_masks=None,
        inpaint_resample_times=5,
        init_images=None,
        skip_steps=None,
        sigma_min=None,
        sigma_max=None,
        video_frames=None,
        batch_size=1,
        cond_scale=1.,
        lowres_sample_noise_level=None,
        start_at_unet_number=1,
        start_image_or_video=None,
        stop_at_unet_number=None,
        return_all_unet_outputs=False,
        return_pil_images=False,
        use_tqdm=True,
        use_one_unet_in_gpu=True,
        device=None,
        *,
        noise_fn: Optional[Callable] = None,   # extensions, as on Imagen.sample
        seed: Optional[int] = None,
        sample_offset: int = 0,
        use_graph: bool = True,
        max_steps: Optional[int] = None,
        conditioning=None,
    ):
        if sigma_min is not None or sigma_max is not None:
            _out_of_scope("sample(sigma_min=/sigma_max=) per-call overrides (set them on the constructor)")
        if inpaint_images is not None or inpaint_videos is not None or inpaint_masks is not None or skip_steps is not None or any(
                i is not None for i in _cast_tuple(init_images)):
            _out_of_scope("ElucidatedImagen.sample(inpaint_images= / init_images= / skip_steps=) (el.py:446-452, 497-533)")
        with self._eval_mode():
            try:
                self._tls.conditioning = conditioning
                if conditioning is not None:
                    assert texts is None and text_embeds is None and text_masks is None, 'pass either `conditioning` or texts / text_embeds'
                    text_embeds, text_masks = conditioning.text_embeds, conditioning.text_masks
                    if text_embeds is None:
                        batch_size = conditioning.batch_size
                return self._sample(texts, text_masks, text_embeds, video_frames, cond_images, cond_video_frames, post_cond_video_frames,
                                    inpaint_videos, inpaint_images, inpaint_masks, inpaint_resample_times, init_images, skip_steps, batch_size,
                                    cond_scale, lowres_sample_noise_level, start_at_unet_number, start_image_or_video, stop_at_unet_number,
                                    return_all_unet_outputs, return_pil_images, device, use_tqdm, noise_fn, seed, sample_offset, use_graph,
                                    max_steps)
            finally:
                self._tls.conditioning = None
