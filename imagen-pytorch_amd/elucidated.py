"""Drop-in `ElucidatedImagen` (Karras et al. EDM sampler over the same cascaded unets): constructor and `.sample()` signature of the
reference (imagen_pytorch/elucidated_imagen.py:76-745 = "el.py"), sampling half only.  SURVEY.md §8(f) NEXT-1 / BASELINE config C4.

A sampling step is [x_hat = x + churn noise, c_in * x_hat] -> denoiser plan (both CFG branches as one 2B-row batch) -> [CFG
combine + c_skip / c_out preconditioning -> exact 0.95-quantile -> Euler step, c_in * x_next] -> denoiser plan again ->
[preconditioning -> quantile -> Heun combination], all HIP kernels (LINCOMB / CFG_X0 / QUANTILE ops), captured ONCE into a
hipGraph and replayed for every step but the last (which has no second-order correction and its own graph).  The per-evaluation
scalars (c_skip, c_out, c_in, c_noise, step weights) live in device tables indexed by a device counter, so replay needs no host
patching.  Preconditioning is expressed through the existing CFG_X0 "noise" form: c_skip*x + c_out*F = (x - sigma'*F) / alpha'
with alpha' = 1/c_skip, sigma' = -c_out/c_skip.

The one_unet_sample options ride on the same tables: `skip_steps` moves the counter's start row, `init_images` is one add on the
initial noise, per-call `sigma_min` / `sigma_max` select another table set, and inpainting (RePaint resampling, el.py:459-470, 486-545)
repeats every timestep's rows `inpaint_resample_times` times with two extra LINCOMB launches in the same captured sequence — the known
pixels blended into x before the churn noise (x_hat = where(mask, known, x) + noise, el.py:497-498) and x += (sigma - sigma_next) * z
after the Heun combination, with identity weights on the rows where the reference skips it.  Video stages (Unet3D, prompt frames)
as in `Imagen`.  Training (`forward`) and self-conditioning unets raise.
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

    @staticmethod
    def _row(i: int, r: int, N: int, R: int) -> int:
        """First table row of inner iteration (timestep i, resample r = R-1..0): two rows per iteration (the evaluations at sigma_hat
        and sigma_next) up to the last timestep, whose iterations have no second evaluation and take one row each."""
        k = R - 1 - r
        return 2 * (i * R + k) if i < N - 1 else 2 * (N - 1) * R + k

    def _tables(self, hp: Hparams, R: int = 1):
        """Per-evaluation device tables, fp32 [rows, 8] each: `coef` (CFG_X0 / time embedding: 1/c_skip, -c_out/c_skip, ..., col 6 =
        c_noise), `w_hat` (x_hat op), `w_euler`, `w_heun`, `w_renoise` (LINCOMB weights w0..w5).  Row layout: `_row` — every timestep
        is repeated R times (inpainting with resampling, el.py:486-535; R = 1 otherwise: row 2i / 2i+1 = first / second evaluation of
        step i).  `w_renoise` is read right after the Euler op's advance (second row of an iteration): x += (sigma - sigma_next) * z
        unless r == 0, the identity there."""
        sigmas = self.sample_schedule(hp.num_sample_steps, hp.rho, hp.sigma_min, hp.sigma_max)
        gammas = torch.where((sigmas >= hp.S_tmin) & (sigmas <= hp.S_tmax), min(hp.S_churn / hp.num_sample_steps, math.sqrt(2) - 1), 0.)
        N = hp.num_sample_steps
        sd = hp.sigma_data
        rows = 2 * (N - 1) * R + 2 * R
        coef = torch.zeros(rows, 8, dtype=torch.float64)
        w_hat, w_euler, w_heun, w_renoise = (torch.zeros(rows, 8, dtype=torch.float64) for _ in range(4))
        w_renoise[:, 0] = 1.0

        def precond(row, sigma):   # el.py:323-336
            c_skip = sd ** 2 / (sigma ** 2 + sd ** 2)
            c_out = sigma * sd * (sd ** 2 + sigma ** 2) ** -0.5
            coef[row, 0], coef[row, 1], coef[row, 6] = 1.0 / c_skip, -c_out / c_skip, math.log(max(sigma, 1e-20)) * 0.25

        for i in range(N):
            sigma, sigma_next, gamma = sigmas[i].item(), sigmas[i + 1].item(), gammas[i].item()
            sigma_hat = sigma + gamma * sigma
            for rs in range(R):
                e = self._row(i, rs, N, R)
                precond(e, sigma_hat)
                w_hat[e, 0] = 1.0
                w_hat[e, 4] = math.sqrt(sigma_hat ** 2 - sigma ** 2) * hp.S_noise          # el.py:489-492
                w_hat[e, 5] = (sigma_hat ** 2 + sd ** 2) ** -0.5                              # c_in(sigma_hat)
                ratio = sigma_next / sigma_hat
                w_euler[e, 0], w_euler[e, 1] = ratio, 1.0 - ratio                             # x_hat + (s_n - s_h)(x_hat - x0)/s_h
                if sigma_next != 0:
                    precond(e + 1, sigma_next)
                    w_euler[e, 5] = (sigma_next ** 2 + sd ** 2) ** -0.5                       # c_in(sigma_next)
                    d = 0.5 * (sigma_next - sigma_hat)
                    w_heun[e + 1, 0], w_heun[e + 1, 1] = 1.0 + d / sigma_hat, -d / sigma_hat
                    w_heun[e + 1, 2], w_heun[e + 1, 3] = d / sigma_next, -d / sigma_next      # el.py:528-529
                    if rs > 0:                                                                # el.py:532-535 (never at the last timestep)
                        w_renoise[e + 1, 4] = sigma - sigma_next
        return sigmas[0].item(), [t.float().contiguous() for t in (coef, w_hat, w_euler, w_heun, w_renoise)]

    # ---- per-stage plans --------------------------------------------------------------------------------------------------
    def _build_stage(self, idx: int, B: int, device, *, cond_scale: float, with_text: bool, inject_noise: bool, sample_offset: int,
                     resample_times: int = 0, frames: int = 0, prompt_frames: tuple = (0, 0)):
        unet = self.unets[idx]
        if getattr(unet, 'self_cond', False):
            from .imagen import _out_of_scope
            _out_of_scope("ElucidatedImagen sampling with self-conditioning unets (el.py:496, 518)")
        S = self.image_sizes[idx]
        hp = self.hparams[idx]
        over = getattr(self._tls, 'sigma_overrides', None)          # sample(sigma_min=, sigma_max=), el.py:425-426, 647-648
        if over is not None:
            hp = hp._replace(**{k: v[idx] for k, v in over.items() if v[idx] is not None})
        cfg = cond_scale != 1.
        R = resample_times
        key = ("edm", idx, B, S, str(device), float(cond_scale), with_text, inject_noise, sample_offset, self.dynamic_thresholding[idx],
               self.dynamic_thresholding_percentile, tuple(hp), frames, prompt_frames, R, self._lane)
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
        init_sigma, (coef, w_hat, w_euler, w_heun, w_renoise) = self._tables(hp, max(R, 1))
        coef, w_hat, w_euler, w_heun, w_renoise = (t.to(dev) for t in (coef, w_hat, w_euler, w_heun, w_renoise))
        if inject_noise:   # the churn noise comes in through t1 (weight col 1) instead of the in-kernel Philox stream (col 4)
            for w in (w_hat, w_renoise):
                w[:, 1] = w[:, 4]
                w[:, 4] = 0
        step_ptr = torch.zeros(1, dtype=torch.int32, device=dev)
        seed_dev = torch.zeros(2, dtype=torch.int32, device=dev)
        eng.bind_step_counter(coef, step_ptr)
        # image stages without inpainting resampling: the timestep-only conditioning chain of both denoiser evaluations of a step comes out
        # of the per-request table (engine.enable_time_table; the table has this sampler's 2T evaluation rows)
        from . import imagen as _imagen
        step_plan = eng.step_plan
        if _imagen.TIME_TABLE and not R and hasattr(eng, "enable_time_table"):
            step_plan = eng.enable_time_table(coef, step_ptr) or step_plan
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

        w_one = torch.zeros(1, 8, device=dev)
        w_one[0, 0] = 1.0
        zero_ptr = torch.zeros(1, dtype=torch.int32, device=dev)
        extra = {}
        if R:   # inpainting (el.py:459-470, 497-498): x_hat = where(mask, known, x) + added noise — the known pixels go into x first
            extra = dict(known=torch.zeros_like(eng.x_in), mask=torch.zeros_like(eng.x_in),
                         noise_renoise=mk() if inject_noise else None)

        def first_eval(plan):
            if R:
                ops.lincomb(plan, extra['known'], x, w_one, zero_ptr, mask=extra['mask'], mask_else=x, label="edm.inpaint.blend", **kw)
            ops.lincomb(plan, x, xhat, w_hat, step_ptr, t1=noise, out2=eng.x_in, label="edm.x_hat", **kw)
            plan.extend(step_plan)
            ops.cfg_x0(plan, xhat, eng.out, coef, step_ptr, x0a, absx0, B=B, n_per_sample=n, cfg=cfg, cond_scale=float(cond_scale),
                       objective="noise", label="edm.precond")
            if dyn:
                ops.quantile(plan, absx0, qa, scr_a, B=B, n=n, q=q)

        full = Plan(f"edm-stage{idx}-step")
        first_eval(full)
        ops.lincomb(full, xhat, xnext, w_euler, step_ptr, t1=x0a, q1=qa if dyn else None, out2=eng.x_in, thr_mode=thr, advance=True,
                    label="edm.euler", **kw)
        full.extend(step_plan)
        ops.cfg_x0(full, xnext, eng.out, coef, step_ptr, x0b, absx0, B=B, n_per_sample=n, cfg=cfg, cond_scale=float(cond_scale),
                   objective="noise", label="edm.precond2")
        if dyn:
            ops.quantile(full, absx0, qb, scr_b, B=B, n=n, q=q)
        ops.lincomb(full, xhat, x, w_heun, step_ptr, t1=x0a, t2=xnext, t3=x0b, q1=qa if dyn else None, q3=qb if dyn else None,
                    thr_mode=thr, advance=not R, label="edm.heun", **kw)
        if R:   # RePaint re-noising before the next resample of the same timestep (identity weights where the reference skips it)
            ops.lincomb(full, x, x, w_renoise, step_ptr, t1=extra['noise_renoise'], advance=True, label="edm.inpaint.renoise",
                        **{**kw, "stream_id": idx | 0x200})

        last = Plan(f"edm-stage{idx}-last")       # sigma_next = 0: Euler step only, then clamp + unnormalise (el.py:515, 540-545)
        first_eval(last)
        ops.lincomb(last, xhat, x, w_euler, step_ptr, t1=x0a, q1=qa if dyn else None, thr_mode=thr, final=True, final_out=final,
                    advance=True, label="edm.euler.final", **kw)

        w_init = torch.zeros(1, 8, device=dev)
        w_init[0, 0] = init_sigma
        st = dict(eng=eng, plan=full, last=last, graph=None, graph_last=None, coef=coef, step_ptr=step_ptr, seed_dev=seed_dev, noise=noise,
                  final=final, T=hp.num_sample_steps, S=S, x=x, w_init=w_init, zero_ptr=zero_ptr,
                  tables=(w_hat, w_euler, w_heun, w_renoise, w_one), video=video, frames=frames, R=R,
                  bufs=(xhat, xnext, x0a, x0b, absx0, qa, qb, scr_a, scr_b), **extra)
        self._stages[key] = st
        return st

    @torch.no_grad()
    def _run_stage(self, st, *, noise_fn: Optional[Callable], stage: int, seed: int, use_graph: bool = True, use_tqdm: bool = False,
                      max_steps: Optional[int] = None, trace: Optional[list] = None, init_images=None, skip_steps=None):
        """el.py:393-545 for one stage: x = sigma_0 * randn (+ init_images), the Karras steps from `skip_steps` on (each run
        `inpaint_resample_times` times when st is an inpainting stage), clamp + unnormalise (the last step's kernel) and the final
        paste of the known pixels."""
        eng, T, x, R = st['eng'], st['T'], st['x'], st['R']
        skip = skip_steps or 0
        assert 0 <= skip < T, 'skip_steps must leave at least one step'
        inner = max(R, 1)
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
            pl = Plan("edm-init-scale")                      # images = init_sigma * randn (el.py:440-442; sigma_0 also when steps are skipped)
            ops.lincomb(pl, x, x, st['w_init'], st['zero_ptr'], B=B, n_per_sample=n)
            pl.run()
            if init_images is not None:
                x.add_(init_images)                          # el.py:446-447
            st['step_ptr'].fill_(self._row(skip, inner - 1, T, inner))   # el.py:477-479: the skipped steps are never run

        st['seed_dev'].copy_(torch.tensor([seed & 0x7FFFFFFF, (seed >> 31) & 0x7FFFFFFF], dtype=torch.int32))
        steps = T - skip if max_steps is None else min(T - skip, max_steps)
        if use_graph and st['graph'] is None:
            if noise_fn is not None:
                st['noise'].zero_()
                if R:
                    st['noise_renoise'].zero_()
            init_state()
            # warm-up outside capture (kernel attributes), then rewind.  Both plans run from table row 0: `plan` advances the device
            # step counter by its two evaluations, and started from the last timestep's rows (skip_steps = T - 1) `last` would read past
            # the 2T-row tables (the kernels do not clamp *step_ptr) and `plan` would feed the all-zero final row into CFG_X0
            st['step_ptr'].zero_()
            st['plan'].run()
            st['step_ptr'].zero_()
            st['last'].run()
            torch.cuda.synchronize()
            st['graph'] = ops.Graph(st['plan'], stream)
            st['graph_last'] = ops.Graph(st['last'], stream)
        init_state()
        for i in range(skip, skip + steps):
            is_last = i == T - 1
            for r in reversed(range(inner)):
                if noise_fn is not None:
                    st['noise'].copy_(draw(("step", stage, i, r) if R else ("step", stage, i), st['noise']))
                    if R and r > 0 and not is_last:           # the reference draws no re-noising sample otherwise (el.py:532)
                        st['noise_renoise'].copy_(draw(("renoise", stage, i, r), st['noise']))
                if use_graph:
                    (st['graph_last'] if is_last else st['graph']).launch()
                else:
                    (st['last'] if is_last else st['plan']).run()
                if trace is not None:
                    trace.append(x.clone())
        if skip + steps == T:
            out = st['final']
        else:
            out = (x.clamp(-1., 1.) + 1) * 0.5                # truncated loop (tests)
        if R:
            out = torch.where(st['mask'] != 0, (st['known'] + 1) * 0.5, out)   # el.py:542-545
        return out

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
        with self._eval_mode():
            try:
                self._tls.conditioning = conditioning
                n_unets = len(self.unets)
                self._tls.sigma_overrides = None if sigma_min is None and sigma_max is None else dict(
                    sigma_min=_cast_tuple(sigma_min, n_unets), sigma_max=_cast_tuple(sigma_max, n_unets))
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
                self._tls.sigma_overrides = None
