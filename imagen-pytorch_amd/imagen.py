"""Drop-in `Imagen` (cascaded DDPM): constructor and `.sample()` signature of the reference
(imagen_pytorch/imagen_pytorch.py:1787-2498 = "ip.py"), sampling half only.

Each cascade stage runs T timesteps; one timestep = [denoiser plan of the stage's UnetEngine (both CFG branches
as one 2B-row batch)] + [CFG combine / x0 / exact 0.95-quantile / dynamic threshold / posterior / noise kernels],
captured ONCE into a hipGraph and replayed T times — the step index lives in a device counter that every
sampler kernel (and the time embedding) reads, so replay needs no host-side parameter patching.

Implemented options: text embeddings or the T5 hook (`texts=`), classifier-free guidance, dynamic thresholding, init_images /
skip_steps, inpainting (images and videos), cond_images, self-conditioning unets, start/stop_at_unet_number, video cascades (Unet3D
stages) with cond_video_frames / post_cond_video_frames.  Training (`forward`, p_losses) raises.  Extensions beyond
the reference signature: `noise_fn`,
`seed`, `sample_offset` (batch sharding), `conditioning` handles, lanes (`with imagen.lane(i)`) and `sample_pipelined`.
"""
from __future__ import annotations

import contextlib
import functools
import os
import queue
import threading
from typing import Callable, Dict, List, Optional

import torch
import torch.nn.functional as F
from torch import nn

from . import ops
from .ops import Plan
from .schedules import GaussianDiffusionContinuousTimes
from .t5 import t5_encode_text
from .unet import NullUnet, Unet
from .unet3d import Unet3D

T5_DIMS = {  # d_model of the encoders the reference accepts by name (t5.py:47-58 reads it from the HF config)
    't5-small': 512, 't5-base': 768, 't5-large': 1024, 't5-3b': 1024, 't5-11b': 1024,
    'google/t5-v1_1-small': 512, 'google/t5-v1_1-base': 768, 'google/t5-v1_1-large': 1024,
    'google/t5-v1_1-xl': 2048, 'google/t5-v1_1-xxl': 4096,
}
DEFAULT_T5_NAME = 'google/t5-v1_1-base'

TIME_TABLE = int(os.environ.get("IMAGEN_TIME_TABLE", "1"))   # A/B switch: the timestep-only conditioning of the image stages from a per-request table (engine.enable_time_table)
_SAMPLING_DEVICE_TYPES = ('cuda',)   # where plans can be launched; tests/test_sample_cpu_replay.py widens it after replacing the launcher
TAG_INIT, TAG_LOWRES = 0x7FFF0001, 0x7FFF0002  # RANDN counter tags (step noise uses the step index)


def _cast_tuple(val, length=None):
    if isinstance(val, list):
        val = tuple(val)
    out = val if isinstance(val, tuple) else ((val,) * (length or 1))
    if length is not None:
        assert len(out) == length
    return out


def _pad_tuple(t, length, fill):
    return t if len(t) >= length else (*t, *((fill,) * (length - len(t))))


def _to_pil_images(batch: torch.Tensor) -> list:
    """torchvision's ToPILImage for float (c, h, w) tensors in [0, 1] (`pic.mul(255).byte()`, HWC; ip.py:2496), without torchvision."""
    from PIL import Image

    modes = {1: 'L', 3: 'RGB', 4: 'RGBA'}
    assert batch.ndim == 4 and batch.shape[1] in modes, 'PIL conversion needs (b, 1 | 3 | 4, h, w) images'
    arr = batch.detach().mul(255).byte().permute(0, 2, 3, 1).cpu().numpy()
    return [Image.fromarray(a[..., 0] if a.shape[-1] == 1 else a, mode=modes[a.shape[-1]]) for a in arr]


def _out_of_scope(what):
    raise NotImplementedError(f"{what} is outside the MI355X sampling hot path of this build (SURVEY.md §2 / §8)")


class Conditioning:
    """Handle on the timestep-invariant conditioning of a batch of prompts (SURVEY.md §8(f) NEXT-3).

    Everything the denoiser derives from the text alone — projected tokens, Perceiver-pooled latents, the non-attention text
    hidden, and the cross-attention K/V of every site (ip.py:1595-1660, 793-808) — is computed by the engines' static plan,
    once per `sample()` call.  Passing the SAME handle to several `sample(conditioning=...)` calls (more seeds for the same
    prompts, `start_at_unet_number` re-runs, ...) lets each stage keep what its static plan wrote the first time: the engine
    buffers are stamped with the handle's token and the static plan is skipped while the stamp matches."""

    def __init__(self, text_embeds: Optional[torch.Tensor], text_masks: Optional[torch.Tensor], batch_size: int):
        self.text_embeds, self.text_masks, self.batch_size = text_embeds, text_masks, batch_size
        self.token = object()



def _seed_spans(seed, batch: int, sample_offset: int = 0):
    """[(Philox seed, first row, rows, global index of the first row)]: an int seed is one span over the whole batch; a list of (seed, rows)
    pairs — merged requests (Imagen.sample_requests) — one span per request, each with sample indices restarting at 0."""
    if isinstance(seed, (list, tuple)):
        spans, r0 = [], 0
        for sd, cnt in seed:
            spans.append((int(sd), r0, int(cnt), 0))
            r0 += int(cnt)
        assert r0 == batch, f'merged requests cover {r0} rows, the batch has {batch}'
        return spans
    return [(int(seed), 0, batch, sample_offset)]

class Imagen(nn.Module):
    def __init__(
        self,
        unets,
        *,
        image_sizes,
        text_encoder_name=DEFAULT_T5_NAME,
        text_embed_dim=None,
        channels=3,
        timesteps=1000,
        cond_drop_prob=0.1,
        loss_type='l2',
        noise_schedules='cosine',
        pred_objectives='noise',
        random_crop_sizes=None,
        lowres_noise_schedule='linear',
        lowres_sample_noise_level=0.2,
        per_sample_random_aug_noise_level=False,
        condition_on_text=True,
        auto_normalize_img=True,
        dynamic_thresholding=True,
        dynamic_thresholding_percentile=0.95,
        only_train_unet_number=None,
        temporal_downsample_factor=1,
        resize_cond_video_frames=True,
        resize_mode='nearest',
        min_snr_loss_weight=True,
        min_snr_gamma=5,
    ):
        super().__init__()
        if loss_type not in ('l1', 'l2', 'huber'):
            raise NotImplementedError()
        self.loss_type = loss_type
        self.condition_on_text = condition_on_text
        self.unconditional = not condition_on_text
        self.channels = channels

        unets = _cast_tuple(unets)
        num_unets = len(unets)
        timesteps = _cast_tuple(timesteps, num_unets)

        # 'cosine', 'cosine', then 'linear' for further super-resolution stages (ip.py:1853-1855)
        noise_schedules = _pad_tuple(_pad_tuple(_cast_tuple(noise_schedules), 2, 'cosine'), num_unets, 'linear')
        self.noise_schedulers = nn.ModuleList([GaussianDiffusionContinuousTimes(noise_schedule=s, timesteps=t)
                                               for t, s in zip(timesteps, noise_schedules)])
        self.random_crop_sizes = _cast_tuple(random_crop_sizes, num_unets)
        self.lowres_noise_schedule = GaussianDiffusionContinuousTimes(noise_schedule=lowres_noise_schedule)
        self.pred_objectives = _cast_tuple(pred_objectives, num_unets)

        self.text_encoder_name = text_encoder_name
        # hook of sample(texts=...), ip.py:1832 / 2326-2332; replace it to plug in another encoder
        self.encode_text = functools.partial(t5_encode_text, name=text_encoder_name)
        if text_embed_dim is None:
            if text_encoder_name not in T5_DIMS:
                raise ValueError(f"unknown text encoder '{text_encoder_name}': pass text_embed_dim explicitly")
            text_embed_dim = T5_DIMS[text_encoder_name]
        self.text_embed_dim = text_embed_dim

        self.unets = nn.ModuleList([])
        self.unet_being_trained_index = -1
        self.only_train_unet_number = only_train_unet_number
        for ind, one_unet in enumerate(unets):
            assert isinstance(one_unet, (Unet, Unet3D, NullUnet))
            one_unet = one_unet.cast_model_parameters(
                lowres_cond=ind > 0, cond_on_text=self.condition_on_text,
                text_embed_dim=self.text_embed_dim if self.condition_on_text else None,
                channels=self.channels, channels_out=self.channels)   # ip.py:1897-1903 (may re-instantiate with fresh weights)
            self.unets.append(one_unet)

        image_sizes = _cast_tuple(image_sizes)
        self.image_sizes = image_sizes
        assert num_unets == len(image_sizes), f'you did not supply the correct number of u-nets ({len(unets)}) for resolutions {image_sizes}'
        self.sample_channels = _cast_tuple(self.channels, num_unets)
        self.is_video = any(isinstance(u, Unet3D) for u in self.unets)     # ip.py:1918-1919
        self.resize_mode = resize_mode                    # ip.py:1924: every image resize of the cascade goes through this mode
        self.temporal_downsample_factor = _cast_tuple(temporal_downsample_factor, num_unets)
        self.resize_cond_video_frames = resize_cond_video_frames                                                           # ip.py:1931
        assert self.temporal_downsample_factor[-1] == 1, 'downsample factor of last stage must be 1'                       # ip.py:1934
        assert tuple(sorted(self.temporal_downsample_factor, reverse=True)) == self.temporal_downsample_factor, \
            'temporal downsample factor must be in order of descending'                                                   # ip.py:1935

        lowres_conditions = tuple(u.lowres_cond for u in self.unets)
        assert lowres_conditions == (False, *((True,) * (num_unets - 1))), \
            'the first unet must be unconditioned (by low resolution image), and the rest of the unets must have `lowres_cond` set to True'

        self.lowres_sample_noise_level = lowres_sample_noise_level
        self.per_sample_random_aug_noise_level = per_sample_random_aug_noise_level
        self.cond_drop_prob = cond_drop_prob
        self.can_classifier_guidance = cond_drop_prob > 0.
        self.auto_normalize_img = auto_normalize_img
        self.input_image_range = (0. if auto_normalize_img else -1., 1.)
        self.normalize_img = (lambda img: img * 2 - 1) if auto_normalize_img else (lambda img: img)          # ip.py:1885-1888
        self.unnormalize_img = (lambda img: (img + 1) * 0.5) if auto_normalize_img else (lambda img: img)
        self.dynamic_thresholding = _cast_tuple(dynamic_thresholding, num_unets)
        self.dynamic_thresholding_percentile = dynamic_thresholding_percentile
        min_snr_loss_weight = _cast_tuple(min_snr_loss_weight, num_unets)
        min_snr_gamma = _cast_tuple(min_snr_gamma, num_unets)
        self.min_snr_gamma = tuple((g if use else None) for use, g in zip(min_snr_loss_weight, min_snr_gamma))

        self.register_buffer('_temp', torch.tensor([0.]), persistent=False)
        self.to(next(self.unets.parameters()).device)
        self._stages = {}
        self._streams: Dict[tuple, torch.cuda.Stream] = {}   # (lane, device) -> private stream
        self._tls = threading.local()                        # per-thread: lane index, conditioning handle
        self._mode_lock = threading.Lock()
        self._build_lock = threading.RLock()                 # stage construction (engine plans + shared weight packing) is serialised
        self._mode_depth = 0
        self._call_counter = 0

    def _next_seed(self) -> int:
        """Derived Philox key of a call without `seed=`: distinct per call, also when lanes call concurrently (the counter is locked)."""
        with self._mode_lock:
            self._call_counter += 1
            c = self._call_counter
        return (int(torch.initial_seed()) * 1000003 + c) & ((1 << 62) - 1)

    # ---- device bookkeeping (API parity; weights stay resident as packed copies, nothing is shuffled over PCIe) ----
    @property
    def device(self):
        return self._temp.device

    def force_unconditional_(self):
        self.condition_on_text = False
        self.unconditional = True
        for unet in self.unets:
            unet.cond_on_text = False

    def get_unet(self, unet_number):
        assert 0 < unet_number <= len(self.unets)
        return self.unets[unet_number - 1]

    def reset_unets_all_one_device(self, device=None):
        device = device if device is not None else self.device
        self.unets.to(device)
        self.unet_being_trained_index = -1

    def state_dict(self, *args, **kwargs):
        return super().state_dict(*args, **kwargs)

    def load_state_dict(self, *args, **kwargs):
        return super().load_state_dict(*args, **kwargs)

    def forward(self, *args, **kwargs):
        _out_of_scope("Imagen.forward (training loss, ip.py:2500-2734)")

    # ---- lanes: independent sampling contexts (own stream, own stage buffers + graphs, shared packed weights) ------------------
    @property
    def _lane(self) -> int:
        return getattr(self._tls, 'lane', 0)

    @contextlib.contextmanager
    def lane(self, index: int):
        """Extension: `with imagen.lane(i): imagen.sample(...)` runs the call on lane i — its own HIP stream and its own
        per-stage buffers / captured graphs; the packed weights are shared.  Calls on different lanes may be issued from
        different threads and overlap on the GPU (results are those of the same calls made one after the other)."""
        prev = self._lane
        self._tls.lane = int(index)
        try:
            yield self
        finally:
            self._tls.lane = prev

    @contextlib.contextmanager
    def _eval_mode(self):
        """eval() for the duration of a sample() call, restored when the LAST concurrent call ends."""
        with self._mode_lock:
            if self._mode_depth == 0:
                self._was_training = self.training
                self.eval()
            self._mode_depth += 1
        try:
            yield
        finally:
            with self._mode_lock:
                self._mode_depth -= 1
                if self._mode_depth == 0:
                    self.train(self._was_training)

    @torch.no_grad()
    def sample_requests(self, requests: List[dict], **common) -> List[torch.Tensor]:
        """Extension (serving): several independent `sample()` requests MERGED into one batch — one set of kernel launches per denoiser step for
        all of them (48 images as two merged batches of 24 sample 16 % faster on MI355X than as six concurrent requests of 8, profiles/r06_q_*).
        `requests` is a list of per-request keyword dicts (`text_embeds=` | `texts=`, `text_masks=`, `seed=`, `batch_size=` for unconditional
        cascades), `common` the keywords shared by all (`cond_scale=`, `max_steps=`, ...).  Every row draws the noise of ITS OWN request — that
        request's Philox key and sample indices 0 .. b-1 (ABI 11: ImagenDdpmUpdateParams.row_keys) — so a request's images are the ones
        `sample(**common, **request)` produces up to the fp16 rounding of a different batch's tile configuration.  Returns one tensor per request.
        Not covered (raise): inpainting, init images, conditioning images / frames, `noise_fn`, `start_image_or_video`."""
        for k in ('inpaint_images', 'inpaint_videos', 'inpaint_masks', 'init_images', 'cond_images', 'cond_video_frames', 'post_cond_video_frames',
                  'noise_fn', 'start_image_or_video', 'conditioning', 'return_pil_images', 'return_all_unet_outputs', 'sample_offset'):
            if common.get(k) is not None or any(r.get(k) is not None for r in requests):
                _out_of_scope(f"sample_requests(..., {k}=...)")
        if type(self)._run_stage is not Imagen._run_stage:
            _out_of_scope(f"{type(self).__name__}.sample_requests (its sampler's noise launches take one key per batch)")
        if not requests:
            return []
        embeds, masks, seeds, sizes = [], [], [], []
        for r in requests:
            extra = set(r) - {'texts', 'text_embeds', 'text_masks', 'seed', 'batch_size'}
            assert not extra, f'per-request keywords {sorted(extra)} must be the same for all merged requests: pass them as common keywords'
            te, tm = self._resolve_text(r.get('texts'), r.get('text_embeds'), r.get('text_masks'), self.device)
            b = te.shape[0] if te is not None else int(r.get('batch_size', 1))
            embeds.append(te)
            masks.append(tm)
            sizes.append(b)
            seeds.append((self._next_seed() if r.get('seed') is None else int(r['seed']), b))
        kw = dict(common)
        kw.setdefault('use_tqdm', False)
        if embeds[0] is not None:
            width = max(e.shape[1] for e in embeds)       # (prompts of different lengths: zero-padded, masked)
            pad = lambda t, fill: torch.cat((t, t.new_full((t.shape[0], width - t.shape[1], *t.shape[2:]), fill)), 1) if t.shape[1] < width else t
            kw['text_embeds'] = torch.cat([pad(e, 0.0) for e in embeds])
            kw['text_masks'] = torch.cat([pad(m, False) for m in masks])
        else:
            kw['batch_size'] = sum(sizes)
        out = self.sample(seed=seeds, **kw)
        return list(torch.split(out, sizes))

    @torch.no_grad()
    def sample_pipelined(self, batches: List[dict], **common) -> List[torch.Tensor]:
        """Extension: sample successive batches with the cascade stages OVERLAPPED — stage s of batch k runs (own thread, own
        lane / stream / graph) while stage s+1 of batch k-1 does.  `batches` is a list of per-batch `sample()` keyword dicts
        (text_embeds=..., seed=..., ...), `common` the keywords shared by all.  Returns the final-stage images per batch, each
        bit-identical to `sample(**common, **batches[k])` (the stage hand-off is the same [0, 1] image the sequential cascade
        passes on, ip.py:2487-2490).  The base stage is latency-bound (few workgroups per launch at 64^2), the super-resolution
        stage throughput-bound: overlapped they fill the chip, so the sustained rate is set by the slower stage alone."""
        n_stages = len(self.unets)
        for k in ('start_at_unet_number', 'stop_at_unet_number', 'start_image_or_video', 'return_all_unet_outputs'):
            assert k not in common and all(k not in b for b in batches), f'sample_pipelined drives `{k}` itself'
        jobs = []
        for b in batches:
            kw = {**common, **b}
            if kw.get('seed') is None:                   # the seed must be the same for every stage of a batch (as in one sample() call)
                kw['seed'] = self._next_seed()
            kw.setdefault('use_tqdm', False)
            jobs.append(kw)
        if n_stages == 1 or len(jobs) == 0:
            return [self.sample(**kw) for kw in jobs]
        caller = torch.cuda.current_stream(self.device)
        ready = torch.cuda.Event()
        ready.record(caller)
        qs = [queue.Queue() for _ in range(n_stages + 1)]
        errors: List[BaseException] = []

        def worker(s: int):
            try:
                with self.lane(0x100 + s), torch.cuda.device(self.device):
                    torch.cuda.current_stream().wait_event(ready)
                    while True:
                        item = qs[s].get()
                        if item is None or errors:
                            break
                        k, img = item
                        kw = dict(jobs[k])
                        if s > 0:
                            kw.update(start_at_unet_number=s + 1, start_image_or_video=img)
                        out = self.sample(**kw, stop_at_unet_number=s + 1)   # returns after this lane's stream has drained
                        qs[s + 1].put((k, out))
            except BaseException as e:   # noqa: BLE001 — re-raised in the caller
                errors.append(e)
            finally:
                qs[s + 1].put(None)

        threads = [threading.Thread(target=worker, args=(s,), daemon=True) for s in range(n_stages)]
        for t in threads:
            t.start()
        for k in range(len(jobs)):
            qs[0].put((k, None))
        qs[0].put(None)
        results: Dict[int, torch.Tensor] = {}
        while True:
            item = qs[n_stages].get()
            if item is None:
                break
            results[item[0]] = item[1]
        for t in threads:
            t.join()
        if errors:
            raise errors[0]
        return [results[k] for k in range(len(jobs))]

    # ---- one cascade stage -------------------------------------------------------------------------------------
    def _stage(self, *args, **kwargs):
        """`_build_stage` under the construction lock; a NEW stage's packing kernels (weights shared with the other lanes) have
        completed before the stage becomes visible."""
        with self._build_lock:
            before = {k: id(v) for k, v in self._stages.items()}
            st = self._build_stage(*args, **kwargs)
            # a stage was constructed (new key, or a stale one rebuilt under its old key after a weight swap): its packing work runs on
            # THIS lane's stream and the packed weights are shared, so it must be complete before another lane can see the stage
            built = any(before.get(k) != id(v) for k, v in self._stages.items())
            if built and torch.cuda.is_available():
                torch.cuda.current_stream().synchronize()
            return st

    def _build_stage(self, idx: int, B: int, device, *, cond_scale: float, with_text: bool, inject_noise: bool, sample_offset: int,
                     resample_times: int = 0, frames: int = 0, prompt_frames: tuple = (0, 0)):
        """Build (or fetch) the per-timestep plan + graph of stage `idx` for batch B.

        resample_times = R > 0 selects the inpainting plan (ip.py:2237-2275): the device counter then counts INNER iterations
        (timestep i, resample r = R-1..0), every coefficient table has one row per inner iteration, and one launch sequence
        [blend known pixels at level t -> denoiser -> posterior step -> re-noise t_next -> t] serves all of them: the re-noising
        weights are the identity on the rows where the reference skips it, so a single captured graph covers the whole loop."""
        unet = self.unets[idx]
        S = self.image_sizes[idx]
        sched = self.noise_schedulers[idx]
        T = sched.num_timesteps
        cfg = cond_scale != 1.
        key = (idx, B, S, str(device), float(cond_scale), with_text, inject_noise, sample_offset, self.dynamic_thresholding[idx],
               self.pred_objectives[idx], self.dynamic_thresholding_percentile, resample_times, frames, prompt_frames, self._lane)
        st = self._stages.get(key)
        if st is not None and not st['eng'].stale():
            return st
        rows = 2 * B if cfg else B
        video = isinstance(unet, Unet3D)
        if video:
            # Imagen-Video stage: the sampler state is the engine's frame-major clip (b, f, c, h, w); every sampler kernel is
            # elementwise per sample (the dynamic threshold is one quantile over the whole clip, ip.py:1921, 2097-2101), so only
            # the sample size changes.  sample() converts from / to the reference's (b, c, f, h, w) at the API boundary.
            assert frames > 0, 'video_frames must be passed in on sample time if training on video'
            from . import engine3d
            eng = engine3d.UnetEngine3D(unet, rows, B, frames, S, device, with_text=with_text, pre_frames=prompt_frames[0],
                                        post_frames=prompt_frames[1])
        else:
            from . import engine
            eng = engine.UnetEngine(unet, rows, B, S, device, with_text=with_text)
        n = eng.x_in[0].numel()
        dev = device
        R = resample_times
        if R:
            coef, blend_coef, renoise_coef = (t.to(dev) for t in sched.inpaint_coefficients(R, philox=not inject_noise))
        else:
            coef = sched.step_coefficients().to(dev)
        step_ptr = torch.zeros(1, dtype=torch.int32, device=dev)
        seed_dev = torch.zeros(2, dtype=torch.int32, device=dev)
        row_keys = torch.zeros(B, 4, dtype=torch.int32, device=dev)   # (Philox key lo, hi, global sample index, 0) of every row: _run_stage fills it per call
        eng.bind_step_counter(coef, step_ptr)
        x0 = torch.empty(B, n, device=dev)
        absx0 = torch.empty(B, n, device=dev)
        quant = torch.empty(B, device=dev)
        scratch = torch.empty(B * ops.ENUMS["IMAGEN_QUANTILE_SCRATCH_WORDS"], dtype=torch.int32, device=dev)
        noise = torch.empty_like(eng.x_in) if inject_noise else None
        final = torch.empty_like(eng.x_in)
        plan = Plan(f"stage{idx}-step")
        extra = {}
        if R:
            known = torch.zeros_like(eng.x_in)           # the known image (video: frame-major clip), normalised, at this stage's size
            mask = torch.zeros_like(eng.x_in)            # 1.0 where the known pixel is kept
            noise_blend = torch.zeros_like(known) if inject_noise else None
            noise_renoise = torch.zeros_like(known) if inject_noise else None
            ops.lincomb(plan, known, eng.x_in, blend_coef, step_ptr, B=B, n_per_sample=n, t1=noise_blend, mask=mask, mask_else=eng.x_in,
                        stream_id=idx | 0x100, sample_offset=sample_offset, seed_ptr=seed_dev, label="inpaint.blend")
            extra = dict(known=known, mask=mask, noise_blend=noise_blend, noise_renoise=noise_renoise)
        # image stages without inpainting resampling: the timestep-only conditioning chain of the denoiser is evaluated for all steps at
        # once per request (engine.enable_time_table), each step copies its rows
        step_plan = eng.step_plan
        if TIME_TABLE and not video and not R:
            step_plan = eng.enable_time_table(coef, step_ptr) or step_plan
        plan.extend(step_plan)
        ops.cfg_x0(plan, eng.x_in, eng.out, coef, step_ptr, x0, absx0, B=B, n_per_sample=n, cfg=cfg, cond_scale=float(cond_scale),
                   objective=self.pred_objectives[idx])
        dyn = bool(self.dynamic_thresholding[idx])
        if dyn:
            ops.quantile(plan, absx0, quant, scratch, B=B, n=n, q=float(self.dynamic_thresholding_percentile))
        x0_thr = None
        if getattr(unet, 'self_cond', False):            # ip.py:2249: each step conditions on the previous step's thresholded x0
            if video:
                _out_of_scope("self-conditioning video unets")
            x0_thr = torch.zeros(B, n, device=dev)
            eng.bind_self_cond(x0_thr)
            extra['x0_thr'] = x0_thr
        ops.ddpm_update(plan, eng.x_in, x0, quant if dyn else None, coef, noise, final, step_ptr, B=B, n_per_sample=n,
                        dynamic_threshold=dyn, total_steps=T * max(R, 1), seed=0, stream_id=idx, sample_offset=sample_offset,
                        seed_ptr=seed_dev, advance=not R, x0_thr=x0_thr, row_keys=None if inject_noise else row_keys)
        if R:
            ops.lincomb(plan, eng.x_in, eng.x_in, renoise_coef, step_ptr, B=B, n_per_sample=n, t1=extra['noise_renoise'], advance=True,
                        stream_id=idx | 0x200, sample_offset=sample_offset, seed_ptr=seed_dev, label="inpaint.renoise")
        st = dict(eng=eng, plan=plan, graph=None, coef=coef, step_ptr=step_ptr, seed_dev=seed_dev, row_keys=row_keys, noise=noise, final=final, T=T, S=S,
                  quant=quant, x0=x0, R=R, video=video, frames=frames, **extra)
        self._stages[key] = st
        return st

    @torch.no_grad()
    def _run_stage(self, st, *, noise_fn: Optional[Callable], stage: int, seed: int, use_graph: bool = True, use_tqdm: bool = False,
                      max_steps: Optional[int] = None, trace: Optional[list] = None, init_images: Optional[torch.Tensor] = None,
                      skip_steps: Optional[int] = None):
        """ip.py:2167-2289 for one stage: x_T ~ N(0, I) (+ init_images), the ancestral steps from `skip_steps` on (each run
        `inpaint_resample_times` times when st is an inpainting plan), clamp + unnormalise (done by the last step's kernel) and
        the final paste of the known pixels."""
        eng, plan, T, R = st['eng'], st['plan'], st['T'], st['R']
        B, S = eng.src_batch, st['S']
        stream = torch.cuda.current_stream()
        skip = skip_steps or 0
        assert 0 <= skip < T, 'skip_steps must leave at least one timestep'
        inner = max(R, 1)
        init = None
        spans = _seed_spans(seed, B, st.get('sample_offset', 0))
        if noise_fn is None:
            init = Plan("init-noise")
            for sd, r0, cnt, i0 in spans:               # (one span per merged request: its own key, its own sample indices)
                ops.randn(init, eng.x_in[r0:r0 + cnt], seed=sd, stream_id=stage, tag=TAG_INIT, sample_offset=i0)

        video = st.get('video', False)

        def draw(tag, like):
            """Injected Gaussian draw for the internal buffer `like`; videos are drawn in the reference's (b, c, f, h, w) layout."""
            if not video:
                return noise_fn(tag, tuple(like.shape))
            b, f, c, h, w = like.shape
            return noise_fn(tag, (b, c, f, h, w)).permute(0, 2, 1, 3, 4)

        def reset_state():
            if noise_fn is not None:
                eng.x_in.copy_(draw(("init", stage), eng.x_in))
            else:
                init.run()
            if init_images is not None:
                eng.x_in.add_(init_images)              # ip.py:2205-2206
            if st.get('x0_thr') is not None:
                st['x0_thr'].zero_()                    # no x0 estimate yet: the first step self-conditions on zeros (ip.py:2210, 1542)
            st['step_ptr'].fill_(skip * inner)           # ip.py:2228-2229: the skipped timesteps are simply never run

        reset_state()
        assert not (R and len(spans) > 1), 'merged requests do not cover inpainting (its re-noising launches take one key per batch)'
        st['seed_dev'].copy_(torch.tensor([spans[0][0] & 0x7FFFFFFF, (spans[0][0] >> 31) & 0x7FFFFFFF], dtype=torch.int32))
        keys = torch.zeros(B, 4, dtype=torch.int64)
        for sd, r0, cnt, i0 in spans:
            keys[r0:r0 + cnt, 0], keys[r0:r0 + cnt, 1] = sd & 0x7FFFFFFF, (sd >> 31) & 0x7FFFFFFF
            keys[r0:r0 + cnt, 2] = torch.arange(i0, i0 + cnt)
        st['row_keys'].copy_(keys.to(torch.int32))
        steps = T - skip if max_steps is None else min(T - skip, max_steps)
        if use_graph and st['graph'] is None:
            plan.run()                                   # warm-up outside capture (sets kernel attributes), then rewind
            reset_state()
            torch.cuda.synchronize()
            st['graph'] = ops.Graph(plan, stream)
        it = range(skip, skip + steps)
        if use_tqdm:
            try:
                from tqdm.auto import tqdm
                it = tqdm(it, desc='sampling loop time step', total=steps)
            except ImportError:
                pass
        for i in it:
            for r in reversed(range(inner)):
                if noise_fn is not None:
                    if R:
                        st['noise_blend'].copy_(draw(("inpaint", stage, i, r), st['noise']))
                        st['noise'].copy_(draw(("step", stage, i, r), st['noise']))
                        if r > 0 and i < T - 1:          # the reference draws no re-noising sample otherwise (ip.py:2268)
                            st['noise_renoise'].copy_(draw(("renoise", stage, i, r), st['noise']))
                    else:
                        st['noise'].copy_(draw(("step", stage, i), st['noise']))
                if use_graph:
                    st['graph'].launch()
                else:
                    plan.run()
                if trace is not None:
                    trace.append(eng.x_in.clone())
        if skip + steps == T:
            out = st['final']
        else:
            out = (eng.x_in.clamp(-1., 1.) + 1) * 0.5    # truncated loop (tests): same epilogue as ip.py:2281-2288
        if R:
            out = torch.where(st['mask'] != 0, (st['known'] + 1) * 0.5, out)   # ip.py:2283-2288
        return out

    # ---- the reference's step-level sampler methods (ip.py:2042-2289) ------------------------------------------------------
    # `sample()` below runs a whole stage as one captured graph per timestep and never calls these.  They are the reference's
    # per-step API, kept for callers that drive single steps themselves (custom loops, guidance experiments): the denoiser
    # evaluation goes through the same kernel plan (Unet.forward_with_cond_scale), the O(B*3*S*S) posterior arithmetic around it is
    # plain device-side tensor code with the reference's signatures, argument meaning and return values.
    def resize_to(self, img, size, **kwargs):
        """resize_image_to with this model's resize_mode (ip.py:152-168, 1924)."""
        assert not kwargs or self.is_video, 'frame arguments are for video stages'
        if img.ndim == 5:                                # resize_video_to (iv.py:134-156): (b, c, f, h, w), optional target_frames
            frames = kwargs.get('target_frames') or img.shape[2]
            if tuple(img.shape[-3:]) == (frames, size, size):
                return img
            return F.interpolate(img, (frames, size, size), mode=self.resize_mode)
        return img if img.shape[-1] == size else F.interpolate(img, size, mode=self.resize_mode)

    def _step_conditioning_checks(self, unet, cond_video_frames, post_cond_video_frames, cond_scale):
        assert not (cond_scale != 1. and not self.can_classifier_guidance), \
            'imagen was not trained with conditional dropout, and thus one cannot use classifier free guidance (cond_scale anything other than 1)'
        # (prompt frames given to an image cascade are ignored, as in the reference: ip.py:2057-2070 builds video_kwargs only if is_video)

    @torch.no_grad()
    def p_mean_variance(self, unet, x, t, *, noise_scheduler, text_embeds=None, text_mask=None, cond_images=None, cond_video_frames=None,
                        post_cond_video_frames=None, lowres_cond_img=None, self_cond=None, lowres_noise_times=None, cond_scale=1.,
                        model_output=None, t_next=None, pred_objective='noise', dynamic_threshold=True):
        """ip.py:2042-2110: denoiser output (or `model_output`) -> x_0 estimate -> threshold -> posterior (mean, variance,
        log variance) of x_{t_next}; returns that triple and the thresholded x_0."""
        self._step_conditioning_checks(unet, cond_video_frames, post_cond_video_frames, cond_scale)
        pred = model_output
        if pred is None:
            video_kwargs = dict(cond_video_frames=cond_video_frames, post_cond_video_frames=post_cond_video_frames) if self.is_video else {}
            pred = unet.forward_with_cond_scale(x, noise_scheduler.get_condition(t), text_embeds=text_embeds, text_mask=text_mask,
                                                cond_images=cond_images, self_cond=self_cond, cond_scale=cond_scale,
                                                lowres_cond_img=lowres_cond_img,
                                                lowres_noise_times=self.lowres_noise_schedule.get_condition(lowres_noise_times),
                                                **video_kwargs)                      # ip.py:2057-2070
        if pred_objective == 'noise':
            x_start = noise_scheduler.predict_start_from_noise(x, t=t, noise=pred)
        elif pred_objective == 'x_start':
            x_start = pred
        elif pred_objective == 'v':
            x_start = noise_scheduler.predict_start_from_v(x, t=t, v=pred)
        else:
            raise ValueError(f'unknown objective {pred_objective}')
        if dynamic_threshold:
            # per-sample percentile of |x_0|, never below 1: clamp to it and rescale into [-1, 1] (Imagen paper, appendix)
            s = torch.quantile(x_start.flatten(1).abs(), self.dynamic_thresholding_percentile, dim=-1).clamp(min=1.)
            s = s.reshape(-1, *((1,) * (x_start.ndim - 1)))
            x_start = x_start.clamp(-s, s) / s
        else:
            x_start = x_start.clamp(-1., 1.)
        return noise_scheduler.q_posterior(x_start=x_start, x_t=x, t=t, t_next=t_next), x_start

    @torch.no_grad()
    def p_sample(self, unet, x, t, *, noise_scheduler, t_next=None, text_embeds=None, text_mask=None, cond_images=None,
                 cond_video_frames=None, post_cond_video_frames=None, cond_scale=1., self_cond=None, lowres_cond_img=None,
                 lowres_noise_times=None, pred_objective='noise', dynamic_threshold=True):
        """ip.py:2112-2165: one ancestral step, x_{t_next} = mean + [t_next != 0] * exp(log_var / 2) * eps; returns it and x_0."""
        (mean, _, log_var), x_start = self.p_mean_variance(
            unet, x=x, t=t, t_next=t_next, noise_scheduler=noise_scheduler, text_embeds=text_embeds, text_mask=text_mask,
            cond_images=cond_images, cond_video_frames=cond_video_frames, post_cond_video_frames=post_cond_video_frames,
            cond_scale=cond_scale, self_cond=self_cond, lowres_cond_img=lowres_cond_img, lowres_noise_times=lowres_noise_times,
            pred_objective=pred_objective, dynamic_threshold=dynamic_threshold)
        noise = torch.randn_like(x)
        last = (t_next == 0) if isinstance(noise_scheduler, GaussianDiffusionContinuousTimes) else (t == 0)
        nonzero = (1 - last.float()).reshape(x.shape[0], *((1,) * (x.ndim - 1)))
        return mean + nonzero * (0.5 * log_var).exp() * noise, x_start

    @torch.no_grad()
    def p_sample_loop(self, unet, shape, *, noise_scheduler, lowres_cond_img=None, lowres_noise_times=None, text_embeds=None,
                      text_mask=None, cond_images=None, cond_video_frames=None, post_cond_video_frames=None, inpaint_images=None,
                      inpaint_videos=None, inpaint_masks=None, inpaint_resample_times=5, init_images=None, skip_steps=None, cond_scale=1,
                      pred_objective='noise', dynamic_threshold=True, use_tqdm=True):
        """ip.py:2167-2289 with the reference's signature: the loop of one stage, step by step through `p_sample` (torch's
        generator supplies the noise, in the reference's draw order).  `sample()` is the fast way to run a stage."""
        device = self.device
        batch = shape[0]
        resize_kwargs = dict(target_frames=shape[-3]) if len(shape) == 5 else {}      # ip.py:2198-2200
        img = torch.randn(shape, device=device)
        if init_images is not None:
            img = img + init_images
        if inpaint_videos is not None:                   # ip.py:2214
            inpaint_images = inpaint_videos
        inpainting = inpaint_images is not None and inpaint_masks is not None
        if inpainting:
            known = self.resize_to(self.normalize_img(inpaint_images), shape[-1], **resize_kwargs)
            keep = self.resize_to(inpaint_masks[:, None].float(), shape[-1], **resize_kwargs).bool()
        steps = noise_scheduler.get_sampling_timesteps(batch, device=device)[(skip_steps or 0):]
        if use_tqdm:
            try:
                from tqdm.auto import tqdm
                steps = tqdm(steps, desc='sampling loop time step', total=len(steps))
            except ImportError:
                pass
        x_start = None
        for times, times_next in steps:
            final_step = bool(torch.all(times_next == 0))
            for r in reversed(range(inpaint_resample_times if inpainting else 1)):
                if inpainting:       # known region at this step's noise level
                    img = torch.where(keep, noise_scheduler.q_sample(known, t=times)[0], img)
                img, x_start = self.p_sample(unet, img, times, t_next=times_next, text_embeds=text_embeds, text_mask=text_mask,
                                       cond_images=cond_images, cond_scale=cond_scale, lowres_cond_img=lowres_cond_img,
                                       self_cond=x_start if getattr(unet, 'self_cond', False) else None,
                                       lowres_noise_times=lowres_noise_times, noise_scheduler=noise_scheduler,
                                       pred_objective=pred_objective, dynamic_threshold=dynamic_threshold,
                                       cond_video_frames=cond_video_frames, post_cond_video_frames=post_cond_video_frames)
                if inpainting and r > 0 and not final_step:      # resample: back up to this step's level (RePaint)
                    img = noise_scheduler.q_sample_from_to(img, times_next, times)
        img = img.clamp(-1., 1.)
        if inpainting:
            img = torch.where(keep, known, img)
        return self.unnormalize_img(img)

    # ---- public sampling API (ip.py:2291-2498) ------------------------------------------------------------------
    @torch.no_grad()
    def sample(
        self,
        texts: Optional[List[str]] = None,
        text_masks=None,
        text_embeds=None,
        video_frames=None,
        cond_images=None,
        cond_video_frames=None,
        post_cond_video_frames=None,
        inpaint_videos=None,
        inpaint_images=None,
        inpaint_masks=None,
        inpaint_resample_times=5,
        init_images=None,
        skip_steps=None,
        batch_size=1,
        cond_scale=1.,
        lowres_sample_noise_level=None,
        start_at_unet_number=1,
        start_image_or_video=None,
        stop_at_unet_number=None,
        return_all_unet_outputs=False,
        return_pil_images=False,
        device=None,
        use_tqdm=True,
        use_one_unet_in_gpu=True,
        *,
        noise_fn: Optional[Callable] = None,   # extension: injectable Gaussian noise (parity tests), tags as in oracle/sampler_oracle.py
        seed: Optional[int] = None,            # extension: Philox seed of this call (sample_requests passes [(seed, rows), ...]: one key per merged request)
        sample_offset: int = 0,                # extension: global index of sample 0 (batch sharding)
        use_graph: bool = True,
        max_steps: Optional[int] = None,
        conditioning: Optional[Conditioning] = None,   # extension: handle from prepare_conditioning() instead of texts / text_embeds
    ):
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

    def _resolve_text(self, texts, text_embeds, text_masks, device):
        """ip.py:2326-2337: texts -> encoder hook; default mask = any non-zero feature."""
        if texts is not None and text_embeds is None and not self.unconditional:
            assert all([*map(len, texts)]), 'text cannot be empty'
            text_embeds, text_masks = self.encode_text(texts, return_attn_mask=True)
        if not self.unconditional:
            assert text_embeds is not None, 'text must be passed in if the network was not trained without text `condition_on_text` must be set to `False` when training'
            text_embeds = text_embeds.to(device)
            text_masks = text_masks.to(device) if text_masks is not None else torch.any(text_embeds != 0., dim=-1)
        return text_embeds, text_masks

    @torch.no_grad()
    def prepare_conditioning(self, texts: Optional[List[str]] = None, *, text_embeds=None, text_masks=None, batch_size: int = 1,
                             device=None) -> Conditioning:
        """Encode / stage the prompts once; reuse the returned handle with `sample(conditioning=handle, ...)`."""
        device = torch.device(device) if device is not None else self.device
        text_embeds, text_masks = self._resolve_text(texts, text_embeds, text_masks, device)
        assert not (self.condition_on_text and text_embeds is None), 'text or text encodings must be passed into imagen if specified'
        assert not (not self.condition_on_text and text_embeds is not None), 'imagen specified not to be conditioned on text, yet it is presented'
        assert not (text_embeds is not None and text_embeds.shape[-1] != self.text_embed_dim), \
            f'invalid text embedding dimension being passed in (should be {self.text_embed_dim})'
        return Conditioning(text_embeds, text_masks, text_embeds.shape[0] if text_embeds is not None else batch_size)

    def _sample(self, texts, text_masks, text_embeds, video_frames, cond_images, cond_video_frames, post_cond_video_frames,
                inpaint_videos, inpaint_images, inpaint_masks, inpaint_resample_times, init_images, skip_steps, batch_size, cond_scale,
                lowres_sample_noise_level, start_at_unet_number, start_image_or_video, stop_at_unet_number, return_all_unet_outputs,
                return_pil_images, device, use_tqdm, noise_fn, seed, sample_offset, use_graph, max_steps):
        device = torch.device(device) if device is not None else self.device
        if device.type not in _SAMPLING_DEVICE_TYPES:
            raise RuntimeError("imagen_pytorch_amd.Imagen.sample runs on MI355X only (move the module to 'cuda'); there is no CPU path")
        self.reset_unets_all_one_device(device)
        if inpaint_videos is not None:                   # ip.py:2342: the video argument wins
            inpaint_images = inpaint_videos
        if not self.is_video:                            # ip.py:2417-2427: only video cascades hand the prompt frames to their unets
            cond_video_frames = post_cond_video_frames = None
        if cond_images is not None:
            cond_images = cond_images.to(device)
            if cond_images.dtype == torch.uint8:         # cast_uint8_images_to_float, ip.py:2324
                cond_images = cond_images.float() / 255
        assert not (self.is_video and video_frames is None), 'video_frames must be passed in on sample time if training on video'   # ip.py:2381
        if self.is_video:
            for f in self.temporal_downsample_factor:                       # calc_all_frame_dims, ip.py:170-183
                assert int(video_frames) % f == 0, f'video_frames {video_frames} must be divisible by the temporal downsample factor {f}'
        frames = int(video_frames) if self.is_video else 0
        to_internal = (lambda t: t.permute(0, 2, 1, 3, 4).contiguous()) if self.is_video else (lambda t: t)   # (b,c,f,h,w) <-> (b,f,c,h,w)
        assert not (return_pil_images and self.is_video), 'converting sampled video tensor to video file is not supported yet'   # ip.py:2494

        text_embeds, text_masks = self._resolve_text(texts, text_embeds, text_masks, device)
        if not self.unconditional:
            batch_size = text_embeds.shape[0]
        if inpaint_images is not None:                   # ip.py:2344-2351
            if self.unconditional and batch_size == 1:
                batch_size = inpaint_images.shape[0]
            assert inpaint_images.shape[0] == batch_size, \
                'number of inpainting images must be equal to the specified batch size on sample `sample(batch_size=<int>)``'
            assert not (self.condition_on_text and inpaint_images.shape[0] != text_embeds.shape[0]), \
                'number of inpainting images must be equal to the number of text to be conditioned on'
        assert not (self.condition_on_text and text_embeds is None), 'text or text encodings must be passed into imagen if specified'
        assert not (not self.condition_on_text and text_embeds is not None), 'imagen specified not to be conditioned on text, yet it is presented'
        assert not (text_embeds is not None and text_embeds.shape[-1] != self.text_embed_dim), \
            f'invalid text embedding dimension being passed in (should be {self.text_embed_dim})'

        assert not ((inpaint_images is not None) ^ (inpaint_masks is not None)), \
            'inpaint images and masks must be both passed in to do inpainting'
        num_unets = len(self.unets)
        normalize = (lambda im: im * 2 - 1) if self.auto_normalize_img else (lambda im: im)             # ip.py:1885-1888
        assert not (self.is_video and self.resize_mode != 'nearest'), 'video cascades resize with nearest only in this build'
        resize = lambda im, size: im if im.shape[-1] == size else F.interpolate(im, size, mode=self.resize_mode)   # ip.py:152-168, 1924

        def resize_clip(v, size, f):
            """resize_video_to (iv.py:134-156) of a (b, c, f', h, w) clip to f frames of size x size — nearest over all three axes —
            returned in the internal frame-major layout."""
            if tuple(v.shape[-3:]) != (f, size, size):
                v = F.interpolate(v, (f, size, size), mode='nearest')
            return to_internal(v)

        known = known_mask = None
        if inpaint_images is not None:                   # ip.py:2217-2220
            known = normalize(inpaint_images.to(device).float())
            known_mask = inpaint_masks.to(device)
            if self.is_video:                            # ip.py:2373-2379
                assert known.ndim == 5, 'inpaint_videos must be (b, c, f, h, w)'
                if known_mask.ndim == 3:
                    known_mask = known_mask[:, None].expand(-1, frames, -1, -1)
                assert known_mask.shape[1] == frames
            known_mask = known_mask[:, None].float()
        init_images = [None if im is None else normalize(im.to(device).float()) for im in _cast_tuple(init_images, num_unets)]  # ip.py:2390-2391
        skip_steps = _cast_tuple(skip_steps, num_unets)
        level = lowres_sample_noise_level if lowres_sample_noise_level is not None else self.lowres_sample_noise_level
        cond_scale = _cast_tuple(cond_scale, num_unets)
        if seed is None:
            seed = self._next_seed()

        img = None
        if start_at_unet_number > 1:
            assert start_at_unet_number <= num_unets, 'must start a unet that is less than the total number of unets'
            assert stop_at_unet_number is None or start_at_unet_number <= stop_at_unet_number
            assert start_image_or_video is not None, 'starting image or video must be supplied if only doing upscaling'
            img = to_internal(start_image_or_video.to(device).float()).contiguous()
            prev_size = self.image_sizes[start_at_unet_number - 2]
            assert img.shape[-3] == self.channels, f'start image must have {self.channels} channels'
            if img.shape[-1] != prev_size:               # ip.py:2400-2404: first to the PREVIOUS stage's size (lowres_prep then resizes to this one's)
                lead = img.shape[:-3]
                img = resize(img.reshape(-1, *img.shape[-3:]), prev_size).reshape(*lead, self.channels, prev_size, prev_size).contiguous()

        stream = self._streams.get((self._lane, str(device)))
        if stream is None:
            stream = self._streams[(self._lane, str(device))] = torch.cuda.Stream(device=device)
        outputs = []
        # inputs were produced on the caller's stream: order this lane's stream behind it (no device-wide drain — other lanes keep running)
        stream.wait_stream(torch.cuda.current_stream(device))
        with torch.cuda.device(device), torch.cuda.stream(stream):
            for idx in range(num_unets):
                unet_number = idx + 1
                if unet_number < start_at_unet_number:
                    continue
                unet = self.unets[idx]
                assert not isinstance(unet, NullUnet), 'one cannot sample from null / placeholder unets'
                cs = cond_scale[idx]
                assert not (cs != 1. and not self.can_classifier_guidance), \
                    'imagen was not trained with conditional dropout, and thus one cannot use classifier free guidance (cond_scale anything other than 1)'
                with_text = text_embeds is not None and unet.cond_on_text
                prompts = [None, None]
                if isinstance(unet, Unet3D):                # ip.py:2417-2434: the same prompt frames for every stage, at the stage's frame rate
                    tds = self.temporal_downsample_factor[idx]
                    for k, v in enumerate((cond_video_frames, post_cond_video_frames)):
                        if v is not None and self.resize_cond_video_frames and tds != 1:     # scale_video_time, iv.py:158-178
                            assert v.shape[2] % tds == 0, f'trying to temporally downsample a conditioning video frames of length ' \
                                                          f'{v.shape[2]} by {tds}, however it is not neatly divisible'
                            v = F.interpolate(v.float(), (v.shape[2] // tds, v.shape[-2], v.shape[-1]), mode='nearest')
                        prompts[k] = v
                st = self._stage(idx, batch_size, device, cond_scale=cs, with_text=with_text, inject_noise=noise_fn is not None,
                                 sample_offset=sample_offset, resample_times=inpaint_resample_times if known is not None else 0,
                                 frames=frames // self.temporal_downsample_factor[idx] if isinstance(unet, Unet3D) else 0,
                                 prompt_frames=tuple(0 if v is None else v.shape[2] for v in prompts))
                st['sample_offset'] = sample_offset
                eng = st['eng']
                S = self.image_sizes[idx]
                assert not (getattr(unet, 'has_cond_image', False) ^ (cond_images is not None)), \
                    'you either requested to condition on an image on the unet, but the conditioning image is not supplied, or vice versa'   # ip.py:1555
                if cond_images is not None:
                    eng.set_cond_images(cond_images)
                if prompts[0] is not None or prompts[1] is not None:
                    eng.set_cond_video_frames(*prompts)
                stage_frames = st['frames']
                if known is not None and st['video']:
                    st['known'].copy_(resize_clip(known, S, stage_frames))
                    st['mask'].copy_(resize_clip(known_mask, S, stage_frames).bool().expand(-1, -1, self.channels, -1, -1))
                elif known is not None:
                    st['known'].copy_(resize(known, S))
                    st['mask'].copy_(resize(known_mask, S).bool().expand(-1, self.channels, -1, -1))
                lowres_logsnr = None
                if unet.lowres_cond:
                    assert img is not None
                    a, s, lsnr = self.lowres_noise_schedule.q_sample_coefficients(level)
                    # Imagen conditions on the log-SNR of the augmentation level (ip.py:2081); ElucidatedImagen.sample passes the
                    # raw level (elucidated_imagen.py:700, 728) — `_lowres_time_raw` is set by that subclass
                    lowres_logsnr = torch.full((batch_size,), level if getattr(self, "_lowres_time_raw", False) else lsnr, dtype=torch.float32)
                    aug = torch.empty_like(eng.lowres_in)
                    if noise_fn is not None:
                        if st.get('video', False):
                            b_, f_, c_, h_, w_ = aug.shape
                            aug.copy_(noise_fn(("lowres", idx), (b_, c_, f_, h_, w_)).permute(0, 2, 1, 3, 4))
                        else:
                            aug.copy_(noise_fn(("lowres", idx), tuple(aug.shape)))
                    prep = Plan("lowres-prep")
                    if noise_fn is None:
                        for sd, r0, cnt, i0 in _seed_spans(seed, batch_size, sample_offset):
                            ops.randn(prep, aug[r0:r0 + cnt], seed=sd, stream_id=idx, tag=TAG_LOWRES, sample_offset=i0)
                    src = img if self.auto_normalize_img else (img + 1) * 0.5                  # kernel normalises [0,1] -> [-1,1]
                    if st.get('video', False) and src.shape[1] != aug.shape[1]:
                        # a stage sampled at a lower frame rate feeds this one: nearest over the frame axis (resize_video_to,
                        # iv.py:134-156; F.interpolate 'nearest' takes source index floor(dst * F_in / F_out)); once per stage
                        f_in, f_out = src.shape[1], aug.shape[1]
                        src = src[:, (torch.arange(f_out, device=src.device) * f_in) // f_out]
                    if self.resize_mode != 'nearest' and src.shape[-1] != S:
                        # LOWRES_PREP resizes with nearest in-kernel; any other mode is resized here, once per stage, and the kernel's
                        # own resize becomes the identity.  The resize acts on the [0, 1] image as in ip.py:2444-2446 (normalisation after).
                        src = resize(img, S) if self.auto_normalize_img else (resize(img, S) + 1) * 0.5
                    src = src.contiguous()
                    # frames are independent images for the nearest resize (resize_video_to with unchanged frame count, iv.py:134-156)
                    as_images = lambda t: t.reshape(-1, *t.shape[-3:])
                    ops.lowres_prep(prep, as_images(src), as_images(aug), as_images(eng.lowres_in), alpha=a, sigma=s)
                    prep.keep += [src, aug, eng.lowres_in]
                    prep.run()
                rows = eng.R
                keep = torch.ones(rows, dtype=torch.bool)
                if rows == 2 * batch_size:
                    keep[batch_size:] = False            # second half = null-conditioned CFG branch (cond_drop_prob = 1, ip.py:1521)
                cond = getattr(self._tls, 'conditioning', None)
                stamp = None if cond is None else (cond.token, None if lowres_logsnr is None else float(lowres_logsnr[0]))
                if stamp is None or getattr(eng, '_cond_stamp', None) != stamp:
                    eng.set_conditioning(text_embeds=text_embeds if with_text else None, text_mask=text_masks if with_text else None,
                                         keep=keep, lowres_noise_times=lowres_logsnr)
                    eng.static_runs = getattr(eng, 'static_runs', 0) + 1
                eng._cond_stamp = stamp
                timing = os.environ.get("IMAGEN_TIMING")
                if timing:
                    stream.synchronize()
                    import time as _time
                    t_stage = _time.perf_counter()
                out = self._run_stage(st, noise_fn=noise_fn, stage=idx, seed=seed, use_graph=use_graph, use_tqdm=use_tqdm,
                                         max_steps=max_steps, skip_steps=skip_steps[idx],
                                         init_images=None if init_images[idx] is None else
                                         (resize_clip(init_images[idx], S, stage_frames) if st['video'] else resize(init_images[idx], S)))
                if timing:
                    stream.synchronize()
                    dt = _time.perf_counter() - t_stage
                    self.last_stage_seconds = getattr(self, "last_stage_seconds", {})
                    self.last_stage_seconds[idx] = dt
                    print(f"[imagen] stage {idx} ({S}x{S}, rows {eng.R}): {dt * 1e3:.1f} ms for {st['T'] if max_steps is None else min(st['T'], max_steps)} steps "
                          f"({len(st['plan'])} launches/step)", flush=True, file=__import__('sys').stderr)
                img = out.clone()                    # internal layout (videos: frame-major), feeds the next stage's low-res conditioning
                if not self.auto_normalize_img:
                    img = img * 2 - 1
                outputs.append(to_internal(img) if st.get('video', False) else img)   # the permutation is its own inverse
                if stop_at_unet_number is not None and stop_at_unet_number == unet_number:
                    break
        stream.synchronize()
        if return_pil_images:                                # ip.py:2488-2498: a list of PIL images (one list per unet with return_all_unet_outputs)
            pil = [_to_pil_images(o) for o in (outputs if return_all_unet_outputs else outputs[-1:])]
            return pil if return_all_unet_outputs else pil[-1]
        return outputs if return_all_unet_outputs else outputs[-1]
