"""Drop-in `Unet3D` (Imagen-Video denoiser): constructor kwargs, attributes and `state_dict` layout of the reference
`Unet3D` (imagen_pytorch/imagen_video.py:1225-1941 = "iv.py"), SURVEY.md §8(f) NEXT-2.

STATUS: the parameter surface (this file, modules3d.py) is checked on CPU against the live reference's state_dict; the kernel
plan is built by engine3d.py and checked on CPU through the plan interpreter against oracle/unet3d_oracle.py; the two kernels the
video path adds (csrc/temporal.hip) and the whole path have NOT been run on a GPU yet — see DESIGN.md §8 NEXT-2.
As for `Unet`, there is no PyTorch fallback: `forward` needs libimagen_hip.so and a CUDA/HIP tensor.
"""
from __future__ import annotations

import functools
import operator
from functools import partial

import torch
from torch import nn

from .modules import Holder, PerceiverResamplerP, SinuPosEmbP
from .modules3d import (Attention3dP, CrossEmbed3dP, Parallel3dP, PixelShuffleUpsample3dP, ResidualP, ResnetBlock3dP,
                        TemporalPixelShuffleUpsampleP, TransformerBlock3dP, conv_frames_p, downsample3d_p, temporal_attn_p,
                        temporal_downsample_p, temporal_peg_p)
from .unet import DEFAULT_TEXT_EMBED_DIM, _cast_tuple, _unsupported


class Unet3D(nn.Module):
    def __init__(
        self,
        *,
        dim,
        text_embed_dim=DEFAULT_TEXT_EMBED_DIM,
        num_resnet_blocks=1,
        cond_dim=None,
        num_image_tokens=4,
        num_time_tokens=2,
        learned_sinu_pos_emb_dim=16,
        out_dim=None,
        dim_mults=(1, 2, 4, 8),
        temporal_strides=1,
        cond_images_channels=0,
        channels=3,
        channels_out=None,
        attn_dim_head=64,
        attn_heads=8,
        ff_mult=2.,
        ff_time_token_shift=True,
        lowres_cond=False,
        layer_attns=False,
        layer_attns_depth=1,
        layer_attns_add_text_cond=True,
        attend_at_middle=True,
        time_rel_pos_bias_depth=2,
        time_causal_attn=True,
        layer_cross_attns=True,
        use_linear_attn=False,
        use_linear_cross_attn=False,
        cond_on_text=True,
        max_text_len=256,
        init_dim=None,
        init_conv_kernel_size=7,
        init_cross_embed=True,
        init_cross_embed_kernel_sizes=(3, 7, 15),
        cross_embed_downsample=False,
        cross_embed_downsample_kernel_sizes=(2, 4),
        attn_pool_text=True,
        attn_pool_num_latents=32,
        dropout=0.,
        memory_efficient=False,
        init_conv_to_final_conv_residual=False,
        use_global_context_attn=True,
        scale_skip_connection=True,
        final_resnet_block=True,
        final_conv_kernel_size=3,
        self_cond=False,
        combine_upsample_fmaps=False,
        pixel_shuffle_upsample=True,
        resize_mode='nearest',
    ):
        super().__init__()
        assert attn_heads > 1, 'you need to have more than 1 attention head, ideally at least 4 or 8'

        ctor_kwargs = dict(locals())
        for drop in ('self', '__class__'):
            ctor_kwargs.pop(drop, None)
        self._locals = ctor_kwargs                                    # iv.py:1291-1293

        for name in ('use_linear_attn', 'use_linear_cross_attn'):
            if any(_cast_tuple(self._locals[name])):
                _unsupported(name)
        for name in ('cross_embed_downsample', 'self_cond', 'combine_upsample_fmaps', 'init_conv_to_final_conv_residual'):
            if self._locals[name]:
                _unsupported(name)
        if not pixel_shuffle_upsample:
            _unsupported('pixel_shuffle_upsample=False')
        if attn_dim_head != 64:
            raise NotImplementedError("the attention kernels are specialised for dim_head = 64")

        self.self_cond = self_cond
        self.channels = channels
        self.channels_out = channels_out if channels_out is not None else channels
        init_channels = channels * (1 + int(lowres_cond))
        init_dim = init_dim if init_dim is not None else dim
        self.has_cond_image = cond_images_channels > 0            # iv.py:1307-1310: extra input channels of the init conv
        self.cond_images_channels = cond_images_channels
        init_channels += cond_images_channels

        self.init_conv = (CrossEmbed3dP(init_channels, kernel_sizes=init_cross_embed_kernel_sizes, dim_out=init_dim, stride=1)
                          if init_cross_embed else conv_frames_p(init_channels, init_dim, init_conv_kernel_size, padding=init_conv_kernel_size // 2))

        dims = [init_dim, *[dim * m for m in dim_mults]]
        in_out = list(zip(dims[:-1], dims[1:]))
        cond_dim = cond_dim if cond_dim is not None else dim
        time_cond_dim = dim * 4 * (2 if lowres_cond else 1)
        self.cond_dim, self.time_cond_dim = cond_dim, time_cond_dim

        def time_nets():                                               # iv.py:1326-1369
            hiddens = nn.Sequential(SinuPosEmbP(learned_sinu_pos_emb_dim), nn.Linear(learned_sinu_pos_emb_dim + 1, time_cond_dim), nn.SiLU())
            cond = nn.Sequential(nn.Linear(time_cond_dim, time_cond_dim))
            tokens = nn.Sequential(nn.Linear(time_cond_dim, cond_dim * num_time_tokens), nn.Identity())
            return hiddens, cond, tokens

        self.to_time_hiddens, self.to_time_cond, self.to_time_tokens = time_nets()
        self.num_time_tokens = num_time_tokens
        self.lowres_cond = lowres_cond
        if lowres_cond:
            self.to_lowres_time_hiddens, self.to_lowres_time_cond, self.to_lowres_time_tokens = time_nets()
        self.norm_cond = nn.LayerNorm(cond_dim)

        self.text_to_cond = None
        if cond_on_text:
            assert text_embed_dim is not None, 'text_embed_dim must be given to the unet if cond_on_text is True'
            self.text_to_cond = nn.Linear(text_embed_dim, cond_dim)
        self.cond_on_text = cond_on_text
        self.attn_pool = (PerceiverResamplerP(dim=cond_dim, depth=2, dim_head=attn_dim_head, heads=attn_heads, num_latents=attn_pool_num_latents)
                          if attn_pool_text else None)
        self.max_text_len = max_text_len
        self.null_text_embed = nn.Parameter(torch.randn(1, max_text_len, cond_dim))
        self.null_text_hidden = nn.Parameter(torch.randn(1, time_cond_dim))
        self.to_text_non_attn_cond = None
        if cond_on_text:
            self.to_text_non_attn_cond = nn.Sequential(nn.LayerNorm(cond_dim), nn.Linear(cond_dim, time_cond_dim), nn.SiLU(),
                                                       nn.Linear(time_cond_dim, time_cond_dim))

        attn_kwargs = dict(heads=attn_heads, dim_head=attn_dim_head)
        num_layers = len(in_out)
        temporal_peg = temporal_peg_p                                   # iv.py:1413-1416
        temporal_attn = lambda d: temporal_attn_p(d, attn_heads, attn_dim_head, time_causal_attn, time_rel_pos_bias_depth)
        self.time_causal_attn, self.ff_time_token_shift = time_causal_attn, ff_time_token_shift

        num_resnet_blocks = _cast_tuple(num_resnet_blocks, num_layers)
        resnet = partial(ResnetBlock3dP, **attn_kwargs)
        layer_attns = _cast_tuple(layer_attns, num_layers)
        layer_attns_depth = _cast_tuple(layer_attns_depth, num_layers)
        layer_cross_attns = _cast_tuple(layer_cross_attns, num_layers)
        temporal_strides = _cast_tuple(temporal_strides, num_layers)
        self.total_temporal_divisor = functools.reduce(operator.mul, temporal_strides, 1)
        self._layer_cfg = dict(in_out=in_out, num_resnet_blocks=num_resnet_blocks, layer_attns=layer_attns, layer_attns_depth=layer_attns_depth,
                               layer_cross_attns=layer_cross_attns, temporal_strides=temporal_strides, memory_efficient=memory_efficient,
                               attend_at_middle=attend_at_middle, init_dim=init_dim, dim=dim)

        self.init_resnet_block = (resnet(init_dim, init_dim, time_cond_dim=time_cond_dim, use_gca=use_global_context_attn)
                                  if memory_efficient else None)
        self.init_temporal_peg = temporal_peg(init_dim)
        self.init_temporal_attn = temporal_attn(init_dim)
        self.skip_connect_scale = 1. if not scale_skip_connection else (2 ** -0.5)

        self.downs = nn.ModuleList([])
        self.ups = nn.ModuleList([])
        skip_connect_dims = []
        tb = partial(TransformerBlock3dP, ff_mult=ff_mult, ff_time_token_shift=ff_time_token_shift, context_dim=cond_dim, **attn_kwargs)
        for ind, ((dim_in, dim_out), n_blocks, l_attn, l_depth, l_cross, t_stride) in enumerate(
                zip(in_out, num_resnet_blocks, layer_attns, layer_attns_depth, layer_cross_attns, temporal_strides)):   # iv.py:1449-1477
            is_last = ind >= (num_layers - 1)
            layer_cond_dim = cond_dim if l_cross else None
            current_dim = dim_in
            pre_downsample = None
            if memory_efficient:
                pre_downsample = downsample3d_p(dim_in, dim_out)
                current_dim = dim_out
            skip_connect_dims.append(current_dim)
            post_downsample = None
            if not memory_efficient:
                post_downsample = downsample3d_p(current_dim, dim_out) if not is_last else Parallel3dP(dim_in, dim_out)
            self.downs.append(nn.ModuleList([
                pre_downsample,
                resnet(current_dim, current_dim, cond_dim=layer_cond_dim, time_cond_dim=time_cond_dim),
                nn.ModuleList([ResnetBlock3dP(current_dim, current_dim, time_cond_dim=time_cond_dim, use_gca=use_global_context_attn)
                               for _ in range(n_blocks)]),
                tb(dim=current_dim, depth=l_depth) if l_attn else nn.Identity(),
                temporal_peg(current_dim),
                temporal_attn(current_dim),
                temporal_downsample_p(current_dim, stride=t_stride) if t_stride > 1 else None,
                post_downsample,
            ]))

        mid_dim = dims[-1]                                              # iv.py:1481-1487
        self.mid_block1 = ResnetBlock3dP(mid_dim, mid_dim, cond_dim=cond_dim, time_cond_dim=time_cond_dim)
        self.mid_attn = ResidualP(Attention3dP(mid_dim, **attn_kwargs)) if attend_at_middle else None
        self.mid_temporal_peg = temporal_peg(mid_dim)
        self.mid_temporal_attn = temporal_attn(mid_dim)
        self.mid_block2 = ResnetBlock3dP(mid_dim, mid_dim, cond_dim=cond_dim, time_cond_dim=time_cond_dim)

        for ind, ((dim_in, dim_out), n_blocks, l_attn, l_depth, l_cross, t_stride) in enumerate(
                zip(reversed(in_out), reversed(num_resnet_blocks), reversed(layer_attns), reversed(layer_attns_depth),
                    reversed(layer_cross_attns), reversed(temporal_strides))):                                           # iv.py:1499-1519
            is_last = ind == (num_layers - 1)
            layer_cond_dim = cond_dim if l_cross else None
            skip_connect_dim = skip_connect_dims.pop()
            self.ups.append(nn.ModuleList([
                resnet(dim_out + skip_connect_dim, dim_out, cond_dim=layer_cond_dim, time_cond_dim=time_cond_dim),
                nn.ModuleList([ResnetBlock3dP(dim_out + skip_connect_dim, dim_out, time_cond_dim=time_cond_dim, use_gca=use_global_context_attn)
                               for _ in range(n_blocks)]),
                tb(dim=dim_out, depth=l_depth) if l_attn else nn.Identity(),
                temporal_peg(dim_out),
                temporal_attn(dim_out),
                TemporalPixelShuffleUpsampleP(dim_out, stride=t_stride) if t_stride > 1 else None,
                PixelShuffleUpsample3dP(dim_out, dim_in) if (not is_last or memory_efficient) else nn.Identity(),
            ]))

        self.upsample_combiner = Holder()
        self.init_conv_to_final_conv_residual = False
        self.final_res_block = ResnetBlock3dP(dim, dim, time_cond_dim=time_cond_dim, use_gca=True) if final_resnet_block else None
        final_conv_dim_in = dim + (channels if lowres_cond else 0)
        self.final_conv = conv_frames_p(final_conv_dim_in, self.channels_out, final_conv_kernel_size, padding=final_conv_kernel_size // 2)
        nn.init.zeros_(self.final_conv.weight)                          # iv.py:1578
        nn.init.zeros_(self.final_conv.bias)
        self.resize_mode = resize_mode
        self._engines = {}

    # ---- cascade plumbing (iv.py:1586-1634) -----------------------------------------------------------
    def cast_model_parameters(self, *, lowres_cond, text_embed_dim, channels, channels_out, cond_on_text):
        if (lowres_cond == self.lowres_cond and channels == self.channels and cond_on_text == self.cond_on_text
                and text_embed_dim == self._locals['text_embed_dim'] and channels_out == self.channels_out):
            return self
        updated = dict(lowres_cond=lowres_cond, text_embed_dim=text_embed_dim, channels=channels, channels_out=channels_out,
                       cond_on_text=cond_on_text)
        return self.__class__(**{**self._locals, **updated})

    def to_config_and_state_dict(self):
        return self._locals, self.state_dict()

    @classmethod
    def from_config_and_state_dict(klass, config, state_dict):
        unet = klass(**config)
        unet.load_state_dict(state_dict)
        return unet

    # ---- execution ------------------------------------------------------------------------------------
    def engine(self, batch_rows: int, src_batch: int, frames: int, image_size: int, device, with_text: bool = True, ignore_time: bool = False,
               pre_frames: int = 0, post_frames: int = 0):
        from .engine3d import UnetEngine3D

        from . import unet as _unet_mod
        device = torch.device(device)
        if device.type not in _unet_mod._ENGINE_DEVICE_TYPES:       # ('cuda',) in the product; the CPU replay tests widen it
            raise RuntimeError("imagen_pytorch_amd.Unet3D runs on MI355X through libimagen_hip.so only; there is no CPU path")
        key = (batch_rows, src_batch, frames, image_size, device.index or 0, bool(with_text), bool(ignore_time), pre_frames, post_frames)
        eng = self._engines.get(key)
        if eng is None or eng.stale():
            eng = UnetEngine3D(self, batch_rows, src_batch, frames, image_size, device, with_text=with_text, ignore_time=ignore_time,
                               pre_frames=pre_frames, post_frames=post_frames)
            self._engines[key] = eng
        return eng

    def release_engines(self):
        self._engines.clear()

    def forward_with_cond_scale(self, *args, cond_scale=1., **kwargs):
        """iv.py:1636-1648 — the cond and null branches run as ONE 2B-row batch."""
        if cond_scale == 1:
            return self.forward(*args, **kwargs)
        kwargs.pop('cond_drop_prob', None)
        both = self._run(*args, cfg=True, **kwargs)
        b = both.shape[0] // 2
        logits, null_logits = both[:b], both[b:]
        return null_logits + (logits - null_logits) * cond_scale

    def forward(self, x, time, *, lowres_cond_img=None, lowres_noise_times=None, text_embeds=None, text_mask=None, cond_images=None,
                cond_video_frames=None, post_cond_video_frames=None, self_cond=None, cond_drop_prob=0., ignore_time=False):
        """iv.py:1650-1941.  x: (b, c, f, h, w); returns fp32 (b, c_out, f, h, w)."""
        return self._run(x, time, lowres_cond_img=lowres_cond_img, lowres_noise_times=lowres_noise_times, text_embeds=text_embeds,
                         text_mask=text_mask, cond_drop_prob=cond_drop_prob, ignore_time=ignore_time, cfg=False, cond_images=cond_images,
                         cond_video_frames=cond_video_frames, post_cond_video_frames=post_cond_video_frames, self_cond=self_cond)

    @torch.no_grad()
    def _run(self, x, time, *, lowres_cond_img=None, lowres_noise_times=None, text_embeds=None, text_mask=None, cond_drop_prob=0.,
             ignore_time=False, cfg=False, cond_images=None, cond_video_frames=None, post_cond_video_frames=None, self_cond=None):
        assert self_cond is None, 'self_cond: Unet3D is built without self-conditioning in this build'
        assert not (self.has_cond_image ^ (cond_images is not None)), \
            'you either requested to condition on an image on the unet, but the conditioning image is not supplied, or vice versa'   # iv.py:1722
        if cond_images is not None:
            assert cond_images.ndim == 4, 'conditioning images must have 4 dimensions only, if you want to condition on frames of video, ' \
                                          'use `cond_video_frames` instead'                                                   # iv.py:1725
        assert x.ndim == 5, 'input to 3d unet must have 5 dimensions (batch, channels, time, height, width)'
        assert not (self.lowres_cond and lowres_cond_img is None), 'low resolution conditioning image must be present'
        assert not (self.lowres_cond and lowres_noise_times is None), 'low resolution conditioning noise time must be present'
        if self.training:
            raise RuntimeError("the MI355X path implements sampling (eval mode) only; call .eval() first")
        B, _, Fr, H, W = x.shape
        assert H == W, 'square frames only'
        assert ignore_time or Fr % self.total_temporal_divisor == 0, \
            f'number of input frames {Fr} must be divisible by {self.total_temporal_divisor}'
        rows = 2 * B if cfg else B
        with_text = bool(self.cond_on_text and text_embeds is not None)
        n_pre = 0 if cond_video_frames is None else cond_video_frames.shape[2]
        n_post = 0 if post_cond_video_frames is None else post_cond_video_frames.shape[2]
        eng = self.engine(rows, B, Fr, H, x.device, with_text=with_text, ignore_time=ignore_time, pre_frames=n_pre, post_frames=n_post)
        if n_pre or n_post:
            eng.set_cond_video_frames(cond_video_frames, post_cond_video_frames)
        if cond_images is not None:
            eng.set_cond_images(cond_images)
        if cfg:
            keep = torch.cat((torch.ones(B, dtype=torch.bool), torch.zeros(B, dtype=torch.bool)))
        elif cond_drop_prob == 0:
            keep = torch.ones(B, dtype=torch.bool)
        elif cond_drop_prob == 1:
            keep = torch.zeros(B, dtype=torch.bool)
        else:
            keep = torch.rand(B) < (1 - cond_drop_prob)
        eng.set_conditioning(text_embeds=text_embeds if with_text else None, text_mask=text_mask, keep=keep,
                             lowres_noise_times=lowres_noise_times)
        # the engine's image layout is frame-major (b, f, c, h, w): frames of one clip are consecutive NHWC images
        to_fm = lambda t: t.float().permute(0, 2, 1, 3, 4).contiguous()
        out = eng.forward(to_fm(x), time.float().contiguous(), lowres_cond_img=None if lowres_cond_img is None else to_fm(lowres_cond_img))
        return out.permute(0, 2, 1, 3, 4).contiguous()
