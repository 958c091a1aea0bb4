"""Drop-in `Unet` (constructor kwargs, attributes, `state_dict` layout, `forward` / `forward_with_cond_scale`
signatures of the reference `Unet`, imagen_pytorch/imagen_pytorch.py:1112-1725 = "ip.py") whose arithmetic
runs entirely in the gfx950 kernels of libimagen_hip.so (see engine.py).  There is no PyTorch fallback:
calling `forward` on a CPU tensor, or without the HIP extension, raises.
"""
from __future__ import annotations

from functools import partial
from pathlib import Path
import sys
import torch
from torch import nn

from .modules import (CrossEmbedP, Holder, ParallelP, PerceiverResamplerP, PixelShuffleUpsampleP,
                      ResnetBlockP, SinuPosEmbP, TransformerBlockP, UpsampleCombinerP, downsample_p, upsample_conv_p)

DEFAULT_TEXT_EMBED_DIM = 768  # d_model of the reference's default T5 ('google/t5-v1_1-base', t5.py:47-58, ip.py:1117)

_printed_dim_hint = False


_ENGINE_DEVICE_TYPES = ('cuda',)   # where plans can be launched; tests/test_sample_cpu_replay.py widens it after replacing the launcher


def _cast_tuple(val, length=None):
    if isinstance(val, list):
        val = tuple(val)
    out = val if isinstance(val, tuple) else ((val,) * (length or 1))
    if length is not None:
        assert len(out) == length
    return out


def _unsupported(flag):
    raise NotImplementedError(
        f"Unet({flag}=...) is accepted by the reference but lies outside the MI355X hot-path scope of this build "
        f"(SURVEY.md §2 'optional L1 variants'); no HIP kernel plan exists for it and there is no PyTorch fallback."
    )


class Unet(nn.Module):
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
        cond_images_channels=0,
        channels=3,
        channels_out=None,
        attn_dim_head=64,
        attn_heads=8,
        ff_mult=2.,
        lowres_cond=False,
        layer_attns=True,
        layer_attns_depth=1,
        layer_mid_attns_depth=1,
        layer_attns_add_text_cond=True,
        attend_at_middle=True,
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
        resize_mode='nearest',
        combine_upsample_fmaps=False,
        pixel_shuffle_upsample=True,
    ):
        super().__init__()
        global _printed_dim_hint

        assert attn_heads > 1, 'you need to have more than 1 attention head, ideally at least 4 or 8'
        if dim < 128 and not _printed_dim_hint:
            _printed_dim_hint = True
            print('The base dimension of your u-net should ideally be no smaller than 128 (reference hint, ip.py:1168)', file=sys.stderr)

        # constructor kwargs are kept for cast_model_parameters / persistence (ip.py:1173-1175)
        ctor_kwargs = dict(locals())
        for drop in ('self', '__class__', '_printed_dim_hint'):
            ctor_kwargs.pop(drop, None)
        self._locals = ctor_kwargs

        # ---- scope gate: flags with no kernel plan fail loudly at construction
        for name in ('use_linear_attn', 'use_linear_cross_attn'):
            v = self._locals[name]
            if any(_cast_tuple(v)):
                _unsupported(name)
        if cross_embed_downsample:
            # the reference cannot build this either: partial(CrossEmbedLayer, kernel_sizes=...) is called with (dim_in, dim_out)
            # positionally (ip.py:1315, 1357, 1366), so dim_out collides with kernel_sizes and Unet(...) raises TypeError — no
            # checkpoint with this flag exists
            _unsupported('cross_embed_downsample')
        if attn_dim_head not in (32, 64):
            raise NotImplementedError(f"attn_dim_head = {attn_dim_head}: the attention kernels are built for head dims 64 (every README config) "
                                      "and 32 (the reference's UnetConfig default, configs.py:48-49)")

        self.channels = channels
        self.channels_out = channels_out if channels_out is not None else channels
        init_channels = channels * (1 + int(lowres_cond) + int(self_cond))
        init_dim = init_dim if init_dim is not None else dim
        self.self_cond = self_cond
        self.has_cond_image = cond_images_channels > 0          # ip.py:1191-1194: extra input channels of the init conv
        self.cond_images_channels = cond_images_channels
        init_channels += cond_images_channels

        # initial convolution (ip.py:1198)
        self.init_conv = (CrossEmbedP(init_channels, kernel_sizes=init_cross_embed_kernel_sizes, dim_out=init_dim, stride=1)
                          if init_cross_embed else nn.Conv2d(init_channels, init_dim, init_conv_kernel_size, padding=init_conv_kernel_size // 2))

        dims = [init_dim, *[dim * m for m in dim_mults]]
        in_out = list(zip(dims[:-1], dims[1:]))

        cond_dim = cond_dim if cond_dim is not None else dim
        time_cond_dim = dim * 4 * (2 if lowres_cond else 1)
        self.cond_dim, self.time_cond_dim = cond_dim, time_cond_dim

        # time / noise-level embedding (ip.py:1210-1248)
        def time_nets():
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

        # text conditioning (ip.py:1254-1287)
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
        num_resnet_blocks = _cast_tuple(num_resnet_blocks, num_layers)
        layer_attns = _cast_tuple(layer_attns, num_layers)
        layer_attns_depth = _cast_tuple(layer_attns_depth, num_layers)
        layer_cross_attns = _cast_tuple(layer_cross_attns, num_layers)
        self._layer_cfg = dict(in_out=in_out, num_resnet_blocks=num_resnet_blocks, layer_attns=layer_attns,
                               layer_attns_depth=layer_attns_depth, layer_cross_attns=layer_cross_attns,
                               memory_efficient=memory_efficient, attend_at_middle=attend_at_middle,
                               layer_mid_attns_depth=layer_mid_attns_depth, init_dim=init_dim, dim=dim)

        resnet = partial(ResnetBlockP, **attn_kwargs)

        self.init_resnet_block = (resnet(init_dim, init_dim, time_cond_dim=time_cond_dim, use_gca=use_global_context_attn)
                                  if memory_efficient else None)
        self.skip_connect_scale = 1. if not scale_skip_connection else (2 ** -0.5)

        # down path (ip.py:1338-1374)
        self.downs = nn.ModuleList([])
        self.ups = nn.ModuleList([])
        skip_connect_dims = []
        for ind, ((dim_in, dim_out), n_blocks, l_attn, l_depth, l_cross) in enumerate(
                zip(in_out, num_resnet_blocks, layer_attns, layer_attns_depth, layer_cross_attns)):
            is_last = ind >= (num_layers - 1)
            layer_cond_dim = cond_dim if l_cross else None
            current_dim = dim_in
            pre_downsample = None
            if memory_efficient:
                pre_downsample = downsample_p(dim_in, dim_out)
                current_dim = dim_out
            skip_connect_dims.append(current_dim)
            post_downsample = None
            if not memory_efficient:
                post_downsample = downsample_p(current_dim, dim_out) if not is_last else ParallelP(dim_in, dim_out)
            self.downs.append(nn.ModuleList([
                pre_downsample,
                resnet(current_dim, current_dim, cond_dim=layer_cond_dim, time_cond_dim=time_cond_dim),
                nn.ModuleList([ResnetBlockP(current_dim, current_dim, time_cond_dim=time_cond_dim, use_gca=use_global_context_attn)
                               for _ in range(n_blocks)]),
                (TransformerBlockP(dim=current_dim, depth=l_depth, ff_mult=ff_mult, context_dim=cond_dim, **attn_kwargs)
                 if l_attn else nn.Identity()),
                post_downsample,
            ]))

        # middle (ip.py:1378-1382) — NB: the mid ResnetBlocks are built without attn_kwargs, i.e. always 8 heads x 64
        mid_dim = dims[-1]
        self.mid_block1 = ResnetBlockP(mid_dim, mid_dim, cond_dim=cond_dim, time_cond_dim=time_cond_dim)
        self.mid_attn = TransformerBlockP(mid_dim, depth=layer_mid_attns_depth, **attn_kwargs) if attend_at_middle else None
        self.mid_block2 = ResnetBlockP(mid_dim, mid_dim, cond_dim=cond_dim, time_cond_dim=time_cond_dim)

        # up path (ip.py:1392-1413)
        upsample_fmap_dims = []
        for ind, ((dim_in, dim_out), n_blocks, l_attn, l_depth, l_cross) in enumerate(
                zip(reversed(in_out), reversed(num_resnet_blocks), reversed(layer_attns), reversed(layer_attns_depth),
                    reversed(layer_cross_attns))):
            is_last = ind == (num_layers - 1)
            layer_cond_dim = cond_dim if l_cross else None
            skip_connect_dim = skip_connect_dims.pop()
            upsample_fmap_dims.append(dim_out)
            self.ups.append(nn.ModuleList([
                resnet(dim_out + skip_connect_dim, dim_out, cond_dim=layer_cond_dim, time_cond_dim=time_cond_dim),
                nn.ModuleList([ResnetBlockP(dim_out + skip_connect_dim, dim_out, time_cond_dim=time_cond_dim, use_gca=use_global_context_attn)
                               for _ in range(n_blocks)]),
                (TransformerBlockP(dim=dim_out, depth=l_depth, ff_mult=ff_mult, context_dim=cond_dim, **attn_kwargs)
                 if l_attn else nn.Identity()),
                ((PixelShuffleUpsampleP if pixel_shuffle_upsample else upsample_conv_p)(dim_out, dim_in)
                 if (not is_last or memory_efficient) else nn.Identity()),
            ]))

        # whether to combine the feature maps of all up levels before the final resnet block (ip.py:1415-1422)
        self.upsample_combiner = UpsampleCombinerP(dim, enabled=combine_upsample_fmaps, dim_ins=upsample_fmap_dims, dim_outs=dim)
        self.init_conv_to_final_conv_residual = init_conv_to_final_conv_residual          # ip.py:1426-1427
        if init_conv_to_final_conv_residual:
            assert init_dim == dim, 'init_conv_to_final_conv_residual needs init_dim == dim (the reference concatenates `dim` channels)'
        final_conv_dim = self.upsample_combiner.dim_out + (dim if init_conv_to_final_conv_residual else 0)

        self.final_res_block = ResnetBlockP(final_conv_dim, dim, time_cond_dim=time_cond_dim, use_gca=True) if final_resnet_block else None
        final_conv_dim_in = dim if final_resnet_block else final_conv_dim
        final_conv_dim_in += (channels if lowres_cond else 0)
        self.final_conv = nn.Conv2d(final_conv_dim_in, self.channels_out, final_conv_kernel_size, padding=final_conv_kernel_size // 2)
        nn.init.zeros_(self.final_conv.weight)   # ip.py:1438 — parity tests must de-zero this
        nn.init.zeros_(self.final_conv.bias)

        self.resize_mode = resize_mode
        self._engines = {}

    # ---- cascade plumbing (ip.py:1446-1506) ----------------------------------------------------------

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

    def persist_to_file(self, path):
        path = Path(path)
        path.parents[0].mkdir(exist_ok=True, parents=True)
        config, state_dict = self.to_config_and_state_dict()
        torch.save(dict(config=config, state_dict=state_dict), str(path))

    @classmethod
    def hydrate_from_file(klass, path, trust_checkpoint: bool = False):
        path = Path(path)
        assert path.exists()
        from .checkpoint import _load_checkpoint_file
        pkg = _load_checkpoint_file(path, trust_checkpoint)
        assert 'config' in pkg and 'state_dict' in pkg
        return klass.from_config_and_state_dict(pkg['config'], pkg['state_dict'])

    # ---- execution -------------------------------------------------------------------------------------

    def engine(self, batch_rows: int, src_batch: int, image_size: int, device, with_text: bool = True) -> "UnetEngine":  # noqa: F821
        """Compiled kernel plan for `batch_rows` denoiser rows (= src_batch, or 2*src_batch with CFG) at image_size^2."""
        from .engine import UnetEngine

        device = torch.device(device)
        if device.type not in _ENGINE_DEVICE_TYPES:
            raise RuntimeError("imagen_pytorch_amd.Unet runs on MI355X through libimagen_hip.so only; there is no CPU path")
        key = (batch_rows, src_batch, image_size, device.index or 0, bool(with_text))
        eng = self._engines.get(key)
        if eng is None or eng.stale():
            eng = UnetEngine(self, batch_rows, src_batch, image_size, device, with_text=with_text)
            self._engines[key] = eng
        return eng

    def release_engines(self):
        self._engines.clear()

    def forward_with_cond_scale(self, *args, cond_scale=1., **kwargs):
        """ip.py:1510-1522 — the cond and null branches run as ONE 2B-row batch through the kernel plan."""
        if cond_scale == 1:
            return self.forward(*args, **kwargs)
        kwargs.pop('cond_drop_prob', None)
        both = self._run(*args, cfg=True, **kwargs)
        b = both.shape[0] // 2
        logits, null_logits = both[:b], both[b:]
        return null_logits + (logits - null_logits) * cond_scale

    def forward(self, x, time, *, lowres_cond_img=None, lowres_noise_times=None, text_embeds=None, text_mask=None,
                cond_images=None, self_cond=None, cond_drop_prob=0.):
        """ip.py:1524-1725.  `time` / `lowres_noise_times` are log-SNR conditions.  Returns fp32 NCHW."""
        return self._run(x, time, lowres_cond_img=lowres_cond_img, lowres_noise_times=lowres_noise_times, text_embeds=text_embeds,
                         text_mask=text_mask, cond_images=cond_images, self_cond=self_cond, cond_drop_prob=cond_drop_prob, cfg=False)

    @torch.no_grad()
    def _run(self, x, time, *, lowres_cond_img=None, lowres_noise_times=None, text_embeds=None, text_mask=None, cond_images=None,
             cond_drop_prob=0., cfg=False, self_cond=None):
        assert not (self.lowres_cond and lowres_cond_img is None), 'low resolution conditioning image must be present'
        assert not (self.lowres_cond and lowres_noise_times is None), 'low resolution conditioning noise time must be present'
        assert not (self.has_cond_image ^ (cond_images is not None)), \
            'you either requested to condition on an image on the unet, but the conditioning image is not supplied, or vice versa'
        if self.training:
            raise RuntimeError("the MI355X path implements sampling (eval mode) only; call .eval() first")
        B, _, H, W = x.shape
        assert H == W, 'square images only'
        rows = 2 * B if cfg else B
        with_text = bool(self.cond_on_text and text_embeds is not None)
        eng = self.engine(rows, B, H, x.device, with_text=with_text)
        if cfg:
            keep = torch.cat((torch.ones(B, dtype=torch.bool), torch.zeros(B, dtype=torch.bool)))
        elif cond_drop_prob == 0:
            keep = torch.ones(B, dtype=torch.bool)
        elif cond_drop_prob == 1:
            keep = torch.zeros(B, dtype=torch.bool)
        else:
            keep = torch.rand(B) < (1 - cond_drop_prob)   # ip.py:201-207
        eng.set_conditioning(text_embeds=text_embeds if with_text else None, text_mask=text_mask, keep=keep,
                             lowres_noise_times=lowres_noise_times)
        if cond_images is not None:
            eng.set_cond_images(cond_images)
        if self.self_cond:
            eng.set_self_cond(self_cond)             # None: zeros (ip.py:1542); a unet built without self_cond ignores the argument, as the reference does
        out = eng.forward(x.float().contiguous(), time.float().contiguous(),
                          lowres_cond_img=None if lowres_cond_img is None else lowres_cond_img.float().contiguous())
        return out.clone()


class NullUnet(nn.Module):
    """ip.py:1729-1739."""

    def __init__(self, *args, **kwargs):
        super().__init__()
        self.lowres_cond = False
        self.dummy_parameter = nn.Parameter(torch.tensor([0.]))

    def cast_model_parameters(self, *args, **kwargs):
        return self

    def forward(self, x, *args, **kwargs):
        return x


class BaseUnet64(Unet):
    """ip.py:1743-1755."""

    def __init__(self, *args, **kwargs):
        defaults = dict(dim=512, dim_mults=(1, 2, 3, 4), num_resnet_blocks=3, layer_attns=(False, True, True, True),
                        layer_cross_attns=(False, True, True, True), attn_heads=8, ff_mult=2., memory_efficient=False)
        super().__init__(*args, **{**defaults, **kwargs})


class SRUnet256(Unet):
    """ip.py:1757-1769."""

    def __init__(self, *args, **kwargs):
        defaults = dict(dim=128, dim_mults=(1, 2, 4, 8), num_resnet_blocks=(2, 4, 8, 8), layer_attns=(False, False, False, True),
                        layer_cross_attns=(False, False, False, True), attn_heads=8, ff_mult=2., memory_efficient=True)
        super().__init__(*args, **{**defaults, **kwargs})


class SRUnet1024(Unet):
    """ip.py:1771-1783."""

    def __init__(self, *args, **kwargs):
        defaults = dict(dim=128, dim_mults=(1, 2, 4, 8), num_resnet_blocks=(2, 4, 8, 8), layer_attns=False,
                        layer_cross_attns=(False, False, False, True), attn_heads=8, ff_mult=2., memory_efficient=True)
        super().__init__(*args, **{**defaults, **kwargs})
