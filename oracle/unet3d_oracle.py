"""CPU oracle for the Imagen-Video denoiser `Unet3D.forward` (TEST INFRASTRUCTURE — see oracle/unet_oracle.py header).

SURVEY.md §8(f) NEXT-2 groundwork: a functional fp32 restatement of the reference's `Unet3D.forward /
forward_with_cond_scale` (iv.py = imagen_pytorch/imagen_video.py:1225-1941) and of the blocks that differ from the image
Unet — pseudo-3D convolution (spatial 3x3 + CAUSAL temporal conv1d k=3, iv.py:397-451), temporal PEG (depthwise causal
(3,1,1) conv, iv.py:1413-1414), causal temporal attention over the F frames of every pixel with an MLP-generated relative
position bias and a learned null-key bias (iv.py:455-570, 1182-1223, 1416), space-time attention over all F*H*W tokens in the
transformer blocks / the mid block (iv.py:1059-1091, 1508, 1878-1885), channel-layout feed-forward with a time token shift
(iv.py:1039-1057), temporal down / pixel-shuffle up-sampling (iv.py:649-686).  It consumes the reference module's
`state_dict` + constructor kwargs; the text / time conditioning front end is identical to the image Unet's and is shared
with oracle/unet_oracle.py.

No HIP path exists for this row yet: the oracle and its golden fixture (tests/golden/unet3d_tiny.pt, generated from the live
reference by `oracle/make_golden.py --unet3d`) are what the kernels of the next round will be checked against.
Parity status of THIS file: pinned against the live reference (tests/test_oracle_vs_reference.py, container only) and the
fixture (tests/test_oracle_golden.py).  Out of scope: linear attention, cross-embed downsample, self-conditioning,
cond_images, cond_video_frames / post_cond_video_frames, upsample combiner, init->final residual, nearest+conv upsample.
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import torch
import torch.nn.functional as F

from .unet_oracle import (UNET_DEFAULTS, _SD, _tup, affine_layernorm, cross_attention, gain_layernorm, l2n, perceiver_resampler,
                          time_conditioning)

Tensor = torch.Tensor

UNET3D_DEFAULTS = {**{k: v for k, v in UNET_DEFAULTS.items() if k != "layer_mid_attns_depth"},
                   **dict(layer_attns=False, temporal_strides=1, ff_time_token_shift=True, time_rel_pos_bias_depth=2,
                          time_causal_attn=True)}   # iv.py:1226-1278 (layer_attns defaults to False here)


def resolve_config3d(kwargs: dict) -> dict:
    """Constructor defaults and per-level settings (iv.py:1226-1278, 1312-1316, 1420-1438)."""
    cfg = dict(UNET3D_DEFAULTS)
    cfg.update(kwargs)
    for flag in ("use_linear_attn", "use_linear_cross_attn"):
        v = cfg[flag]
        if any(v) if isinstance(v, (list, tuple)) else v:
            raise NotImplementedError(f"oracle: {flag} is outside the scope")
    for flag in ("cross_embed_downsample", "self_cond", "combine_upsample_fmaps", "init_conv_to_final_conv_residual"):
        if cfg[flag]:
            raise NotImplementedError(f"oracle: {flag} is outside the scope")
    if not cfg["pixel_shuffle_upsample"]:
        raise NotImplementedError("oracle: nearest+conv upsample is outside the scope")
    dim, n = cfg["dim"], len(cfg["dim_mults"])
    cfg["init_dim"] = cfg["init_dim"] or dim
    cfg["cond_dim"] = cfg["cond_dim"] or dim
    dims = [cfg["init_dim"]] + [dim * m for m in cfg["dim_mults"]]
    cfg["in_out"] = list(zip(dims[:-1], dims[1:]))
    for name in ("num_resnet_blocks", "layer_attns", "layer_attns_depth", "layer_cross_attns", "temporal_strides"):
        cfg[name + "_t"] = _tup(cfg[name], n)
    cfg["total_temporal_divisor"] = math.prod(cfg["temporal_strides_t"])
    cfg["skip_scale"] = 2 ** -0.5 if cfg["scale_skip_connection"] else 1.0
    return cfg


# ---------------------------------------------------------------- layout helpers (b c f h w)

def to_tokens(x: Tensor) -> Tensor:
    """'b c f h w -> b (f h w) c' (iv.py:1081-1082)."""
    b, c = x.shape[:2]
    return x.reshape(b, c, -1).transpose(1, 2)


def from_tokens(tok: Tensor, like: Tensor) -> Tensor:
    b, c = like.shape[0], tok.shape[-1]
    return tok.transpose(1, 2).reshape(b, c, *like.shape[2:])


def conv_frames(x: Tensor, w: Tensor, b: Optional[Tensor], padding: int = 0) -> Tensor:
    """The reference's `Conv2d(...)` = nn.Conv3d with a (1, k, k) kernel (iv.py:574-588): a per-frame 2-D convolution."""
    return F.conv3d(x, w, b, padding=(0, padding, padding))


def chan_rmsnorm3d(x: Tensor, gamma: Tensor) -> Tensor:
    """iv.py:207-214 — per (b, f, h, w) position over C; gamma is (C, 1, 1, 1)."""
    return F.normalize(x, dim=1, eps=1e-12) * math.sqrt(x.shape[1]) * gamma


def chan_layernorm3d(x: Tensor, g: Tensor, eps: float = 1e-5) -> Tensor:
    """iv.py:216-227 — gain-only LayerNorm over the channel axis (fp32 eps branch); g is (1, C, 1, 1, 1)."""
    var = x.var(dim=1, unbiased=False, keepdim=True)
    mean = x.mean(dim=1, keepdim=True)
    return (x - mean) * torch.rsqrt(var + eps) * g


# ---------------------------------------------------------------- pseudo-3D convolution and resnet blocks

def pseudo_conv3d(p: _SD, x: Tensor, ignore_time: bool = False) -> Tensor:
    """iv.py:397-451 — 3x3 spatial conv per frame, then a CAUSAL temporal conv1d (k = 3, left pad 2) per pixel."""
    b, c, f, h, w = x.shape
    y = F.conv2d(x.transpose(1, 2).reshape(b * f, c, h, w), p("spatial_conv.weight"), p("spatial_conv.bias"), padding=1)
    co = y.shape[1]
    y = y.reshape(b, f, co, h, w).transpose(1, 2)                                   # b c f h w
    if ignore_time or not p.has("temporal_conv.weight"):
        return y
    wt = p("temporal_conv.weight")
    seq = y.permute(0, 3, 4, 1, 2).reshape(b * h * w, co, f)                        # (b h w) c f
    seq = F.conv1d(F.pad(seq, (wt.shape[-1] - 1, 0)), wt, p("temporal_conv.bias"))
    return seq.reshape(b, h, w, co, f).permute(0, 3, 4, 1, 2)


def block3d(p: _SD, x: Tensor, scale_shift=None, ignore_time: bool = False) -> Tensor:
    """iv.py:716-741."""
    h = chan_rmsnorm3d(x, p("norm.gamma"))
    if scale_shift is not None:
        scale, shift = scale_shift
        h = h * (scale + 1.0) + shift
    return pseudo_conv3d(p.sub("project"), F.silu(h), ignore_time)


def global_context_gate3d(p: _SD, x: Tensor) -> Tensor:
    """iv.py:1002-1027 — the softmax pools over ALL f*h*w positions; returns (b, c, 1, 1, 1)."""
    b, c = x.shape[:2]
    logits = conv_frames(x, p("to_k.weight"), p("to_k.bias")).reshape(b, 1, -1)
    pooled = torch.einsum("bin,bcn->bci", logits.softmax(dim=-1), x.reshape(b, c, -1)).reshape(b, c, 1, 1, 1)
    hid = F.silu(conv_frames(pooled, p("net.0.weight"), p("net.0.bias")))
    return torch.sigmoid(conv_frames(hid, p("net.2.weight"), p("net.2.bias")))


def resnet_block3d(p: _SD, x: Tensor, t: Optional[Tensor], cond: Optional[Tensor], ignore_time: bool = False) -> Tensor:
    """iv.py:743-815 — the cross attention runs over all f*h*w tokens of the clip."""
    scale_shift = None
    if p.has("time_mlp.1.weight") and t is not None:
        ss = F.linear(F.silu(t), p("time_mlp.1.weight"), p("time_mlp.1.bias"))
        half = ss.shape[1] // 2
        scale_shift = (ss[:, :half, None, None, None], ss[:, half:, None, None, None])
    h = block3d(p.sub("block1"), x, None, ignore_time)
    if p.has("cross_attn.to_q.weight"):
        assert cond is not None
        tok = to_tokens(h)
        tok = cross_attention(p.sub("cross_attn"), tok, cond, 0) + tok
        h = from_tokens(tok, h)
    h = block3d(p.sub("block2"), h, scale_shift, ignore_time)
    if p.has("gca.to_k.weight"):
        h = h * global_context_gate3d(p.sub("gca"), h)
    res = conv_frames(x, p("res_conv.weight"), p("res_conv.bias")) if p.has("res_conv.weight") else x
    return h + res


# ---------------------------------------------------------------- attention (iv.py:455-570)

def dynamic_position_bias(p: _SD, n: int) -> Tensor:
    """iv.py:1182-1223 — MLP over the signed frame distance -> (heads, n, n) bias, bias[h, i, j] = f(i - j)."""
    pos = torch.arange(-n + 1, n, dtype=torch.float32).reshape(-1, 1)
    depth = 0
    while p.has(f"mlp.{depth}.0.weight"):
        q = p.sub(f"mlp.{depth}")
        pos = F.silu(gain_layernorm(F.linear(pos, q("0.weight"), q("0.bias")), q("1.g")))
        depth += 1
    pos = F.linear(pos, p(f"mlp.{depth}.weight"), p(f"mlp.{depth}.bias"))          # (2n-1, heads)
    idx = torch.arange(n).reshape(-1, 1) - torch.arange(n).reshape(1, -1) + (n - 1)
    return pos[idx].permute(2, 0, 1)


def attention3d(p: _SD, x: Tensor, context: Optional[Tensor] = None, causal: bool = False) -> Tensor:
    """iv.py:499-570 — multi-query cosine-sim attention; key order [context, null, self]; with a relative position bias the
    null key gets its own learned per-head bias (context keys are never combined with a bias in the network); the causal mask
    `triu(j - i + 1)` leaves the null key (and any context) visible to every query."""
    b, n, _ = x.shape
    xn = gain_layernorm(x, p("norm.g"))
    q = F.linear(xn, p("to_q.weight"))
    kv = F.linear(xn, p("to_kv.weight"))
    dh = kv.shape[-1] // 2
    heads = q.shape[-1] // dh
    k, v = kv[..., :dh], kv[..., dh:]
    q = q.reshape(b, n, heads, dh).permute(0, 2, 1, 3)
    null_kv = p("null_kv")
    k = torch.cat((null_kv[0].expand(b, 1, dh), k), dim=1)
    v = torch.cat((null_kv[1].expand(b, 1, dh), v), dim=1)
    if context is not None:
        cn = affine_layernorm(context, p("to_context.0.weight"), p("to_context.0.bias"))
        ckv = F.linear(cn, p("to_context.1.weight"), p("to_context.1.bias"))
        k = torch.cat((ckv[..., :dh], k), dim=1)
        v = torch.cat((ckv[..., dh:], v), dim=1)
    qh = l2n(q) * p("q_scale")
    kh = l2n(k) * p("k_scale")
    sim = torch.einsum("bhid,bjd->bhij", qh, kh) * 8.0
    if p.has("rel_pos_bias.mlp.0.0.weight"):
        assert context is None
        bias = dynamic_position_bias(p.sub("rel_pos_bias"), n)                      # (h, n, n)
        null_bias = p("null_attn_bias").reshape(heads, 1, 1).expand(heads, n, 1)
        sim = sim + torch.cat((null_bias, bias), dim=-1)
    if causal:
        i, j = sim.shape[-2:]
        mask = torch.ones(i, j, dtype=torch.bool).triu(j - i + 1)
        sim = sim.masked_fill(mask, -torch.finfo(sim.dtype).max)
    out = torch.einsum("bhij,bjd->bhid", sim.softmax(dim=-1), v)
    out = out.permute(0, 2, 1, 3).reshape(b, n, heads * dh)
    return gain_layernorm(F.linear(out, p("to_out.0.weight")), p("to_out.1.g"))


def temporal_peg(p: _SD, x: Tensor, causal: bool = True) -> Tensor:
    """iv.py:1413-1414 — x + depthwise conv over 3 frames (causal: two zero frames in front)."""
    pad = (0, 0, 0, 0, 2, 0) if causal else (0, 0, 0, 0, 1, 1)
    return F.conv3d(F.pad(x, pad), p("fn.1.weight"), p("fn.1.bias"), groups=x.shape[1]) + x


def temporal_attention(p: _SD, x: Tensor, causal: bool = True) -> Tensor:
    """iv.py:257-270, 1416 — every pixel's F frames form one sequence: x + Attention(causal, rel_pos_bias)."""
    b, c, f, h, w = x.shape
    seq = x.permute(0, 3, 4, 2, 1).reshape(b * h * w, f, c)
    seq = attention3d(p.sub("fn.fn"), seq, None, causal) + seq
    return seq.reshape(b, h, w, f, c).permute(0, 4, 3, 1, 2)


def chan_feed_forward(p: _SD, x: Tensor, time_token_shift: bool) -> Tensor:
    """iv.py:1039-1057 — ChanLayerNorm, 1x1x1 conv (no bias), GELU, [shift the 2nd half of the channels one frame later],
    ChanLayerNorm, 1x1x1 conv.  Module indices move down by one when the shift is absent (`Sequential` drops None)."""
    h = F.gelu(F.conv3d(chan_layernorm3d(x, p("0.g")), p("1.weight")))
    j = 3
    if time_token_shift:
        half = h.shape[1] // 2 + h.shape[1] % 2          # torch.chunk(2): the first chunk takes the ceiling
        keep, shift = h[:, :half], h[:, half:]
        shift = F.pad(shift, (0, 0, 0, 0, 1, -1))
        h = torch.cat((keep, shift), dim=1)
        j = 4
    return F.conv3d(chan_layernorm3d(h, p(f"{j}.g")), p(f"{j + 1}.weight"))


def transformer_block3d(p: _SD, x: Tensor, context: Optional[Tensor], depth: int, time_token_shift: bool) -> Tensor:
    """iv.py:1059-1091 — attention over all f*h*w tokens (+ context keys), then the channel feed-forward."""
    for d in range(depth):
        lp = p.sub(f"layers.{d}")
        tok = to_tokens(x)
        tok = attention3d(lp.sub("0"), tok, context) + tok
        x = from_tokens(tok, x)
        x = chan_feed_forward(lp.sub("1"), x, time_token_shift) + x
    return x


# ---------------------------------------------------------------- resampling

def downsample3d(p: _SD, x: Tensor) -> Tensor:
    """iv.py:640-645 — 'b c f (h p1) (w p2) -> b (c p1 p2) f h w' then a 1x1 conv."""
    b, c, f, h, w = x.shape
    y = x.reshape(b, c, f, h // 2, 2, w // 2, 2).permute(0, 1, 4, 6, 2, 3, 5).reshape(b, c * 4, f, h // 2, w // 2)
    return conv_frames(y, p("1.weight"), p("1.bias"))


def pixel_shuffle_up3d(p: _SD, x: Tensor) -> Tensor:
    """iv.py:609-638 — 1x1 conv to 4*C_out, SiLU, PixelShuffle(2) per frame."""
    y = F.silu(conv_frames(x, p("net.0.weight"), p("net.0.bias")))
    b, c, f, h, w = y.shape
    y = F.pixel_shuffle(y.transpose(1, 2).reshape(b * f, c, h, w), 2)
    return y.reshape(b, f, c // 4, h * 2, w * 2).transpose(1, 2)


def temporal_downsample(p: _SD, x: Tensor, stride: int) -> Tensor:
    """iv.py:681-686 — 'b c (f p) h w -> b (c p) f h w' then a 1x1 conv."""
    b, c, f, h, w = x.shape
    y = x.reshape(b, c, f // stride, stride, h, w).permute(0, 1, 3, 2, 4, 5).reshape(b, c * stride, f // stride, h, w)
    return conv_frames(y, p("1.weight"), p("1.bias"))


def temporal_pixel_shuffle_up(p: _SD, x: Tensor, stride: int) -> Tensor:
    """iv.py:649-679 — conv1d(C -> C*stride, k = 1) per pixel, SiLU, 'b (c r) n -> b c (n r)'."""
    b, c, f, h, w = x.shape
    seq = x.permute(0, 3, 4, 1, 2).reshape(b * h * w, c, f)
    seq = F.silu(F.conv1d(seq, p("net.0.weight"), p("net.0.bias")))
    co = seq.shape[1] // stride
    seq = seq.reshape(b * h * w, co, stride, f).permute(0, 1, 3, 2).reshape(b * h * w, co, f * stride)
    return seq.reshape(b, h, w, co, f * stride).permute(0, 3, 4, 1, 2)


# ---------------------------------------------------------------- full network

def unet3d_forward(sd: Dict[str, Tensor], kwargs: dict, x: Tensor, time: Tensor, *, lowres_cond_img: Optional[Tensor] = None,
                   lowres_noise_times: Optional[Tensor] = None, text_embeds: Optional[Tensor] = None, text_mask: Optional[Tensor] = None,
                   cond_drop_prob: float = 0.0, ignore_time: bool = False, taps: Optional[dict] = None,
                   cond_video_frames: Optional[Tensor] = None, post_cond_video_frames: Optional[Tensor] = None,
                   cond_images: Optional[Tensor] = None) -> Tensor:
    """iv.py:1650-1941.  x: (b, c, f, h, w); `time` / `lowres_noise_times` are log-SNR values.

    cond_video_frames / post_cond_video_frames (iv.py:1682-1718, 1933-1939) are restated as the reference executes them, including
    its frame order: BOTH prompts are concatenated in FRONT of x ([post, pre, x] — iv.py:1716 prepends the succeeding frames too)
    while the low-res clip read by final_conv is extended as [pre, lowres, post]; the output keeps frames [len(pre), len(pre) + f)."""
    cfg = resolve_config3d(kwargs)
    p = _SD(sd)
    heads = cfg["attn_heads"]
    causal, shift = cfg["time_causal_attn"], cfg["ff_time_token_shift"]
    assert x.ndim == 5, "input to 3d unet must have 5 dimensions (batch, channels, time, height, width)"
    b, frames = x.shape[0], x.shape[2]
    assert ignore_time or frames % cfg["total_temporal_divisor"] == 0
    assert cond_drop_prob in (0.0, 1.0), "oracle: sampling only uses deterministic keep masks (iv.py:182-188)"

    def tap(name, val):
        if taps is not None:
            taps[name] = val.detach().clone()

    if cfg["lowres_cond"]:
        assert lowres_cond_img is not None and lowres_noise_times is not None
    if lowres_cond_img is not None:
        x = torch.cat((x, lowres_cond_img), dim=1)
        if cond_video_frames is not None:                          # iv.py:1686-1688
            lowres_cond_img = torch.cat((cond_video_frames, lowres_cond_img), dim=2)
            cond_video_frames = torch.cat((cond_video_frames, cond_video_frames), dim=1)
        if post_cond_video_frames is not None:                     # iv.py:1690-1692
            lowres_cond_img = torch.cat((lowres_cond_img, post_cond_video_frames), dim=2)
            post_cond_video_frames = torch.cat((post_cond_video_frames, post_cond_video_frames), dim=1)

    def to_size(v):                                                # resize_video_to, frame count unchanged (iv.py:134-156)
        size = x.shape[-1]
        return v if v.shape[-1] == size and v.shape[-2] == size else F.interpolate(v, (v.shape[2], size, size), mode="nearest")

    num_preceding = num_succeeding = 0
    if cond_video_frames is not None:                              # iv.py:1697-1705
        num_preceding = cond_video_frames.shape[2]
        assert num_preceding % cfg["total_temporal_divisor"] == 0
        x = torch.cat((to_size(cond_video_frames), x), dim=2)
    if post_cond_video_frames is not None:                         # iv.py:1710-1718 (prepended as well — reproduced, not corrected)
        num_succeeding = post_cond_video_frames.shape[2]
        assert num_succeeding % cfg["total_temporal_divisor"] == 0
        x = torch.cat((to_size(post_cond_video_frames), x), dim=2)

    assert (cfg["cond_images_channels"] > 0) == (cond_images is not None)                           # iv.py:1722
    if cond_images is not None:                                   # iv.py:1724-1731: one image per sample, repeated over the frames
        assert cond_images.ndim == 4 and cond_images.shape[1] == cfg["cond_images_channels"]
        ci = cond_images[:, :, None].expand(-1, -1, x.shape[2], -1, -1)
        if ci.shape[-1] != x.shape[-1]:
            ci = F.interpolate(ci, (x.shape[2], x.shape[-1], x.shape[-1]), mode=kwargs.get("resize_mode", "nearest"))
        x = torch.cat((ci, x), dim=1)

    if cfg["init_cross_embed"]:                                   # iv.py:1121-1146, 1751
        fmaps = [conv_frames(x, p(f"init_conv.convs.{i}.weight"), p(f"init_conv.convs.{i}.bias"), padding=(ksz - 1) // 2)
                 for i, ksz in enumerate(sorted(cfg["init_cross_embed_kernel_sizes"]))]
        x = torch.cat(fmaps, dim=1)
    else:
        x = conv_frames(x, p("init_conv.weight"), p("init_conv.bias"), padding=cfg["init_conv_kernel_size"] // 2)
    if not ignore_time:                                           # iv.py:1753-1755
        x = temporal_peg(p.sub("init_temporal_peg"), x, causal)
        x = temporal_attention(p.sub("init_temporal_attn"), x, causal)
    tap("init", x)

    t, time_tokens = time_conditioning(p, "", time, cfg["cond_dim"])        # iv.py:1764-1781 (same as the image Unet)
    if cfg["lowres_cond"]:
        lt, ltok = time_conditioning(p, "lowres_", lowres_noise_times, cfg["cond_dim"])
        t = t + lt
        time_tokens = torch.cat((time_tokens, ltok), dim=1)

    text_tokens = None
    if text_embeds is not None and cfg["cond_on_text"]:            # iv.py:1787-1844
        keep = cond_drop_prob == 0.0
        L = cfg["max_text_len"]
        tok = F.linear(text_embeds, p("text_to_cond.weight"), p("text_to_cond.bias"))[:, :L]
        if tok.shape[1] < L:
            tok = F.pad(tok, (0, 0, 0, L - tok.shape[1]))
        if keep:
            mask = torch.ones(b, L, dtype=torch.bool)
            if text_mask is not None:
                mask = text_mask[:, :L]
                if mask.shape[1] < L:
                    mask = F.pad(mask, (0, L - mask.shape[1]), value=False)
        else:
            mask = torch.zeros(b, L, dtype=torch.bool)
        tok = torch.where(mask.unsqueeze(-1), tok, p("null_text_embed").expand(b, -1, -1))
        if cfg["attn_pool_text"]:
            tok = perceiver_resampler(p.sub("attn_pool"), tok, heads)
        text_tokens = tok
        q = p.sub("to_text_non_attn_cond")
        hid = affine_layernorm(tok.mean(dim=1), q("0.weight"), q("0.bias"))
        hid = F.linear(F.silu(F.linear(hid, q("1.weight"), q("1.bias"))), q("3.weight"), q("3.bias"))
        if not keep:
            hid = p("null_text_hidden").expand(b, -1)
        t = t + hid

    c = time_tokens if text_tokens is None else torch.cat((time_tokens, text_tokens), dim=1)
    c = affine_layernorm(c, p("norm_cond.weight"), p("norm_cond.bias"))

    if cfg["memory_efficient"]:
        x = resnet_block3d(p.sub("init_resnet_block"), x, t, None, ignore_time)

    hiddens = []
    n_levels = len(cfg["in_out"])
    for i in range(n_levels):                                     # iv.py:1859-1883
        lp = p.sub(f"downs.{i}")
        if cfg["memory_efficient"]:
            x = downsample3d(lp.sub("0"), x)
        x = resnet_block3d(lp.sub("1"), x, t, c if cfg["layer_cross_attns_t"][i] else None, ignore_time)
        for j in range(cfg["num_resnet_blocks_t"][i]):
            x = resnet_block3d(lp.sub(f"2.{j}"), x, t, None, ignore_time)
            hiddens.append(x)
        if cfg["layer_attns_t"][i]:
            x = transformer_block3d(lp.sub("3"), x, c, cfg["layer_attns_depth_t"][i], shift)
        if not ignore_time:
            x = temporal_peg(lp.sub("4"), x, causal)
            x = temporal_attention(lp.sub("5"), x, causal)
        hiddens.append(x)
        tap(f"down{i}", x)
        if cfg["temporal_strides_t"][i] > 1 and not ignore_time:
            x = temporal_downsample(lp.sub("6"), x, cfg["temporal_strides_t"][i])
        if not cfg["memory_efficient"]:
            if i < n_levels - 1:
                x = downsample3d(lp.sub("7"), x)
            else:                                                 # Parallel(3x3, 1x1), summed (iv.py:1465)
                x = (conv_frames(x, lp("7.fns.0.weight"), lp("7.fns.0.bias"), padding=1)
                     + conv_frames(x, lp("7.fns.1.weight"), lp("7.fns.1.bias")))

    tap("mid_in", x)
    x = resnet_block3d(p.sub("mid_block1"), x, t, c, ignore_time)             # iv.py:1885-1901
    tap("mid_block1", x)
    if cfg["attend_at_middle"]:
        tok = to_tokens(x)
        tok = attention3d(p.sub("mid_attn.fn"), tok) + tok
        x = from_tokens(tok, x)
    tap("mid_attn", x)
    if not ignore_time:
        x = temporal_peg(p.sub("mid_temporal_peg"), x, causal)
        tap("mid_peg", x)
        x = temporal_attention(p.sub("mid_temporal_attn"), x, causal)
        tap("mid_tattn", x)
    x = resnet_block3d(p.sub("mid_block2"), x, t, c, ignore_time)
    tap("mid", x)

    s = cfg["skip_scale"]
    for i in range(n_levels):                                     # iv.py:1907-1926
        lvl = n_levels - 1 - i
        lp = p.sub(f"ups.{i}")
        if cfg["temporal_strides_t"][lvl] > 1 and not ignore_time:
            x = temporal_pixel_shuffle_up(lp.sub("5"), x, cfg["temporal_strides_t"][lvl])
        x = torch.cat((x, hiddens.pop() * s), dim=1)
        x = resnet_block3d(lp.sub("0"), x, t, c if cfg["layer_cross_attns_t"][lvl] else None, ignore_time)
        for j in range(cfg["num_resnet_blocks_t"][lvl]):
            x = torch.cat((x, hiddens.pop() * s), dim=1)
            x = resnet_block3d(lp.sub(f"1.{j}"), x, t, None, ignore_time)
        if cfg["layer_attns_t"][lvl]:
            x = transformer_block3d(lp.sub("2"), x, c, cfg["layer_attns_depth_t"][lvl], shift)
        if not ignore_time:
            x = temporal_peg(lp.sub("3"), x, causal)
            x = temporal_attention(lp.sub("4"), x, causal)
        if i < n_levels - 1 or cfg["memory_efficient"]:
            x = pixel_shuffle_up3d(lp.sub("6"), x)
        tap(f"up{i}", x)

    if cfg["final_resnet_block"]:
        x = resnet_block3d(p.sub("final_res_block"), x, t, None, ignore_time)
    if lowres_cond_img is not None:
        x = torch.cat((x, lowres_cond_img), dim=1)
    out = conv_frames(x, p("final_conv.weight"), p("final_conv.bias"), padding=cfg["final_conv_kernel_size"] // 2)
    if num_preceding > 0:                                          # iv.py:1933-1939
        out = out[:, :, num_preceding:]
    if num_succeeding > 0:
        out = out[:, :, :-num_succeeding]
    return out


def unet3d_forward_with_cond_scale(sd, kwargs, x, time, *, cond_scale: float = 1.0, **kw) -> Tensor:
    """iv.py:1636-1648 — classifier-free guidance: null + (cond - null) * cond_scale."""
    logits = unet3d_forward(sd, kwargs, x, time, **kw)
    if cond_scale == 1:
        return logits
    null_logits = unet3d_forward(sd, kwargs, x, time, **{**kw, "cond_drop_prob": 1.0})
    return null_logits + (logits - null_logits) * cond_scale
