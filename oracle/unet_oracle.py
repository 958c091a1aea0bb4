"""CPU oracle for the Imagen denoiser forward pass (TEST INFRASTRUCTURE — never shipped, never timed as product).

A functional, stateless restatement of the reference's `Unet.forward` /
`Unet.forward_with_cond_scale` (reference: imagen_pytorch/imagen_pytorch.py, cited
per function as `ip.py:<lines>`).  It consumes a plain `state_dict` laid out exactly
like the reference module's, plus the constructor kwargs, and evaluates the network
in fp32 with stock `torch.nn.functional` ops on the CPU.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may
import this module, and only as the checker / the reported CPU baseline.

Pinning: the reference ships no golden vectors for this path (its only tests assert a
step counter).  This restatement is pinned instead against the live
reference imported in the build container (tests/test_oracle_vs_reference.py) and
against fixtures generated from the reference by oracle/make_golden.py
(tests/golden/*.pt), which travel to the GPU box.

Scope: every constructor flag used by the BASELINE.json configs.  Flags outside that
scope (linear attention, cross-embed downsample, upsample combiner) raise.
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import torch
import torch.nn.functional as F

Tensor = torch.Tensor

UNET_DEFAULTS = dict(
    text_embed_dim=768, num_resnet_blocks=1, cond_dim=None, num_image_tokens=4, num_time_tokens=2,
    learned_sinu_pos_emb_dim=16, out_dim=None, dim_mults=(1, 2, 4, 8), cond_images_channels=0, channels=3,
    channels_out=None, attn_dim_head=64, attn_heads=8, ff_mult=2.0, lowres_cond=False, layer_attns=True,
    layer_attns_depth=1, layer_mid_attns_depth=1, layer_attns_add_text_cond=True, attend_at_middle=True,
    layer_cross_attns=True, use_linear_attn=False, use_linear_cross_attn=False, cond_on_text=True,
    max_text_len=256, init_dim=None, init_conv_kernel_size=7, init_cross_embed=True,
    init_cross_embed_kernel_sizes=(3, 7, 15), cross_embed_downsample=False,
    cross_embed_downsample_kernel_sizes=(2, 4), attn_pool_text=True, attn_pool_num_latents=32, dropout=0.0,
    memory_efficient=False, init_conv_to_final_conv_residual=False, use_global_context_attn=True,
    scale_skip_connection=True, final_resnet_block=True, final_conv_kernel_size=3, self_cond=False,
    resize_mode="nearest", combine_upsample_fmaps=False, pixel_shuffle_upsample=True,
)


def _tup(v, n):
    if isinstance(v, (list, tuple)):
        assert len(v) == n
        return tuple(v)
    return (v,) * n


def resolve_config(kwargs: dict) -> dict:
    """Fill constructor defaults (ip.py:1113-1161) and derive per-level settings (ip.py:1200-1206, 1297-1306)."""
    cfg = dict(UNET_DEFAULTS)
    cfg.update(kwargs)
    for flag in ("use_linear_attn", "use_linear_cross_attn"):
        v = cfg[flag]
        if any(v) if isinstance(v, (list, tuple)) else v:
            raise NotImplementedError(f"oracle: {flag} is outside the hot-path scope")
    for flag in ("cross_embed_downsample",):
        if cfg[flag]:
            raise NotImplementedError(f"oracle: {flag} is outside the hot-path scope")
    dim = cfg["dim"]
    n = len(cfg["dim_mults"])
    cfg["init_dim"] = cfg["init_dim"] or dim
    cfg["cond_dim"] = cfg["cond_dim"] or dim
    cfg["channels_out"] = cfg["channels_out"] or cfg["channels"]
    cfg["time_cond_dim"] = dim * 4 * (2 if cfg["lowres_cond"] else 1)
    dims = [cfg["init_dim"]] + [dim * m for m in cfg["dim_mults"]]
    cfg["in_out"] = list(zip(dims[:-1], dims[1:]))
    cfg["num_resnet_blocks_t"] = _tup(cfg["num_resnet_blocks"], n)
    cfg["layer_attns_t"] = _tup(cfg["layer_attns"], n)
    cfg["layer_attns_depth_t"] = _tup(cfg["layer_attns_depth"], n)
    cfg["layer_cross_attns_t"] = _tup(cfg["layer_cross_attns"], n)
    cfg["skip_scale"] = 2 ** -0.5 if cfg["scale_skip_connection"] else 1.0
    return cfg


class _SD:
    """Prefix view over a flat state_dict."""

    def __init__(self, sd: Dict[str, Tensor], prefix: str = ""):
        self.sd, self.prefix = sd, prefix

    def __call__(self, name: str) -> Tensor:
        return self.sd[self.prefix + name]

    def has(self, name: str) -> bool:
        return (self.prefix + name) in self.sd

    def sub(self, name: str) -> "_SD":
        return _SD(self.sd, self.prefix + name + ".")


# ---------------------------------------------------------------- elementary pieces

def chan_rmsnorm(x: Tensor, gamma: Tensor) -> Tensor:
    """ip.py:322-329 — x / max(||x||_2 over C, 1e-12) * sqrt(C) * gamma, per pixel."""
    c = x.shape[1]
    return F.normalize(x, dim=1, eps=1e-12) * math.sqrt(c) * gamma.reshape(1, c, 1, 1)


def gain_layernorm(x: Tensor, g: Tensor, eps: float = 1e-5) -> Tensor:
    """ip.py:331-349 — gain-only LayerNorm over the last dim, biased variance.

    eps is pinned to the fp32 branch of ip.py:345 (1e-5): the oracle contract is fp32 semantics.
    """
    mean = x.mean(dim=-1, keepdim=True)
    var = x.var(dim=-1, unbiased=False, keepdim=True)
    return (x - mean) * torch.rsqrt(var + eps) * g


def affine_layernorm(x: Tensor, w: Tensor, b: Tensor) -> Tensor:
    """nn.LayerNorm(dim) as used at ip.py:394-395, 405, 527, 1252, 1283 (eps 1e-5, with bias)."""
    return F.layer_norm(x, (x.shape[-1],), w, b, 1e-5)


def l2n(t: Tensor) -> Tensor:
    """ip.py:133-134."""
    return F.normalize(t, dim=-1, eps=1e-12)


def block(p: _SD, x: Tensor, scale_shift=None) -> Tensor:
    """ip.py:671-691 — ChanRMSNorm -> optional x*(scale+1)+shift -> SiLU -> 3x3 conv."""
    h = chan_rmsnorm(x, p("norm.gamma"))
    if scale_shift is not None:
        scale, shift = scale_shift
        h = h * (scale + 1.0) + shift
    h = F.silu(h)
    return F.conv2d(h, p("project.weight"), p("project.bias"), padding=1)


def feed_forward(p: _SD, x: Tensor) -> Tensor:
    """ip.py:972-980 — LN -> Linear(no bias) -> GELU(erf) -> LN -> Linear(no bias)."""
    h = gain_layernorm(x, p("0.g"))
    h = F.linear(h, p("1.weight"))
    h = F.gelu(h)
    h = gain_layernorm(h, p("3.g"))
    return F.linear(h, p("4.weight"))


def cosine_attend(q: Tensor, k: Tensor, v: Tensor, q_scale: Tensor, k_scale: Tensor, scale: float = 8.0) -> Tensor:
    """Shared tail of ip.py:559-590 / 812-833 / 424-444: l2-normalised q,k times learned per-dim scales, sim*8, fp32 softmax, AV.

    q: (b, h, i, d); k, v: (b, h|1, j, d).
    """
    q = l2n(q) * q_scale
    k = l2n(k) * k_scale
    sim = torch.matmul(q, k.transpose(-1, -2)) * scale
    attn = sim.softmax(dim=-1)
    return torch.matmul(attn, v)


def self_attention(p: _SD, x: Tensor, heads: int, context: Optional[Tensor]) -> Tensor:
    """ip.py:502-591 — multi-query self attention: one shared k/v head, learned null k/v, optional context k/v in front."""
    b, n, _ = x.shape
    xn = gain_layernorm(x, p("norm.g"))
    q = F.linear(xn, p("to_q.weight"))
    kv = F.linear(xn, p("to_kv.weight"))
    dh = kv.shape[-1] // 2
    k, v = kv[..., :dh], kv[..., dh:]
    heads = q.shape[-1] // dh  # head count is a property of the weights (mid blocks ignore attn_heads, ip.py:1380-1382)
    q = q.reshape(b, n, heads, dh).permute(0, 2, 1, 3)
    null_kv = p("null_kv")
    k = torch.cat((null_kv[0].expand(b, 1, dh), k), dim=1)
    v = torch.cat((null_kv[1].expand(b, 1, dh), v), dim=1)
    if context is not None:
        cn = affine_layernorm(context, p("to_context.0.weight"), p("to_context.0.bias"))
        ckv = F.linear(cn, p("to_context.1.weight"), p("to_context.1.bias"))
        k = torch.cat((ckv[..., :dh], k), dim=1)
        v = torch.cat((ckv[..., dh:], v), dim=1)
    out = cosine_attend(q, k.unsqueeze(1), v.unsqueeze(1), p("q_scale"), p("k_scale"))
    out = out.permute(0, 2, 1, 3).reshape(b, n, heads * dh)
    out = F.linear(out, p("to_out.0.weight"))
    return gain_layernorm(out, p("to_out.1.g"))


def cross_attention(p: _SD, x: Tensor, context: Tensor, heads: int) -> Tensor:
    """ip.py:759-834 — per-head k/v from the conditioning tokens, null k/v first."""
    b, n, _ = x.shape
    xn = gain_layernorm(x, p("norm.g"))
    q = F.linear(xn, p("to_q.weight"))
    kv = F.linear(context, p("to_kv.weight"))
    inner = q.shape[-1]
    null_kv = p("null_kv")
    dh = null_kv.shape[-1]
    heads = inner // dh  # mid_block1/2 are built without attn_kwargs -> always 8 x 64 (ip.py:1380-1382)
    k, v = kv[..., :inner], kv[..., inner:]
    split = lambda t: t.reshape(b, t.shape[1], heads, dh).permute(0, 2, 1, 3)
    q, k, v = split(q), split(k), split(v)
    k = torch.cat((null_kv[0].expand(b, heads, 1, dh), k), dim=2)
    v = torch.cat((null_kv[1].expand(b, heads, 1, dh), v), dim=2)
    out = cosine_attend(q, k, v, p("q_scale"), p("k_scale"))
    out = out.permute(0, 2, 1, 3).reshape(b, n, inner)
    out = F.linear(out, p("to_out.0.weight"))
    return gain_layernorm(out, p("to_out.1.g"))


def global_context_gate(p: _SD, x: Tensor) -> Tensor:
    """ip.py:945-970 — softmax-over-pixels pooled context -> 1x1 -> SiLU -> 1x1 -> sigmoid; returns (b, c, 1, 1)."""
    b, c, h, w = x.shape
    logits = F.conv2d(x, p("to_k.weight"), p("to_k.bias")).reshape(b, 1, h * w)
    pooled = torch.einsum("bin,bcn->bci", logits.softmax(dim=-1), x.reshape(b, c, h * w)).unsqueeze(-1)
    hid = F.silu(F.conv2d(pooled, p("net.0.weight"), p("net.0.bias")))
    return torch.sigmoid(F.conv2d(hid, p("net.2.weight"), p("net.2.bias")))


def resnet_block(p: _SD, x: Tensor, t: Optional[Tensor], cond: Optional[Tensor], heads: int) -> Tensor:
    """ip.py:693-757."""
    scale_shift = None
    if p.has("time_mlp.1.weight") and t is not None:
        ss = F.linear(F.silu(t), p("time_mlp.1.weight"), p("time_mlp.1.bias"))
        half = ss.shape[1] // 2
        scale_shift = (ss[:, :half, None, None], ss[:, half:, None, None])
    h = block(p.sub("block1"), x)
    if p.has("cross_attn.to_q.weight"):
        assert cond is not None
        b, c, hh, ww = h.shape
        tok = h.permute(0, 2, 3, 1).reshape(b, hh * ww, c)
        tok = cross_attention(p.sub("cross_attn"), tok, cond, heads) + tok
        h = tok.reshape(b, hh, ww, c).permute(0, 3, 1, 2)
    h = block(p.sub("block2"), h, scale_shift)
    if p.has("gca.to_k.weight"):
        h = h * global_context_gate(p.sub("gca"), h)
    res = F.conv2d(x, p("res_conv.weight"), p("res_conv.bias")) if p.has("res_conv.weight") else x
    return h + res


def transformer_block(p: _SD, x: Tensor, context: Optional[Tensor], heads: int, depth: int) -> Tensor:
    """ip.py:992-1022 — NCHW -> tokens; depth x (attention + residual, feed-forward + residual)."""
    b, c, hh, ww = x.shape
    tok = x.permute(0, 2, 3, 1).reshape(b, hh * ww, c)
    for d in range(depth):
        lp = p.sub(f"layers.{d}")
        tok = self_attention(lp.sub("0"), tok, heads, context) + tok
        tok = feed_forward(lp.sub("1"), tok) + tok
    return tok.reshape(b, hh, ww, c).permute(0, 3, 1, 2)


def perceiver_resampler(p: _SD, x: Tensor, heads: int, depth: int = 2) -> Tensor:
    """ip.py:379-498 — learned latents (+ latents from the mean-pooled sequence) cross-attend to cat(x+pos, latents)."""
    b, n, d = x.shape
    x_pos = x + p("pos_emb.weight")[:n]
    latents = p("latents").expand(b, -1, -1)
    if p.has("to_latents_from_mean_pooled_seq.1.weight"):
        pooled = x.mean(dim=1)  # masked_mean with an all-true mask (ip.py:490)
        pooled = gain_layernorm(pooled, p("to_latents_from_mean_pooled_seq.0.g"))
        extra = F.linear(pooled, p("to_latents_from_mean_pooled_seq.1.weight"), p("to_latents_from_mean_pooled_seq.1.bias"))
        latents = torch.cat((extra.reshape(b, -1, d), latents), dim=1)
    for i in range(depth):
        ap, fp = p.sub(f"layers.{i}.0"), p.sub(f"layers.{i}.1")
        xn = affine_layernorm(x_pos, ap("norm.weight"), ap("norm.bias"))
        ln = affine_layernorm(latents, ap("norm_latents.weight"), ap("norm_latents.bias"))
        q = F.linear(ln, ap("to_q.weight"))
        kv = F.linear(torch.cat((xn, ln), dim=1), ap("to_kv.weight"))
        inner = q.shape[-1]
        dh = ap("q_scale").shape[0]
        heads = inner // dh
        split = lambda t: t.reshape(b, t.shape[1], heads, dh).permute(0, 2, 1, 3)
        o = cosine_attend(split(q), split(kv[..., :inner]), split(kv[..., inner:]), ap("q_scale"), ap("k_scale"))
        o = o.permute(0, 2, 1, 3).reshape(b, -1, inner)
        o = F.linear(o, ap("to_out.0.weight"))
        o = affine_layernorm(o, ap("to_out.1.weight"), ap("to_out.1.bias"))
        latents = o + latents
        latents = feed_forward(fp, latents) + latents
    return latents


def sinusoidal_features(p_weights: Tensor, x: Tensor) -> Tensor:
    """ip.py:654-669 — [x, sin(2*pi*x*w), cos(2*pi*x*w)]."""
    x = x.reshape(-1, 1)
    freqs = x * p_weights.reshape(1, -1) * 2 * math.pi
    return torch.cat((x, freqs.sin(), freqs.cos()), dim=-1)


def time_conditioning(sd: _SD, prefix: str, values: Tensor, cond_dim: int):
    """ip.py:1573-1578 / 1583-1586 — hiddens -> (t, tokens)."""
    hid = sinusoidal_features(sd(f"to_{prefix}time_hiddens.0.weights"), values)
    hid = F.silu(F.linear(hid, sd(f"to_{prefix}time_hiddens.1.weight"), sd(f"to_{prefix}time_hiddens.1.bias")))
    tokens = F.linear(hid, sd(f"to_{prefix}time_tokens.0.weight"), sd(f"to_{prefix}time_tokens.0.bias"))
    t = F.linear(hid, sd(f"to_{prefix}time_cond.0.weight"), sd(f"to_{prefix}time_cond.0.bias"))
    return t, tokens.reshape(values.shape[0], -1, cond_dim)


def pixel_unshuffle_conv(p: _SD, x: Tensor) -> Tensor:
    """ip.py:633-640 — 'b c (h s1) (w s2) -> b (c s1 s2) h w' then 1x1 conv."""
    return F.conv2d(F.pixel_unshuffle(x, 2), p("1.weight"), p("1.bias"))


def pixel_shuffle_up(p: _SD, x: Tensor) -> Tensor:
    """ip.py:603-631 — 1x1 conv to 4*C_out, SiLU, PixelShuffle(2)."""
    return F.pixel_shuffle(F.silu(F.conv2d(x, p("net.0.weight"), p("net.0.bias"))), 2)


# ---------------------------------------------------------------- full network

def unet_forward(
    sd: Dict[str, Tensor],
    kwargs: dict,
    x: Tensor,
    time: Tensor,
    *,
    lowres_cond_img: Optional[Tensor] = None,
    lowres_noise_times: Optional[Tensor] = None,
    text_embeds: Optional[Tensor] = None,
    text_mask: Optional[Tensor] = None,
    cond_images: Optional[Tensor] = None,
    self_cond: Optional[Tensor] = None,
    cond_drop_prob: float = 0.0,
    taps: Optional[dict] = None,
) -> Tensor:
    """ip.py:1524-1725.  `time` / `lowres_noise_times` are log-SNR values (ip.py:2074, 2081).

    `taps`, if given, is filled with named intermediate tensors (NCHW) for per-stage parity checks.
    """
    cfg = resolve_config(kwargs)
    p = _SD(sd)
    heads = cfg["attn_heads"]
    b = x.shape[0]
    assert cond_drop_prob in (0.0, 1.0), "oracle: sampling only uses deterministic keep masks (ip.py:201-207)"

    def tap(name, val):
        if taps is not None:
            taps[name] = val.detach().clone()

    if cfg["self_cond"]:    # ip.py:1541-1543: the previous step's x0 estimate (zeros when there is none yet), right behind x
        x = torch.cat((x, torch.zeros_like(x) if self_cond is None else self_cond), dim=1)
    if cfg["lowres_cond"]:
        assert lowres_cond_img is not None and lowres_noise_times is not None  # ip.py:1547-1548
    if lowres_cond_img is not None:
        x = torch.cat((x, lowres_cond_img), dim=1)
    # conditioning image: nearest-resized to the input's size, IN FRONT of the other channels, not normalised (ip.py:1555-1560)
    assert (cfg["cond_images_channels"] > 0) == (cond_images is not None)
    if cond_images is not None:
        assert cond_images.shape[1] == cfg["cond_images_channels"]
        if cond_images.shape[-1] != x.shape[-1]:
            cond_images = F.interpolate(cond_images, x.shape[-1], mode=kwargs.get("resize_mode", "nearest"))   # ip.py:1559
        x = torch.cat((cond_images, x), dim=1)

    # initial convolution (ip.py:1564, 1051-1076 / 1198)
    if cfg["init_cross_embed"]:
        fmaps = []
        for i, ksz in enumerate(sorted(cfg["init_cross_embed_kernel_sizes"])):
            fmaps.append(F.conv2d(x, p(f"init_conv.convs.{i}.weight"), p(f"init_conv.convs.{i}.bias"), padding=(ksz - 1) // 2))
        x = torch.cat(fmaps, dim=1)
    else:
        x = F.conv2d(x, p("init_conv.weight"), p("init_conv.bias"), padding=cfg["init_conv_kernel_size"] // 2)
    tap("init_conv", x)
    init_conv_residual = x.clone() if cfg["init_conv_to_final_conv_residual"] else None   # ip.py:1568-1569

    # time conditioning (ip.py:1573-1589)
    t, time_tokens = time_conditioning(p, "", time, cfg["cond_dim"])
    if cfg["lowres_cond"]:
        lt, ltok = time_conditioning(p, "lowres_", lowres_noise_times, cfg["cond_dim"])
        t = t + lt
        time_tokens = torch.cat((time_tokens, ltok), dim=1)

    # text conditioning (ip.py:1593-1652)
    text_tokens = None
    if text_embeds is not None and cfg["cond_on_text"]:
        keep = cond_drop_prob == 0.0
        L = cfg["max_text_len"]
        tok = F.linear(text_embeds, p("text_to_cond.weight"), p("text_to_cond.bias"))[:, :L]
        n_tok = tok.shape[1]
        if n_tok < L:
            tok = F.pad(tok, (0, 0, 0, L - n_tok))
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
        pooled = tok.mean(dim=1)
        q = p.sub("to_text_non_attn_cond")
        hid = affine_layernorm(pooled, q("0.weight"), q("0.bias"))
        hid = F.silu(F.linear(hid, q("1.weight"), q("1.bias")))
        hid = F.linear(hid, q("3.weight"), q("3.bias"))
        if not keep:
            hid = p("null_text_hidden").expand(b, -1)
        t = t + hid

    c = time_tokens if text_tokens is None else torch.cat((time_tokens, text_tokens), dim=1)
    c = affine_layernorm(c, p("norm_cond.weight"), p("norm_cond.bias"))
    tap("t", t)
    tap("c", c)

    if cfg["memory_efficient"]:
        x = resnet_block(p.sub("init_resnet_block"), x, t, None, heads)

    hiddens = []
    n_levels = len(cfg["in_out"])
    for i in range(n_levels):
        lp = p.sub(f"downs.{i}")
        if cfg["memory_efficient"]:
            x = pixel_unshuffle_conv(lp.sub("0"), x)
        x = resnet_block(lp.sub("1"), x, t, c if cfg["layer_cross_attns_t"][i] else None, heads)
        for j in range(cfg["num_resnet_blocks_t"][i]):
            x = resnet_block(lp.sub(f"2.{j}"), x, t, None, heads)
            hiddens.append(x)
        if cfg["layer_attns_t"][i]:
            x = transformer_block(lp.sub("3"), x, c, heads, cfg["layer_attns_depth_t"][i])
        hiddens.append(x)
        tap(f"down{i}", x)
        if not cfg["memory_efficient"]:
            if i < n_levels - 1:
                x = pixel_unshuffle_conv(lp.sub("4"), x)
            else:  # ip.py:1366 — Parallel(3x3, 1x1), summed
                x = (F.conv2d(x, lp("4.fns.0.weight"), lp("4.fns.0.bias"), padding=1)
                     + F.conv2d(x, lp("4.fns.1.weight"), lp("4.fns.1.bias")))

    x = resnet_block(p.sub("mid_block1"), x, t, c, heads)
    if cfg["attend_at_middle"]:
        x = transformer_block(p.sub("mid_attn"), x, None, heads, cfg["layer_mid_attns_depth"])
    x = resnet_block(p.sub("mid_block2"), x, t, c, heads)
    tap("mid", x)

    s = cfg["skip_scale"]
    up_hiddens = []
    for i in range(n_levels):
        lvl = n_levels - 1 - i
        lp = p.sub(f"ups.{i}")
        x = torch.cat((x, hiddens.pop() * s), dim=1)
        x = resnet_block(lp.sub("0"), x, t, c if cfg["layer_cross_attns_t"][lvl] else None, heads)
        for j in range(cfg["num_resnet_blocks_t"][lvl]):
            x = torch.cat((x, hiddens.pop() * s), dim=1)
            x = resnet_block(lp.sub(f"1.{j}"), x, t, None, heads)
        if cfg["layer_attns_t"][lvl]:
            x = transformer_block(lp.sub("2"), x, c, heads, cfg["layer_attns_depth_t"][lvl])
        up_hiddens.append(x)                                                              # ip.py:1707
        if i < n_levels - 1 or cfg["memory_efficient"]:
            if cfg["pixel_shuffle_upsample"]:
                x = pixel_shuffle_up(lp.sub("3"), x)
            else:   # ip.py:595-601 — nearest x2, then a 3x3 conv
                x = F.conv2d(F.interpolate(x, scale_factor=2, mode="nearest"), lp("3.1.weight"), lp("3.1.bias"), padding=1)
        tap(f"up{i}", x)

    if cfg["combine_upsample_fmaps"]:       # UpsampleCombiner (ip.py:1078-1110, 1712): every up level's feature map, resized
        size = x.shape[-1]                                # to the output resolution (nearest) and passed through its own Block
        outs = [block(p.sub(f"upsample_combiner.fmap_convs.{i}"), f if f.shape[-1] == size else F.interpolate(f, size, mode="nearest"))
                for i, f in enumerate(up_hiddens)]
        x = torch.cat((x, *outs), dim=1)
    if init_conv_residual is not None:                                                    # ip.py:1716-1717
        x = torch.cat((x, init_conv_residual), dim=1)
    if cfg["final_resnet_block"]:
        x = resnet_block(p.sub("final_res_block"), x, t, None, heads)
    tap("final_res", x)
    if lowres_cond_img is not None:
        x = torch.cat((x, lowres_cond_img), dim=1)
    return F.conv2d(x, p("final_conv.weight"), p("final_conv.bias"), padding=cfg["final_conv_kernel_size"] // 2)


def unet_forward_with_cond_scale(sd, kwargs, x, time, *, cond_scale: float = 1.0, **kw) -> Tensor:
    """ip.py:1510-1522 — classifier-free guidance: null + (cond - null) * cond_scale."""
    logits = unet_forward(sd, kwargs, x, time, **kw)
    if cond_scale == 1:
        return logits
    null_logits = unet_forward(sd, kwargs, x, time, **{**kw, "cond_drop_prob": 1.0})
    return null_logits + (logits - null_logits) * cond_scale
