"""Parameter containers for the drop-in `Unet`.

These nn.Modules own the learnable tensors under exactly the attribute paths the reference module tree
uses (so `state_dict()` keys / shapes are interchangeable with lucidrains/imagen-pytorch)
and initialise them with the same distributions — but they carry NO forward computation: the arithmetic
of every one of them is executed by the HIP kernels planned in engine.py.  Reference lines are cited per
container so the key layout can be checked against the source.
"""
from __future__ import annotations

import torch
from torch import nn


class Holder(nn.Module):
    """A module that only owns parameters / sub-holders.  Calling it is an error by design."""

    def forward(self, *a, **k):  # pragma: no cover
        raise RuntimeError("parameter container: computation runs in the HIP engine (imagen_pytorch_amd.engine)")


def _ones(*shape):
    return nn.Parameter(torch.ones(*shape))


class GainNorm(Holder):
    """ip.py:331-349 `LayerNorm` (gain `g` only)."""

    def __init__(self, dim):
        super().__init__()
        self.g = _ones(dim)


class ChanRMSNormP(Holder):
    """ip.py:322-329: `gamma` of shape (dim, 1, 1)."""

    def __init__(self, dim):
        super().__init__()
        self.gamma = _ones(dim, 1, 1)


class SinuPosEmbP(Holder):
    """ip.py:654-669: `weights` ~ N(0, 1) of length dim/2."""

    def __init__(self, dim):
        super().__init__()
        assert dim % 2 == 0
        self.weights = nn.Parameter(torch.randn(dim // 2))


class BlockP(Holder):
    """ip.py:671-691: norm.gamma + project (3x3 conv)."""

    def __init__(self, dim, dim_out):
        super().__init__()
        self.norm = ChanRMSNormP(dim)
        self.project = nn.Conv2d(dim, dim_out, 3, padding=1)


class UpsampleCombinerP(Holder):
    """ip.py:1078-1110: one Block(dim_in -> dim) per up level when enabled (no parameters otherwise)."""

    def __init__(self, dim, *, enabled=False, dim_ins=(), dim_outs=()):
        super().__init__()
        dim_outs = tuple(dim_outs) if isinstance(dim_outs, (tuple, list)) else (dim_outs,) * len(dim_ins)
        assert len(dim_ins) == len(dim_outs)
        self.enabled = enabled
        if not enabled:
            self.dim_out = dim
            return
        self.fmap_convs = nn.ModuleList([BlockP(di, do) for di, do in zip(dim_ins, dim_outs)])
        self.dim_out = dim + sum(dim_outs)


class CrossAttentionP(Holder):
    """ip.py:759-791."""

    def __init__(self, dim, context_dim, dim_head=64, heads=8):
        super().__init__()
        inner = dim_head * heads
        self.heads, self.dim_head = heads, dim_head
        self.norm = GainNorm(dim)
        self.null_kv = nn.Parameter(torch.randn(2, dim_head))
        self.to_q = nn.Linear(dim, inner, bias=False)
        self.to_kv = nn.Linear(context_dim, inner * 2, bias=False)
        self.q_scale = _ones(dim_head)
        self.k_scale = _ones(dim_head)
        self.to_out = nn.Sequential(nn.Linear(inner, dim, bias=False), GainNorm(dim))


class AttentionP(Holder):
    """ip.py:502-532 (multi-query: to_kv produces ONE head)."""

    def __init__(self, dim, dim_head=64, heads=8, context_dim=None):
        super().__init__()
        inner = dim_head * heads
        self.heads, self.dim_head = heads, dim_head
        self.norm = GainNorm(dim)
        self.null_kv = nn.Parameter(torch.randn(2, dim_head))
        self.to_q = nn.Linear(dim, inner, bias=False)
        self.to_kv = nn.Linear(dim, dim_head * 2, bias=False)
        self.q_scale = _ones(dim_head)
        self.k_scale = _ones(dim_head)
        self.to_context = (nn.Sequential(nn.LayerNorm(context_dim), nn.Linear(context_dim, dim_head * 2))
                           if context_dim is not None else None)
        self.to_out = nn.Sequential(nn.Linear(inner, dim, bias=False), GainNorm(dim))


def feed_forward_p(dim, mult=2.0):
    """ip.py:972-980: indices 0 (norm), 1 (linear), 3 (norm), 4 (linear) carry parameters."""
    hidden = int(dim * mult)
    return nn.Sequential(GainNorm(dim), nn.Linear(dim, hidden, bias=False), nn.GELU(), GainNorm(hidden),
                         nn.Linear(hidden, dim, bias=False))


class TransformerBlockP(Holder):
    """ip.py:992-1010: layers[d] = [Attention, FeedForward]."""

    def __init__(self, dim, depth=1, heads=8, dim_head=32, ff_mult=2, context_dim=None):
        super().__init__()
        self.layers = nn.ModuleList([
            nn.ModuleList([AttentionP(dim, dim_head=dim_head, heads=heads, context_dim=context_dim), feed_forward_p(dim, ff_mult)])
            for _ in range(depth)
        ])


class GlobalContextP(Holder):
    """ip.py:945-963."""

    def __init__(self, dim_in, dim_out):
        super().__init__()
        self.to_k = nn.Conv2d(dim_in, 1, 1)
        hidden = max(3, dim_out // 2)
        self.net = nn.Sequential(nn.Conv2d(dim_in, hidden, 1), nn.SiLU(), nn.Conv2d(hidden, dim_out, 1), nn.Sigmoid())


class ResnetBlockP(Holder):
    """ip.py:693-732."""

    def __init__(self, dim, dim_out, *, cond_dim=None, time_cond_dim=None, use_gca=False, heads=8, dim_head=64):
        super().__init__()
        self.dim, self.dim_out = dim, dim_out
        self.time_mlp = nn.Sequential(nn.SiLU(), nn.Linear(time_cond_dim, dim_out * 2)) if time_cond_dim is not None else None
        self.cross_attn = CrossAttentionP(dim_out, cond_dim, dim_head=dim_head, heads=heads) if cond_dim is not None else None
        self.block1 = BlockP(dim, dim_out)
        self.block2 = BlockP(dim_out, dim_out)
        self.gca = GlobalContextP(dim_out, dim_out) if use_gca else None
        self.res_conv = nn.Conv2d(dim, dim_out, 1) if dim != dim_out else None


class PerceiverAttentionP(Holder):
    """ip.py:379-406."""

    def __init__(self, dim, dim_head=64, heads=8):
        super().__init__()
        inner = dim_head * heads
        self.heads, self.dim_head = heads, dim_head
        self.norm = nn.LayerNorm(dim)
        self.norm_latents = nn.LayerNorm(dim)
        self.to_q = nn.Linear(dim, inner, bias=False)
        self.to_kv = nn.Linear(dim, inner * 2, bias=False)
        self.q_scale = _ones(dim_head)
        self.k_scale = _ones(dim_head)
        self.to_out = nn.Sequential(nn.Linear(inner, dim, bias=False), nn.LayerNorm(dim))


class PerceiverResamplerP(Holder):
    """ip.py:447-479."""

    def __init__(self, dim, depth, dim_head=64, heads=8, num_latents=64, num_latents_mean_pooled=4, max_seq_len=512, ff_mult=4):
        super().__init__()
        self.pos_emb = nn.Embedding(max_seq_len, dim)
        self.latents = nn.Parameter(torch.randn(num_latents, dim))
        self.num_latents_mean_pooled = num_latents_mean_pooled
        self.to_latents_from_mean_pooled_seq = None
        if num_latents_mean_pooled > 0:
            self.to_latents_from_mean_pooled_seq = nn.Sequential(GainNorm(dim), nn.Linear(dim, dim * num_latents_mean_pooled), nn.Identity())
        self.layers = nn.ModuleList([
            nn.ModuleList([PerceiverAttentionP(dim, dim_head=dim_head, heads=heads), feed_forward_p(dim, ff_mult)]) for _ in range(depth)
        ])


class CrossEmbedP(Holder):
    """ip.py:1051-1072: `convs` at sorted kernel sizes; channel split dim_out/2, /4, ..., remainder."""

    def __init__(self, dim_in, kernel_sizes, dim_out=None, stride=2):
        super().__init__()
        dim_out = dim_out or dim_in
        kernel_sizes = sorted(kernel_sizes)
        n = len(kernel_sizes)
        scales = [int(dim_out / (2 ** i)) for i in range(1, n)]
        scales = [*scales, dim_out - sum(scales)]
        self.kernel_sizes, self.dim_scales, self.stride = kernel_sizes, scales, stride
        self.convs = nn.ModuleList([nn.Conv2d(dim_in, ds, k, stride=stride, padding=(k - stride) // 2) for k, ds in zip(kernel_sizes, scales)])


def downsample_p(dim, dim_out):
    """ip.py:633-640: Sequential(Rearrange, Conv2d(4*dim -> dim_out, 1)); the conv sits at index 1."""
    return nn.Sequential(nn.Identity(), nn.Conv2d(dim * 4, dim_out, 1))


class ParallelP(Holder):
    """ip.py:368-375 / 1366: `fns` = [Conv2d 3x3, Conv2d 1x1], outputs summed."""

    def __init__(self, dim_in, dim_out):
        super().__init__()
        self.fns = nn.ModuleList([nn.Conv2d(dim_in, dim_out, 3, padding=1), nn.Conv2d(dim_in, dim_out, 1)])


class PixelShuffleUpsampleP(Holder):
    """ip.py:603-631: net = [Conv2d(dim -> 4*dim_out, 1), SiLU, PixelShuffle(2)], ICNR-style init (kaiming on o/4, repeated x4)."""

    def __init__(self, dim, dim_out=None):
        super().__init__()
        dim_out = dim_out or dim
        conv = nn.Conv2d(dim, dim_out * 4, 1)
        self.net = nn.Sequential(conv, nn.SiLU(), nn.PixelShuffle(2))
        o, i, h, w = conv.weight.shape
        base = torch.empty(o // 4, i, h, w)
        nn.init.kaiming_uniform_(base)
        with torch.no_grad():
            conv.weight.copy_(base.repeat_interleave(4, dim=0))
            conv.bias.zero_()


def upsample_conv_p(dim, dim_out=None):
    """ip.py:595-601 `Upsample`: Sequential(nn.Upsample(x2, nearest), Conv2d(dim, dim_out, 3, padding = 1)); the conv sits at index 1."""
    return nn.Sequential(nn.Upsample(scale_factor=2, mode='nearest'), nn.Conv2d(dim, dim_out or dim, 3, padding=1))
