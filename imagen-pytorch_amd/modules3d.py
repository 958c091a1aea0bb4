"""Parameter containers for the drop-in `Unet3D` (Imagen-Video denoiser, iv.py = imagen_pytorch/imagen_video.py).

Same rule as modules.py: these nn.Modules own the learnable tensors under exactly the attribute paths (and with the shapes and
initialisations) of the reference module tree, so `state_dict()` is interchangeable — and carry no computation.  The reference
expresses every per-frame 2-D convolution as an `nn.Conv3d` with a (1, k, k) kernel (iv.py:574-588), hence the 5-D weights.
"""
from __future__ import annotations

import torch
from torch import nn

from .modules import GainNorm, Holder, _ones, feed_forward_p  # noqa: F401  (feed_forward_p: the Perceiver's token feed-forward)


def conv_frames_p(dim_in, dim_out, kernel, stride=1, padding=0, bias=True):
    """iv.py:574-588 `Conv2d(...)`: nn.Conv3d with kernel (1, k, k)."""
    return nn.Conv3d(dim_in, dim_out, (1, kernel, kernel), stride=(1, stride, stride), padding=(0, padding, padding), bias=bias)


class ChanRMSNorm3dP(Holder):
    """iv.py:207-214: `gamma` of shape (dim, 1, 1, 1)."""

    def __init__(self, dim):
        super().__init__()
        self.gamma = _ones(dim, 1, 1, 1)


class ChanGainNorm3dP(Holder):
    """iv.py:216-227 `ChanLayerNorm`: `g` of shape (1, dim, 1, 1, 1)."""

    def __init__(self, dim):
        super().__init__()
        self.g = _ones(1, dim, 1, 1, 1)


class PseudoConv3dP(Holder):
    """iv.py:397-417 `Conv3d`: spatial_conv (k x k per frame) + temporal_conv (Conv1d over frames, identity-initialised)."""

    def __init__(self, dim, dim_out, kernel_size=3):
        super().__init__()
        self.spatial_conv = nn.Conv2d(dim, dim_out, kernel_size, padding=kernel_size // 2)
        self.temporal_conv = nn.Conv1d(dim_out, dim_out, kernel_size) if kernel_size > 1 else None
        if self.temporal_conv is not None:
            nn.init.dirac_(self.temporal_conv.weight.data)
            nn.init.zeros_(self.temporal_conv.bias.data)


class Block3dP(Holder):
    """iv.py:716-726."""

    def __init__(self, dim, dim_out):
        super().__init__()
        self.norm = ChanRMSNorm3dP(dim)
        self.project = PseudoConv3dP(dim, dim_out, 3)


class CrossAttention3dP(Holder):
    """iv.py:817-849 (same tensors as the image Unet's CrossAttention)."""

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


class GlobalContext3dP(Holder):
    """iv.py:1002-1020."""

    def __init__(self, dim_in, dim_out):
        super().__init__()
        self.to_k = conv_frames_p(dim_in, 1, 1)
        hidden = max(3, dim_out // 2)
        self.net = nn.Sequential(conv_frames_p(dim_in, hidden, 1), nn.SiLU(), conv_frames_p(hidden, dim_out, 1), nn.Sigmoid())


class ResnetBlock3dP(Holder):
    """iv.py:743-783."""

    def __init__(self, dim, dim_out, *, cond_dim=None, time_cond_dim=None, use_gca=False, heads=8, dim_head=64):
        super().__init__()
        self.dim, self.dim_out = dim, dim_out
        self.time_mlp = nn.Sequential(nn.SiLU(), nn.Linear(time_cond_dim, dim_out * 2)) if time_cond_dim is not None else None
        self.cross_attn = CrossAttention3dP(dim_out, cond_dim, dim_head=dim_head, heads=heads) if cond_dim is not None else None
        self.block1 = Block3dP(dim, dim_out)
        self.block2 = Block3dP(dim_out, dim_out)
        self.gca = GlobalContext3dP(dim_out, dim_out) if use_gca else None
        self.res_conv = conv_frames_p(dim, dim_out, 1) if dim != dim_out else None


class DynamicPositionBiasP(Holder):
    """iv.py:1182-1206: mlp = [Seq(Linear(1, dim), LayerNorm, SiLU), (depth-1) x Seq(Linear(dim, dim), LayerNorm, SiLU), Linear(dim, heads)]."""

    def __init__(self, dim, heads, depth):
        super().__init__()
        layers = [nn.Sequential(nn.Linear(1, dim), GainNorm(dim), nn.SiLU())]
        for _ in range(max(depth - 1, 0)):
            layers.append(nn.Sequential(nn.Linear(dim, dim), GainNorm(dim), nn.SiLU()))
        layers.append(nn.Linear(dim, heads))
        self.mlp = nn.ModuleList(layers)


class Attention3dP(Holder):
    """iv.py:455-497: the image Unet's multi-query attention + `null_attn_bias` and the optional relative position bias.
    Parameter registration order follows the reference constructor (it fixes the state_dict key order)."""

    def __init__(self, dim, dim_head=64, heads=8, causal=False, context_dim=None, rel_pos_bias=False, rel_pos_bias_mlp_depth=2,
                 init_zero=False):
        super().__init__()
        inner = dim_head * heads
        self.heads, self.dim_head, self.causal = heads, dim_head, causal
        self.rel_pos_bias = DynamicPositionBiasP(dim, heads, rel_pos_bias_mlp_depth) if rel_pos_bias else None
        self.norm = GainNorm(dim)
        self.null_attn_bias = nn.Parameter(torch.randn(heads))
        self.null_kv = nn.Parameter(torch.randn(2, dim_head))
        self.to_q = nn.Linear(dim, inner, bias=False)
        self.to_kv = nn.Linear(dim, dim_head * 2, bias=False)
        self.q_scale = _ones(dim_head)
        self.k_scale = _ones(dim_head)
        self.to_context = (nn.Sequential(nn.LayerNorm(context_dim), nn.Linear(context_dim, dim_head * 2))
                           if context_dim is not None else None)
        self.to_out = nn.Sequential(nn.Linear(inner, dim, bias=False), GainNorm(dim))
        if init_zero:
            nn.init.zeros_(self.to_out[-1].g)


class ResidualP(Holder):
    """iv.py:238-244: `fn`."""

    def __init__(self, fn):
        super().__init__()
        self.fn = fn


def temporal_peg_p(dim):
    """iv.py:1413-1414: Residual(Sequential(Pad, Conv3d(dim, dim, (3, 1, 1), groups = dim))) -> keys `fn.1.weight/bias`."""
    return ResidualP(nn.Sequential(nn.Identity(), nn.Conv3d(dim, dim, (3, 1, 1), groups=dim)))


def temporal_attn_p(dim, heads, dim_head, causal, rel_pos_bias_depth):
    """iv.py:1416: RearrangeTimeCentric(Residual(Attention(causal, init_zero, rel_pos_bias))) -> keys `fn.fn.*`."""
    return ResidualP(ResidualP(Attention3dP(dim, dim_head=dim_head, heads=heads, causal=causal, rel_pos_bias=True,
                                             rel_pos_bias_mlp_depth=rel_pos_bias_depth, init_zero=True)))


def chan_feed_forward_p(dim, mult=2, time_token_shift=True):
    """iv.py:1048-1057: [ChanLayerNorm, Conv(1x1x1, no bias), GELU, (TimeTokenShift), ChanLayerNorm, Conv(1x1x1, no bias)];
    `Sequential` drops the shift slot when it is disabled, which moves the last two indices down by one."""
    hidden = int(dim * mult)
    mods = [ChanGainNorm3dP(dim), conv_frames_p(dim, hidden, 1, bias=False), nn.GELU()]
    if time_token_shift:
        mods.append(nn.Identity())
    mods += [ChanGainNorm3dP(hidden), conv_frames_p(hidden, dim, 1, bias=False)]
    seq = nn.Sequential(*mods)
    seq.time_token_shift = time_token_shift
    return seq


class TransformerBlock3dP(Holder):
    """iv.py:1059-1078: layers[d] = [Attention, ChanFeedForward]."""

    def __init__(self, dim, depth=1, heads=8, dim_head=32, ff_mult=2, ff_time_token_shift=True, context_dim=None):
        super().__init__()
        self.layers = nn.ModuleList([
            nn.ModuleList([Attention3dP(dim, dim_head=dim_head, heads=heads, context_dim=context_dim),
                           chan_feed_forward_p(dim, ff_mult, ff_time_token_shift)])
            for _ in range(depth)
        ])


class CrossEmbed3dP(Holder):
    """iv.py:1121-1142."""

    def __init__(self, dim_in, kernel_sizes, dim_out=None, stride=2):
        super().__init__()
        dim_out = dim_out or dim_in
        kernel_sizes = sorted(kernel_sizes)
        n = len(kernel_sizes)
        scales = [int(dim_out / (2 ** i)) for i in range(1, n)]
        scales = [*scales, dim_out - sum(scales)]
        self.kernel_sizes, self.dim_scales, self.stride = kernel_sizes, scales, stride
        self.convs = nn.ModuleList([conv_frames_p(dim_in, ds, k, stride=stride, padding=(k - stride) // 2) for k, ds in zip(kernel_sizes, scales)])


def downsample3d_p(dim, dim_out):
    """iv.py:640-645: Sequential(Rearrange, Conv(4*dim -> dim_out, 1x1)); conv at index 1."""
    return nn.Sequential(nn.Identity(), conv_frames_p(dim * 4, dim_out, 1))


def temporal_downsample_p(dim, dim_out=None, stride=2):
    """iv.py:681-686: Sequential(Rearrange, Conv(dim*stride -> dim_out, 1x1)); conv at index 1."""
    seq = nn.Sequential(nn.Identity(), conv_frames_p(dim * stride, dim_out or dim, 1))
    seq.stride = stride
    return seq


class Parallel3dP(Holder):
    """iv.py:246-253 / 1465."""

    def __init__(self, dim_in, dim_out):
        super().__init__()
        self.fns = nn.ModuleList([conv_frames_p(dim_in, dim_out, 3, padding=1), conv_frames_p(dim_in, dim_out, 1)])


class PixelShuffleUpsample3dP(Holder):
    """iv.py:609-631: net = [Conv(dim -> 4*dim_out, 1x1), SiLU], kaiming on o/4 repeated x4, zero bias."""

    def __init__(self, dim, dim_out=None):
        super().__init__()
        dim_out = dim_out or dim
        conv = conv_frames_p(dim, dim_out * 4, 1)
        self.net = nn.Sequential(conv, nn.SiLU())
        base = torch.empty(conv.weight.shape[0] // 4, *conv.weight.shape[1:])
        nn.init.kaiming_uniform_(base)
        with torch.no_grad():
            conv.weight.copy_(base.repeat_interleave(4, dim=0))
            conv.bias.zero_()


class TemporalPixelShuffleUpsampleP(Holder):
    """iv.py:649-672: net = [Conv1d(dim -> dim_out*stride, 1), SiLU]."""

    def __init__(self, dim, dim_out=None, stride=2):
        super().__init__()
        self.stride = stride
        dim_out = dim_out or dim
        conv = nn.Conv1d(dim, dim_out * stride, 1)
        self.net = nn.Sequential(conv, nn.SiLU())
        base = torch.empty(conv.weight.shape[0] // stride, *conv.weight.shape[1:])
        nn.init.kaiming_uniform_(base)
        with torch.no_grad():
            conv.weight.copy_(base.repeat_interleave(stride, dim=0))
            conv.bias.zero_()
