"""UnetEngine — the host-side planner that turns a drop-in `Unet` (parameters only) into a static list of
HIP kernel launches for a fixed (rows, image_size): the MI355X replacement of `Unet.forward`
(ip.py:1524-1725).  Three plans are built:

  * static plan   — everything that does not change over the T timesteps of a sample() call: text projection,
                    null-embedding select, PerceiverResampler, text hiddens, low-res noise-level embedding,
                    norm_cond of the static tokens and their K/V rows for every attention block
                    (results-preserving because norm_cond / to_kv are per-token);
  * step plan     — one denoiser evaluation: time embedding -> batched time-MLPs -> U-net traversal;
  * (sampler ops are appended by imagen.py to form the per-timestep graph).

Row convention: `rows` = B (plain forward) or 2B (classifier-free guidance: rows [0,B) conditional,
rows [B,2B) with the null text conditioning — ip.py:1510-1522 evaluated as ONE batch).  Row r reads image
`r % src_batch`.

Nothing here computes with torch: torch only owns device buffers.  `dry=True` builds the plan on CPU memory
without ever launching (host-logic tests).
"""
from __future__ import annotations

import math
import os
from typing import Dict, List, Optional

import torch
from torch import nn

from . import ops
from .modules import CrossAttentionP, CrossEmbedP, ParallelP, PixelShuffleUpsampleP, ResnetBlockP, TransformerBlockP
from .ops import ACT_GELU, ACT_SILU, LOG2E, OUT_NCHW_F32, OUT_PIXEL_SHUFFLE, Act, Plan

SIM_SCALE = 8.0  # cosine-sim attention scale (ip.py:510, 768, 386)


def _f32(t, dev):
    return t.detach().float().contiguous().to(dev)


def _f16(t, dev):
    return t.detach().to(torch.float16).contiguous().to(dev)


def _pad_vec(v: torch.Tensor, n: int) -> torch.Tensor:
    out = torch.zeros(n, dtype=torch.float32)
    out[: v.numel()] = v.detach().float().flatten().cpu()
    return out


class Weights:
    """Device-resident packed parameters of one Unet (shared by all engines of that unet on a device)."""

    def __init__(self, unet, device):
        self.dev = device
        self.cache: Dict[str, object] = {}
        self.unet = unet

    def get(self, key, make):
        if key not in self.cache:
            self.cache[key] = make()
        return self.cache[key]

    def conv(self, key, mod: nn.Module, in_scale=None, G=None, cin_pad_to=None, out_perm=None, split=False):
        """split: the split-precision form of the weight (ops.pack_weight) where the layer's input channels allow it."""
        if split and mod.weight.shape[1] % 8 == 0 and cin_pad_to is None:
            key = key + "|split"
        else:
            split = False

        def make():
            w, b = mod.weight.detach().float(), (mod.bias.detach().float() if mod.bias is not None else None)
            if w.ndim == 2:
                w = w[:, :, None, None]
            if out_perm is not None:
                w, b = w[out_perm], (b[out_perm] if b is not None else None)
            if cin_pad_to is not None and cin_pad_to > w.shape[1]:
                wp = torch.zeros(w.shape[0], cin_pad_to, *w.shape[2:])
                wp[:, : w.shape[1]] = w
                w = wp
            sc = None
            if in_scale is not None:
                sc = torch.ones(w.shape[1])
                sc[: in_scale.numel()] = in_scale
            return ops.pack_weight(w, b, self.dev, in_scale=sc, G=G, split=split)
        return self.get(key, make)

    def raw(self, key, w, b, G=None, split=False):
        split = bool(split and w.shape[1] % 8 == 0)
        return self.get(key + ("|split" if split else ""), lambda: ops.pack_weight(w, b, self.dev, G=G, split=split))

    def f32(self, key, t_fn):
        return self.get(key, lambda: _f32(t_fn(), self.dev))

    def f16(self, key, t_fn):
        return self.get(key, lambda: _f16(t_fn(), self.dev))


def _fingerprint(unet) -> tuple:
    return tuple(p._version for p in unet.parameters()) + tuple(p.data_ptr() for p in unet.parameters())


# ACT_PREP + all-DMA conv for the Blocks with at least this many output channels (the MFMA-bound layers: the prologue pass costs one
# read + one write of the input, the conv kernel drops its staging instruction stream); 0 = never
INIT_CONV_SHARED_MIN_PIXELS = 1 << 17   # ... of stages with at least this many distinct pixels (below, the two copies cost what the half launch saves)
INIT_CONV_SHARED = 1   # (module constant; measured in round 3, call X) under CFG the init conv of the large stage runs on the B distinct images only
BIG_PREP = 1   # (module constant; round 3, call S) ACT_PREP + conv_big for the Blocks conv_big applies to (else they keep the fused prologue)
ACT_PREP_MIN_COUT = 0   # (measured in the model: the extra pass costs more than it saves — off)
TAIL_FUSED = 1   # (module constant; measured in call B, profiles/r03_b_tail_ab.jsonl) GCA_FINAL + GATE_RESIDUAL of an identity ResnetBlock as one GCA_TAIL launch
TAIL_ACT = 1     # (module constant; call B) ... which also writes the next block1's activated input
LN_STATS_FUSED = 1   # (module constant; call B) LayerNorm statistics from the producing launch (GCA_TAIL / LN_RESIDUAL) instead of a ROWSTAT pass


# Split-precision weights (ops.pack_weight(split=True): the fp32 weight as two fp16 MFMA operands, the input read twice) where the doubled K
# costs nothing that matters — measured with tools/parity_budget.py (the plan interpreter against the oracle; README unet1 null branch):
#  * SPLIT_STATIC: everything that runs once per request (text projection, Perceiver resampler, text hiddens, the K / V projections of the
#    conditioning tokens) and the timestep-only chain (time embedding -> time conditioning / tokens -> the ResnetBlocks' scale / shift; one
#    batched pass per request in sampler mode, R-row GEMMs otherwise).  The null branch's conditioning is input-independent, so the rounding
#    error of these weights is the SAME vector in every null row: a bias, not noise — 1.11e-3 -> 0.99e-3 on the null rows of README unet1.
#  * SPLIT_SMALL: block1 of the 32-channel ResnetBlocks and the final conv of stages whose launches are latency-bound (<= SPLIT_SMALL_FLOPS
#    executed FLOPs): the layers next to the output, whose weight rounding reaches it unattenuated (0.99e-3 -> 0.94e-3).
TIME_TABLE_MAX_BYTES = int(float(os.environ.get("IMAGEN_TIME_TABLE_MAX_GB", "4")) * (1 << 30))   # per stage and lane (enable_time_table)
TIME_TABLE_CHUNK_BYTES = 256 << 20   # intermediates of the batched all-steps pass: it runs in chunks of steps that need at most this much
SPLIT_STATIC = 1
SPLIT_SMALL = 1
SPLIT_SMALL_MAX_K = 320        # taps * input channels of the unsplit weight
SPLIT_1X1 = 0      # (measured in round 4, call A: no parity gain worth its launches; a split 1x1 layer also leaves the ROWCHAIN path)
SPLIT_BLOCK2 = 1   # round 5: on (measured -4 % whole-Unet error for +0.03 ms per step pair: the margin under 1e-3)
SPLIT_SMALL_FLOPS = 3.0e9      # 2 * pixels * Cout * (2 K) of the split launch
# The timestep-conditioning chain in fp32 (ops.linear_f32, IMAGEN_OP_LINEAR_F32): to_time_cond and the ResnetBlocks' batched time MLPs act on ONE row per
# sample, and what they produce — the (scale, shift) of every Block — multiplies every pixel of that sample, so the fp16 roundings of those rows (t as
# stored, SiLU(t) as an MFMA operand, the scale / shift rows as stored) are one error vector for the whole map.  Plan interpreter, README unet1, null
# rows of seeds 0 / 1 / 2: 0.957 / 1.012 / 0.91e-3 -> 0.890 / 0.951 / 0.893e-3 (the scale / shift rows alone: 0.904 / 0.981); cond rows unchanged.
# The split weights of these two layers (SPLIT_STATIC) are superseded by the fp32 weights.  0: the fp16 GEMMs of rounds 1-5.
TIME_CHAIN_F32 = 1


class UnetEngine:
    def __init__(self, unet, rows: int, src_batch: int, size: int, device, with_text: bool = True, dry: bool = False):
        assert rows % src_batch == 0
        self.unet, self.R, self.S, self.dev, self.dry = unet, rows, size, torch.device(device), dry
        self.src_batch = src_batch
        self.fp = _fingerprint(unet)
        wkey = (str(self.dev),)
        cache = getattr(unet, "_weight_cache", None)
        if cache is None or cache[0] != self.fp or cache[1] != wkey:
            cache = (self.fp, wkey, Weights(unet, self.dev))
            unet._weight_cache = cache
        self.W: Weights = cache[2]
        self.lc = unet._layer_cfg
        self.cond_dim, self.Tc = unet.cond_dim, unet.time_cond_dim
        self.ntt = unet.num_time_tokens
        self.lowres = unet.lowres_cond
        self.has_text = bool(unet.cond_on_text and with_text)
        self.NTX = 0
        if self.has_text:
            self.NTX = (unet.attn_pool.latents.shape[0] + unet.attn_pool.num_latents_mean_pooled) if unet.attn_pool is not None else unet.max_text_len
        self.NS = (self.ntt if self.lowres else 0) + self.NTX          # static conditioning tokens
        self.NT = self.ntt + self.NS                                    # all conditioning tokens
        self.attn_sites: List[dict] = []   # attention K/V buffers + how to fill their conditioning rows
        self.taps: Dict[str, Act] = {}     # named intermediates (persistent buffers) for per-stage parity checks
        self._static_plans: Dict[int, tuple] = {}   # n_tok -> (plan, te16, mask_u8)
        self._cond_ready = False
        self._tt_plan = None        # sampler mode: the batched all-steps pass of the timestep-only conditioning (enable_time_table)
        self._alloc_io()
        self.step_plan = self._build_step_plan()

    # ------------------------------------------------------------------------------------------ helpers
    def stale(self) -> bool:
        return _fingerprint(self.unet) != self.fp

    def new(self, B, H, W, C, zero=False) -> Act:
        return ops.new_act(B, H, W, C, self.dev, zero=zero)

    def _split_small(self, Cin: int, taps: int, Cout: int, pixels: int) -> bool:
        """A split-precision weight for this launch (SPLIT_SMALL above)?"""
        return bool(SPLIT_SMALL and Cin % 8 == 0 and taps * Cin <= SPLIT_SMALL_MAX_K and 4.0 * pixels * Cout * taps * Cin <= SPLIT_SMALL_FLOPS)

    def f32buf(self, *shape, zero=False):
        return (torch.zeros if zero else torch.empty)(*shape, dtype=torch.float32, device=self.dev)

    def _alloc_io(self):
        R, S, u = self.R, self.S, self.unet
        self.x_in = self.f32buf(self.src_batch, u.channels, S, S, zero=True)      # fp32 NCHW staging (or the sampler's state)
        self.lowres_in = self.f32buf(self.src_batch, u.channels, S, S, zero=True) if self.lowres else None
        # Extra init-conv inputs, read by the init conv as a second (channel-concatenated) fp16 NHWC tensor `cimg` = [self_cond | cond image]:
        #  * the conditioning image (Unet(cond_images_channels=...), ip.py:1555-1560): fp32 NCHW staging, packed once per
        #    set_cond_images() — it does not change over the timesteps;
        #  * the self-conditioning input (Unet(self_cond=True), ip.py:1541-1543): the previous step's thresholded x0 (zeros at first),
        #    packed at the start of every step from `self_cond_in`, or from the sampler's buffer once bind_self_cond() has been called.
        cc = getattr(u, 'cond_images_channels', 0)
        sc = u.channels if getattr(u, 'self_cond', False) else 0
        self.cond_in = self.f32buf(self.src_batch, cc, S, S, zero=True) if cc else None
        self.self_cond_in = self.f32buf(self.src_batch, sc, S, S, zero=True) if sc else None
        self.cimg = self.new(R, S, S, (sc + cc + 7) // 8 * 8, zero=True) if sc + cc else None
        self._cond_pack = None
        self._self_cond_op = None
        self.times = self.f32buf(R, zero=True)          # log-SNR per row (plain forward mode)
        self.lowres_times = self.f32buf(R, zero=True)
        self.out = self.f32buf(R, u.channels_out, S, S)
        self.coef = None
        self.step_ptr = None
        # per-row conditioning produced by the static plan, consumed by the step plan
        self.t_const = self.new(1, 1, R, self.Tc, zero=True)               # text hiddens (+ lowres t)
        self.keep_u8 = torch.ones(R, dtype=torch.uint8, device=self.dev)
        self.src_idx = torch.zeros(R, dtype=torch.int32, device=self.dev)
        self.arange_idx = torch.arange(R, dtype=torch.int32, device=self.dev)

    # ------------------------------------------------------------------------------------------ weights
    def _time_mlp_tables(self, blocks: List[ResnetBlockP]):
        """Concatenate every ResnetBlock's time_mlp Linear (ip.py:711-714) into one GEMM; build the gather
        tables turning its output into block2's per-(row, channel) affine (ImagenScaleShiftParams)."""
        ws, bs, gam, idx_scale, idx_shift, offs = [], [], [], [], [], []
        col, coff = 0, 0
        for rb in blocks:
            lin = rb.time_mlp[1]
            C = rb.dim_out
            ws.append(lin.weight.detach().float().cpu())
            bs.append(lin.bias.detach().float().cpu())
            gam.append(rb.block2.norm.gamma.detach().float().flatten().cpu() * math.sqrt(C))
            idx_scale.append(torch.arange(col, col + C))
            idx_shift.append(torch.arange(col + C, col + 2 * C))
            offs.append(coff)
            col += 2 * C
            coff += C
        return torch.cat(ws), torch.cat(bs), torch.cat(gam), torch.cat(idx_scale).int(), torch.cat(idx_shift).int(), offs, coff

    # ------------------------------------------------------------------------------------------ step plan
    def _all_resnet_blocks(self) -> List[ResnetBlockP]:
        u = self.unet
        out = []
        if u.init_resnet_block is not None:
            out.append(u.init_resnet_block)
        for lvl in u.downs:
            out.append(lvl[1])
            out.extend(lvl[2])
        out += [u.mid_block1, u.mid_block2]
        for lvl in u.ups:
            out.append(lvl[0])
            out.extend(lvl[1])
        if u.final_res_block is not None:
            out.append(u.final_res_block)
        return out

    def _build_step_plan(self) -> Plan:
        u, R, S, W = self.unet, self.R, self.S, self.W
        plan = Plan("unet-step")
        self._plan = plan
        # ---- input image (+ low-res conditioning image) -> fp16 NHWC, 8 channels
        cin = u.channels * (2 if self.lowres else 1)
        assert cin <= 8, "init conv packs the input image into 8 channels"
        self.img = self.new(R, S, S, 8)
        self._pack_op = ops.pack_image(plan, self.x_in, self.lowres_in, self.img, brep=R // self.src_batch, label="pack_image")
        if self.self_cond_in is not None:
            self._self_cond_op = ops.pack_image(plan, self.self_cond_in, self.cond_in, self.cimg, brep=R // self.src_batch, label="pack_self_cond")

        # ---- time conditioning (ip.py:1573-1578)
        chain_begin = len(plan.ops)
        self.hid = self.new(1, 1, R, self.Tc)
        self._time_embed_op = ops.time_embed(
            plan, times=self.times, coef=None, step_ptr=None,
            freqs=W.f32("time.freqs", lambda: u.to_time_hiddens[0].weights), w=W.f32("time.w", lambda: u.to_time_hiddens[1].weight),
            bias=W.f32("time.b", lambda: u.to_time_hiddens[1].bias), hid=self.hid, label="time_embed")
        if TIME_CHAIN_F32:
            self.t = self.f32buf(R, self.Tc)
            ops.linear_f32(plan, self.hid, W.f32("time.cond.wt32", lambda: u.to_time_cond[0].weight.t()), W.f32("time.cond.b32", lambda: u.to_time_cond[0].bias),
                           self.t, res=self.t_const, label="to_time_cond")
        else:
            self.t = self.new(1, 1, R, self.Tc)
            ops.igemm(plan, self.hid, W.conv("time.cond", u.to_time_cond[0], split=SPLIT_STATIC), self.t, res=self.t_const, label="to_time_cond")
        # time tokens -> norm_cond (ip.py:1577, 1660)
        tok_raw = self.new(1, 1, R, self.ntt * self.cond_dim)
        ops.igemm(plan, self.hid, W.conv("time.tokens", u.to_time_tokens[0], split=SPLIT_STATIC), tok_raw, label="to_time_tokens")
        self.c_time = self.new(1, 1, R * self.ntt, self.cond_dim)
        tok_rows = Act(tok_raw.t, 1, 1, R * self.ntt, self.cond_dim, self.cond_dim, R * self.ntt * self.cond_dim)
        ops.ln_residual(plan, tok_rows, W.f32("norm_cond.w", lambda: u.norm_cond.weight), self.c_time,
                        beta=W.f32("norm_cond.b", lambda: u.norm_cond.bias), eps=1e-5, label="norm_cond(time)")

        # ---- all ResnetBlock time-MLPs in one GEMM + scale/shift tables (ip.py:738-741)
        blocks = self._all_resnet_blocks()
        self._blk_index = {id(rb): i for i, rb in enumerate(blocks)}
        tw, tb, gam, isc, ish, self._blk_off, total_c = W.get("timemlp.tables", lambda: self._time_mlp_tables(blocks))
        self.total_c = total_c
        if TIME_CHAIN_F32:
            ss = self.f32buf(R, tw.shape[0])
            ops.linear_f32(plan, self.t, W.f32("timemlp.wt32", lambda: tw.t()), W.f32("timemlp.b32", lambda: tb), ss, act_in=ACT_SILU, label="time_mlps")
        else:
            ss = self.new(1, 1, R, tw.shape[0])
            ops.igemm(plan, self.t, W.raw("timemlp.w", tw, tb, split=SPLIT_STATIC), ss, act_in=ACT_SILU, label="time_mlps")
        self.pa2 = self.f32buf(R, total_c)
        self.ps2 = self.f32buf(R, total_c)
        ops.scale_shift(plan, ss, W.f32("timemlp.gam", lambda: gam), W.get("timemlp.isc", lambda: isc.to(self.dev)),
                        W.get("timemlp.ish", lambda: ish.to(self.dev)), self.pa2, self.ps2)

        # ---- conditioning K/V rows that depend on the timestep (filled after the traversal registers the sites)
        self._kv_dynamic_anchor = len(plan.ops)
        self._time_chain_ops = [st for _, st, label in plan.ops[chain_begin:] if label in self._TIME_CHAIN]

        # ---- U-net traversal
        x = self.new(R, S, S, self.lc["init_dim"])
        self._init_conv(plan, x)
        self.taps['init_conv'] = x
        init_res = x if u.init_conv_to_final_conv_residual else None      # ip.py:1568-1569 (buffers are never overwritten: no clone)
        if u.init_resnet_block is not None:
            x = self._resnet(plan, x, None, u.init_resnet_block, "init_resnet", with_cond=False)
        hiddens: List[Act] = []
        n_levels = len(self.lc["in_out"])
        mem_eff = self.lc["memory_efficient"]
        for i, lvl in enumerate(u.downs):
            pre, init_block, res_blocks, attn_block, post = lvl
            if pre is not None:
                x = self._downsample(plan, x, pre, f"downs.{i}.0")
            x = self._resnet(plan, x, None, init_block, f"downs.{i}.1", with_cond=True)
            for j, rb in enumerate(res_blocks):
                x = self._resnet(plan, x, None, rb, f"downs.{i}.2.{j}", with_cond=False)
                hiddens.append(x)
            if isinstance(attn_block, TransformerBlockP):
                x = self._transformer(plan, x, attn_block, f"downs.{i}.3", with_context=True)
            hiddens.append(x)
            self.taps[f'down{i}'] = x
            if post is not None:
                x = self._downsample(plan, x, post, f"downs.{i}.4")
        x = self._resnet(plan, x, None, u.mid_block1, "mid_block1", with_cond=True)
        if u.mid_attn is not None:
            x = self._transformer(plan, x, u.mid_attn, "mid_attn", with_context=False)
        x = self._resnet(plan, x, None, u.mid_block2, "mid_block2", with_cond=True)
        self.taps['mid'] = x
        up_hiddens: List[Act] = []
        for i, lvl in enumerate(u.ups):
            init_block, res_blocks, attn_block, upsample = lvl
            x = self._resnet(plan, x, hiddens.pop(), init_block, f"ups.{i}.0", with_cond=True)
            for j, rb in enumerate(res_blocks):
                x = self._resnet(plan, x, hiddens.pop(), rb, f"ups.{i}.1.{j}", with_cond=False)
            if isinstance(attn_block, TransformerBlockP):
                x = self._transformer(plan, x, attn_block, f"ups.{i}.2", with_context=True)
            up_hiddens.append(x)                                            # ip.py:1707
            if isinstance(upsample, PixelShuffleUpsampleP):
                x = self._upsample(plan, x, upsample, f"ups.{i}.3")
            elif isinstance(upsample, nn.Sequential):
                x = self._upsample_nearest_conv(plan, x, upsample, f"ups.{i}.3")
            self.taps[f'up{i}'] = x
        assert not hiddens
        if getattr(u.upsample_combiner, 'enabled', False):
            x = self._combine_upsample_fmaps(plan, x, up_hiddens)
        if u.final_res_block is not None:   # with init_conv_to_final_conv_residual its input is cat(x, init conv output), unscaled (ip.py:1716-1720)
            x = self._resnet(plan, x, init_res, u.final_res_block, "final_res_block", with_cond=False, skip_scale=1.0)
        elif init_res is not None:          # no final resnet block: final_conv itself reads cat(x, init conv output[, lowres image]) — the two
            cat = self.new(R, x.H, x.W, x.C + init_res.C)      # feature tensors are joined first (two strided row copies), the image rides as x2
            for src, off in ((x, 0), (init_res, x.C)):
                ops.rows_copy(plan, src.t, cat.t, B=1, rows=R * x.H * x.W, C=src.C, src_bs=0, src_rs=src.ld, dst_bs=0, dst_rs=cat.C,
                              src_off=src.off, dst_off=off, label="final_conv.cat")
            x = cat
        self.taps['final_res'] = x
        self._final_conv(plan, x)

        # ---- now that every attention site is known: the per-step conditioning K/V ops, spliced in before the traversal
        dyn = Plan("kv-dynamic")
        self._dyn_proj = self._emit_context_kv(dyn, self.c_time, rows_per_batch=self.ntt, k_row0_self=0, k_row0_cross=1, tag="dyn")
        plan.ops[self._kv_dynamic_anchor:self._kv_dynamic_anchor] = dyn.ops
        self._time_chain_ops += [st for _, st, label in dyn.ops if label in self._TIME_CHAIN]
        plan.keep.extend(dyn.keep)
        plan._arr = None
        return plan

    # ---- init / final convs
    def _init_conv(self, plan, out: Act):
        """CrossEmbedLayer (ip.py:1051-1076, stride 1) as ONE kmax x kmax conv with the smaller kernels zero-embedded;
        or the plain init conv (ip.py:1198)."""
        u = self.unet

        cc = self.cond_in.shape[1] if self.cond_in is not None else 0
        sc = self.self_cond_in.shape[1] if self.self_cond_in is not None else 0
        c = u.channels
        nl = c if self.lowres else 0
        auxp = self.cimg.C if self.cimg is not None else 0

        def spread(w):
            """Reference input channels [cond image | x | self_cond | lowres] (ip.py:1541-1560) ->
            ours [x | lowres | 0.. (8)] ++ [self_cond | cond image | 0.. (auxp)]."""
            assert w.shape[1] == cc + c + sc + nl
            wp = torch.zeros(w.shape[0], 8 + auxp, *w.shape[2:])
            wp[:, :c] = w[:, cc: cc + c]
            wp[:, c: c + nl] = w[:, cc + c + sc:]
            wp[:, 8: 8 + sc] = w[:, cc + c: cc + c + sc]
            wp[:, 8 + sc: 8 + sc + cc] = w[:, :cc]
            return wp

        def make():
            if isinstance(u.init_conv, CrossEmbedP):
                kmax = max(u.init_conv.kernel_sizes)
                ws, bs = [], []
                for conv, k in zip(u.init_conv.convs, u.init_conv.kernel_sizes):
                    w = torch.zeros(*conv.weight.shape[:2], kmax, kmax)
                    p = (kmax - k) // 2
                    w[:, :, p:p + k, p:p + k] = conv.weight.detach().float()
                    ws.append(spread(w))
                    bs.append(conv.bias.detach().float())
                return ops.pack_weight(torch.cat(ws), torch.cat(bs), self.dev, G=1)
            conv = u.init_conv
            return ops.pack_weight(spread(conv.weight.detach().float()), conv.bias.detach().float(), self.dev, G=1)

        out.ssq = self.f32buf(out.rows)
        R, B = self.R, self.src_batch
        px = out.H * out.W
        # classifier-free guidance runs rows [B, 2B) on the SAME image: nothing conditions the init conv (ip.py:1562), so it is computed
        # for the first B rows and copied — on the 256^2 stage half of a 150 us launch against two copies of 17 + 5 us
        shared = (INIT_CONV_SHARED and R == 2 * B and B * px >= INIT_CONV_SHARED_MIN_PIXELS and out.ld == out.C and out.bs == px * out.C and px % 4 == 0)
        if shared:
            half = lambda a: Act(a.t, B, a.H, a.W, a.C, a.ld, a.bs, a.off)
            op = ops.igemm(plan, half(self.img), self.W.get("init_conv", make), half(out), x2=half(self.cimg) if self.cimg is not None else None,
                           ssq_out=out.ssq, label="init_conv")
            ops.rows_copy(plan, out.t, out.t, B=1, rows=B * px, C=out.C, src_bs=0, src_rs=out.C, dst_bs=0, dst_rs=out.C, src_off=out.off,
                          dst_off=out.off + B * px * out.C, label="init_conv.cfg_rows")
            if op.ssq_emitted:   # the statistics too: fp32 [rows] seen as rows of 4 floats (8 halves)
                ssq16 = out.ssq.view(torch.float16)
                ops.rows_copy(plan, ssq16, ssq16, B=1, rows=B * px // 4, C=8, src_bs=0, src_rs=8, dst_bs=0, dst_rs=8, src_off=0, dst_off=2 * B * px,
                              label="init_conv.cfg_rows.ssq")
                plan.keep.append(out.ssq)
        else:
            op = ops.igemm(plan, self.img, self.W.get("init_conv", make), out, x2=self.cimg, ssq_out=out.ssq, label="init_conv")
        if not op.ssq_emitted:
            out.ssq = None

    def _final_conv(self, plan, x: Act):
        """final_conv over cat(x, lowres_cond_img) (ip.py:1722-1725) -> fp32 NCHW."""
        u = self.unet
        extra = self.img if self.lowres else None  # channels [C, 2C) of the packed image are the low-res image
        kh = u.final_conv.weight.shape[-1]
        split = extra is None and self._split_small(x.C, kh * kh, u.final_conv.weight.shape[0], x.rows)

        def make():
            w = u.final_conv.weight.detach().float()
            co, ci, kh, kw = w.shape
            if extra is None:
                return ops.pack_weight(w, u.final_conv.bias.detach().float(), self.dev, split=split)
            wp = torch.zeros(co, x.C + 8, kh, kw)
            wp[:, : x.C] = w[:, : x.C]
            # packed image channel layout: [x (C) | lowres (C) | zero pad]; the reference concatenates lowres after the features
            wp[:, x.C + u.channels: x.C + 2 * u.channels] = w[:, x.C:]
            # 32-channel k-chunks whenever the feature part allows it (the 8 image channels then occupy one group of a second,
            # otherwise zero chunk): the 8-channel-chunk path (G = 1) took 100 us for this layer at 256^2, twice a 32->32 conv
            G = 4 if x.C % 32 == 0 else None
            return ops.pack_weight(wp, u.final_conv.bias.detach().float(), self.dev, G=G)

        ops.igemm(plan, x, self.W.get("final_conv|split" if split else "final_conv", make), self.out, x2=extra, out_mode=OUT_NCHW_F32, label="final_conv")

    def _combine_upsample_fmaps(self, plan, x: Act, fmaps: List[Act]) -> Act:
        """UpsampleCombiner (ip.py:1078-1110, 1712): cat(x, Block_i(nearest-resize(fmap_i))) over the up levels.  Kernels of the image
        path only: the nearest resize by an integer factor f is f*f strided row copies (output pixel (y*f + dy, x*f + dx) takes input
        pixel (y, x): one copy per (dy, dx) with the input rows as the batch), the Block is the fused norm -> SiLU -> 3x3 conv, each
        writing its channel slice of the concatenated tensor directly.  A non-default flag: nothing here is tuned."""
        u, R = self.unet, self.R
        comb = u.upsample_combiner
        S, dim = x.H, x.C
        douts = [blk.project.weight.shape[0] for blk in comb.fmap_convs]
        Ctot = dim + sum(douts)
        assert Ctot == comb.dim_out and dim % 8 == 0 and all(d % 8 == 0 for d in douts), "combiner slices must be 8-channel aligned"
        cat = self.new(R, S, S, Ctot)
        ops.rows_copy(plan, x.t, cat.t, B=1, rows=R * S * S, C=dim, src_bs=0, src_rs=x.ld, dst_bs=0, dst_rs=Ctot, src_off=x.off,
                      label="upsample_combiner.x")
        off = dim
        for i, (f_act, blk, dout) in enumerate(zip(fmaps, comb.fmap_convs, douts)):
            name = f"upsample_combiner.fmap_convs.{i}"
            C = f_act.C
            assert f_act.ld == C and C % 8 == 0
            if f_act.H != S:
                assert S % f_act.H == 0 and f_act.H == f_act.W, "nearest resize by an integer factor only"
                f, h = S // f_act.H, f_act.H
                up = self.new(R, S, S, C)
                for dy in range(f):
                    for dx in range(f):
                        ops.rows_copy(plan, f_act.t, up.t, B=R * h, rows=h, C=C, src_bs=h * C, src_rs=C, dst_bs=f * S * C, dst_rs=f * C,
                                      src_off=f_act.off, dst_off=(dy * S + dx) * C, label=f"{name}.resize")
            else:
                up = f_act
            ssq = self._ssq_of(plan, up, name + ".stat")
            w = self.W.conv(name, blk.project)
            pa = self.W.f32(name + ".pa", lambda blk=blk, C=C, w=w: _pad_vec(blk.norm.gamma.detach().float().flatten().cpu() * math.sqrt(C), w.Cin_pad))
            dst = Act(cat.t, R, S, S, dout, Ctot, S * S * Ctot, off)
            ops.igemm(plan, up, w, dst, ssq_a=ssq, pa=pa, pstride=0, act_in=ACT_SILU, label=name)
            off += dout
        return cat

    # ---- ResnetBlock (ip.py:693-757)
    def _resnet(self, plan, x: Act, skip: Optional[Act], rb: ResnetBlockP, name: str, with_cond: bool, skip_scale: Optional[float] = None) -> Act:
        W, R = self.W, self.R
        C1, C2 = x.C, (skip.C if skip is not None else 0)
        Cin, Cout = C1 + C2, rb.dim_out
        assert Cin == rb.dim, f"{name}: {Cin} input channels, block expects {rb.dim}"
        s = self.unet.skip_connect_scale if skip_scale is None else skip_scale
        H, Wd = x.H, x.W
        in_scale = None
        if skip is not None:
            in_scale = torch.ones(Cin)
            in_scale[C1:] = s
        # block1: ChanRMSNorm over the (scaled) concat -> SiLU -> conv3x3.  The norm statistics come from the per-pixel sums of
        # squares the producers of x / skip emitted in their epilogues (a ROWSTAT pass only where a producer could not).
        stat_x = lambda: self._ssq_of(plan, x, name + ".block1.stat_x")
        # what the launch that produced x may read of the skip's statistics (request_prep below wires them into THAT earlier launch): only sums of
        # squares a producer already emitted — _ssq_of() would append a ROWSTAT at the current plan position, behind the launch that needs them
        skip_ssq_ready = skip.ssq if skip is not None else None
        ss = self._ssq_of(plan, skip, name + ".block1.stat_skip") if skip is not None else None
        w1 = W.conv(name + ".block1", rb.block1.project, split=skip is None and self._split_small(Cin, 9, Cout, R * H * Wd))
        pa1 = W.f32(name + ".block1.pa", lambda: _pad_vec(rb.block1.norm.gamma.detach().float().flatten().cpu() * math.sqrt(Cin)
                                                         * (in_scale if in_scale is not None else 1.0), w1.Cin_pad))
        off = self._blk_off[self._blk_index[id(rb)]]
        pa2 = self.pa2[:, off:]
        ps2 = self.ps2[:, off:]
        h1 = self.new(R, H, Wd, Cout)
        h1.ssq = self.f32buf(R * H * Wd)
        # without a cross-attention in between, block1's epilogue applies block2's ChanRMSNorm -> (scale+1, shift) -> SiLU itself
        # (its consumer waves have the slack; block2's producers then stage h1 with no arithmetic at all)
        post = dict(pa=pa2, ps=ps2, pstride=self.total_c) if rb.cross_attn is None else None
        prep = ops.CONV_DMA and ACT_PREP_MIN_COUT > 0 and Cout >= ACT_PREP_MIN_COUT and Cin % 32 == 0 and Cout % 32 == 0
        # the big-tile all-DMA family (conv_big.hip) wants activated inputs: where it applies, the Block prologue runs once per element as
        # its own pass (ACT_PREP, which also reduces x's statistics itself where no producer emitted them) instead of once per staging
        # workgroup inside the wave-specialised kernel
        big = BIG_PREP and ops.big_cfg(Cout, H, Wd, R) is not None
        big1 = big and Cin % 32 == 0 and Cin <= 512 and w1.Cin_pad == Cin and not w1.split
        # x comes out of a fused ResnetBlock tail (GCA_TAIL): that launch also writes silu(ChanRMSNorm(x) * gamma) — it has the pixel's
        # channels and its sum of squares in registers — and block1 stages its input with no arithmetic (prologue-free kernel families)
        xa = ops.request_act(x, pa1) if (skip is None and TAIL_ACT and w1.Cin_pad == w1.Cin) else None
        # ... or out of the previous block's res_conv as a ROWCHAIN launch (RESPREP), which then writes the activated concat(x, skip) as well
        xr = ops.request_prep(x, skip, skip_ssq_ready, s * s, pa1) if (xa is None and (prep or big1)) else None
        if xa is not None:
            op = ops.igemm(plan, xa, w1, h1, ssq_out=h1.ssq, post=post, label=name + ".block1")
        elif xr is not None:     # (no ACT_PREP pass)
            op = ops.igemm(plan, Act(xr.t, R, H, Wd, Cin, Cin, H * Wd * Cin), w1, h1, ssq_out=h1.ssq, post=post, label=name + ".block1")
        elif prep or big1:   # MFMA-bound layer: the prologue as its own pass, then an all-DMA conv on the activated concat
            xa = self.new(R, H, Wd, Cin)
            if big1 and x.ssq is None:
                ops.act_prep(plan, x, xa, x2=skip, ssq_b=ss, ssq_wb=s * s, pa=pa1, pstride=0, act_in=ACT_SILU, self_stat=True, label=name + ".block1.prep")
            else:
                ops.act_prep(plan, x, xa, x2=skip, ssq_a=stat_x(), ssq_b=ss, ssq_wb=s * s, pa=pa1, pstride=0, act_in=ACT_SILU, label=name + ".block1.prep")
            op = ops.igemm(plan, xa, w1, h1, ssq_out=h1.ssq, post=post, label=name + ".block1")
        else:
            op = ops.igemm(plan, x, w1, h1, x2=skip, ssq_a=stat_x(), ssq_b=ss, ssq_wb=s * s, pa=pa1, pstride=0, act_in=ACT_SILU, ssq_out=h1.ssq,
                           post=post, label=name + ".block1")
        if not op.ssq_emitted:
            h1.ssq = None
        if rb.cross_attn is not None:
            assert with_cond
            h1 = self._cross_attn(plan, h1, rb.cross_attn, name + ".cross_attn")
        h2 = self.new(R, H, Wd, Cout)
        gate, gca_args = None, None
        if rb.gca is not None:
            g = rb.gca
            hidden = g.net[0].weight.shape[0]
            gate = self.f32buf(R, Cout)
            gca_args = dict(wk=W.f32(name + ".gca.wk", lambda: g.to_k.weight.reshape(-1)), bk=float(g.to_k.bias.detach().float().item()),
                            w1t=W.f32(name + ".gca.w1t", lambda: g.net[0].weight.reshape(hidden, Cout).t()),
                            b1=W.f32(name + ".gca.b1", lambda: g.net[0].bias),
                            w2t=W.f32(name + ".gca.w2t", lambda: g.net[2].weight.reshape(Cout, hidden).t()),
                            b2=W.f32(name + ".gca.b2", lambda: g.net[2].bias), gate=gate)
        gca_ep = dict(wk=gca_args["wk"], bk=gca_args["bk"]) if rb.gca is not None else None   # GlobalContext partials from block2's epilogue
        if op.post_applied:     # h1 already holds silu(norm(h1) * (scale + 1) + shift)
            op2 = ops.igemm(plan, h1, W.conv(name + ".block2", rb.block2.project, split=SPLIT_BLOCK2 and self._split_small(Cout, 9, Cout, R * H * Wd)), h2,
                            gca=gca_ep, label=name + ".block2")
        else:                   # block2: ChanRMSNorm -> (scale+1, shift) from the time MLP -> SiLU -> conv3x3
            w2 = W.conv(name + ".block2", rb.block2.project)
            if (prep or big) and Cout % 32 == 0 and Cout <= 512:   # the prologue as its own pass (it reduces h1's statistics itself where block1 could not emit them)
                ha = self.new(R, H, Wd, Cout)
                if h1.ssq is None and big:
                    ops.act_prep(plan, h1, ha, pa=pa2, ps=ps2, pstride=self.total_c, act_in=ACT_SILU, self_stat=True, label=name + ".block2.prep")
                else:
                    ops.act_prep(plan, h1, ha, ssq_a=self._ssq_of(plan, h1, name + ".block2.stat"), pa=pa2, ps=ps2, pstride=self.total_c,
                                 act_in=ACT_SILU, label=name + ".block2.prep")
                op2 = ops.igemm(plan, ha, w2, h2, gca=gca_ep, label=name + ".block2")
            else:
                op2 = ops.igemm(plan, h1, w2, h2, ssq_a=self._ssq_of(plan, h1, name + ".block2.stat"), pa=pa2, ps=ps2, pstride=self.total_c,
                                act_in=ACT_SILU, gca=gca_ep, label=name + ".block2")
        # identity block: GlobalContext finalisation + h2 * gate + x (+ statistics) as ONE launch (GCA_TAIL) where its shapes allow
        fused_tail = (TAIL_FUSED and rb.res_conv is None and x.ld == x.C and x.bs == H * Wd * x.C
                      and ops.gca_tail_ok(Cout, (hidden if rb.gca is not None else None)))
        tail_part, tail_chunks, gate_ready = None, 0, False
        if rb.gca is not None:
            wide = ops.GCA_FINAL_SPLIT and ops.gca_final_is_wide(Cout, hidden)   # (a fused tail would stream 1-4 MB of MLP weights in EVERY workgroup)
            if op2.gca_part_t is not None:   # the partials came out of block2's epilogue: only the merge + squeeze MLP is left
                if fused_tail and op2.gca_chunks <= 1024 and not wide:
                    tail_part, tail_chunks = op2.gca_part_t, op2.gca_chunks
                else:
                    ops.gca_final(plan, op2.gca_part_t, gca_args["w1t"], gca_args["b1"], gca_args["w2t"], gca_args["b2"], gate, B=R, C=Cout,
                                  chunks=op2.gca_chunks, label=name + ".gca")
                    gate_ready = True
            else:                            # stand-alone pass over h2 (+ in-kernel finalisation where one workgroup covers the image)
                chunks = ops.gca_chunks(H * Wd, R, Cout)
                part = self.f32buf(R, chunks, Cout + 2)
                gate_ready = ops.gca(plan, h2, gca_args["wk"], gca_args["bk"], gca_args["w1t"], gca_args["b1"], gca_args["w2t"], gca_args["b2"], part,
                                     gate, chunks, label=name + ".gca", final=(not fused_tail) or wide)
                if not gate_ready:
                    tail_part, tail_chunks = part, chunks
        out = self.new(R, H, Wd, Cout)
        out.ssq = self.f32buf(R * H * Wd)
        if fused_tail:
            assert skip is None
            ops.gca_tail(plan, h2, x, out, part=tail_part, chunks=tail_chunks, w1t=gca_args and gca_args["w1t"], b1=gca_args and gca_args["b1"],
                         w2t=gca_args and gca_args["w2t"], b2=gca_args and gca_args["b2"], gate_in=gate if gate_ready else None,
                         gate=gate if tail_part is not None else None, ssq_out=out.ssq, label=name + ".tail")
        elif rb.res_conv is not None:
            wr = W.conv(name + ".res_conv", rb.res_conv, in_scale=in_scale)
            if big and ops.resprep_ok(x, skip, wr, H * Wd):
                # on the levels whose Blocks take the big-tile all-DMA convs (their prologue is a pass of its own), the res_conv + gate tail is
                # a ROWCHAIN launch that the NEXT block asks for its activated input too (request_prep): that block needs no ACT_PREP
                op = ops.rowchain_resprep(plan, x.tokens(), skip.tokens() if skip is not None else None, h2.tokens(), gate, out.tokens(), wr,
                                          rows_per_batch=H * Wd, ssq_out=out.ssq, label=name + ".res_conv.chain")
                out.res_op = (op, plan)
                op.ssq_emitted = True
            elif gate is not None:
                op = ops.igemm(plan, x, wr, out, x2=skip, addend=h2, gate=gate, ssq_out=out.ssq, label=name + ".res_conv")
            else:
                op = ops.igemm(plan, x, wr, out, x2=skip, res=h2, ssq_out=out.ssq, label=name + ".res_conv")
            if not op.ssq_emitted:
                out.ssq = None
        else:
            assert skip is None
            ops.gate_residual(plan, h2, gate, x, out, rs_out=out.ssq, raw_ssq=True, label=name + ".tail")
        return out

    def _ssq_of(self, plan, a: Act, label: str) -> torch.Tensor:
        """Per-pixel sum of squares of `a`: the producer's epilogue output if it emitted one, else one ROWSTAT (mode 2) pass."""
        if a.ssq is None:
            a.ssq = self.f32buf(a.rows)
            ops.rowstat(plan, a, mode=2, rs=a.ssq, label=label)
        return a.ssq

    # ---- CrossAttention inside a ResnetBlock (ip.py:745-751, 759-834)
    def _cross_attn(self, plan, h: Act, ca: CrossAttentionP, name: str) -> Act:
        W, R = self.W, self.R
        N, C = h.H * h.W, h.C
        heads, dh = ca.heads, ca.dim_head
        inner = heads * dh
        tok = h.tokens()
        wq = W.conv(name + ".to_q", ca.to_q, split=SPLIT_1X1 and self._split_small(C, 1, inner, R * N))
        wo = W.conv(name + ".to_out", ca.to_out[0], split=SPLIT_1X1 and self._split_small(inner, 1, C, R * N))
        J = self.NT + 1
        Jp = ops._round_up(J, 32)
        khat = torch.zeros(R, heads, Jp, dh, dtype=torch.float16, device=self.dev)
        vt = torch.zeros(R, heads, dh, Jp, dtype=torch.float16, device=self.dev)
        site = dict(kind="cross", name=name, mod=ca, khat=khat, vt=vt, heads=heads, Jp=Jp, dh=dh,
                    k_strides=(heads * Jp * dh, Jp * dh, dh), vt_strides=(heads * dh * Jp, dh * Jp, Jp))
        self.attn_sites.append(site)
        out = self.new(R, h.H, h.W, C)
        out.ssq = self.f32buf(R * N)
        g_norm = W.f32(name + ".norm.g", lambda: _pad_vec(ca.norm.g, wq.Cin_pad))
        g_out = W.f32(name + ".out_g", lambda: ca.to_out[1].g)
        q_scale = W.f32(name + ".q_scale", lambda: ca.q_scale)
        if ops.rowchain_ok(C, N, heads, dh, wq, wo) and h.ld == C:
            # the whole cross-attention as ONE launch (ROWCHAIN mode XATTN): its rows only meet constants of their image
            ops.rowchain_xattn(plan, tok, out.tokens(), wq, g_norm, wo, g_out, khat, vt, heads=heads, J=J, k_strides=site["k_strides"],
                               vt_strides=site["vt_strides"], q_scale=q_scale, q_mult=SIM_SCALE * LOG2E, rows_per_batch=N, ssq_out=out.ssq,
                               label=name + ".chain")
            return out
        mu, rs = self.f32buf(R * N), self.f32buf(R * N)
        ops.rowstat(plan, tok, mode=1, rs=rs, mu=mu, eps=1e-5, label=name + ".norm")
        q = self.new(R, 1, N, inner)
        ops.igemm(plan, tok, wq, q, mu=mu, rs=rs, pa=g_norm, label=name + ".to_q")
        o = self.new(R, 1, N, inner)
        ops.attention(plan, q.t, khat, vt, o.t, B=R, heads=heads, rows=N, J=J, q_strides=(N * inner, dh, inner),
                      k_strides=site["k_strides"], vt_strides=site["vt_strides"], o_strides=(N * inner, dh, inner),
                      q_scale=q_scale, q_mult=SIM_SCALE * LOG2E, label=name + ".attn", head_dim=dh,
                      logit_bound=ops.attention_logit_bound(ca.q_scale, ca.k_scale, SIM_SCALE * LOG2E))
        y = self.new(R, 1, N, C)
        ops.igemm(plan, o, wo, y, label=name + ".to_out")
        ops.ln_residual(plan, y, g_out, out.tokens(), res=tok, eps=1e-5, ssq_out=out.ssq, label=name + ".out_norm")
        return out

    # ---- TransformerBlock (ip.py:992-1022): depth x [multi-query self attention + FeedForward]
    def _transformer(self, plan, x: Act, tb: TransformerBlockP, name: str, with_context: bool) -> Act:
        R, W = self.R, self.W
        N, C = x.H * x.W, x.C
        cur = x
        for d, (attn, ff) in enumerate(tb.layers):
            nm = f"{name}.layers.{d}"
            # LayerNorm statistics of the attention input from the launch that produced it (a fused ResnetBlock tail), where there is one
            stats_in = ops.request_ln_stats(cur) if LN_STATS_FUSED else None
            hidden = ff[1].weight.shape[0]
            heads, dh = attn.heads, attn.dim_head
            inner = heads * dh
            w_out = W.conv(nm + ".to_out", attn.to_out[0], split=SPLIT_1X1 and self._split_small(inner, 1, C, R * N))
            split_ff = SPLIT_1X1 and self._split_small(2 * C, 1, C, R * N)
            w1, w2 = W.conv(nm + ".ff.w1", ff[1], split=split_ff), W.conv(nm + ".ff.w2", ff[4], split=split_ff)
            if ops.rowchain_ok(C, N, heads, dh, w_out, w1, w2, hidden=hidden) and cur.ld == C:
                # attention out-projection -> LayerNorm + residual -> FeedForward as ONE launch behind the attention (ROWCHAIN mode FF)
                o = self._self_attn(plan, cur.tokens(), attn, nm, with_context, ln_stats=stats_in, stop_at_attention=True)
                out = self.new(R, 1, N, C)
                out.ssq = self.f32buf(R * N)
                ops.rowchain_ff(plan, o, cur.tokens(), out, w_out, W.f32(nm + ".out_g", lambda: attn.to_out[1].g), w1,
                                W.f32(nm + ".ff.g0", lambda: _pad_vec(ff[0].g, w1.Cin_pad)), w2,
                                W.f32(nm + ".ff.g1", lambda: _pad_vec(ff[3].g, w2.Cin_pad)), rows_per_batch=N, ssq_out=out.ssq, label=nm + ".ff.chain")
                cur = Act(out.t, R, x.H, x.W, C, C, N * C, ssq=out.ssq)
                continue
            x1, st = self._self_attn(plan, cur.tokens(), attn, nm, with_context, ln_stats=stats_in, want_stats=True)
            ffo = self._feed_forward(plan, x1, ff, nm + ".ff", ln_stats=st, split=split_ff)
            cur = Act(ffo.t, R, x.H, x.W, C, C, N * C, ssq=ffo.ssq)
        return cur

    def _self_attn(self, plan, tok: Act, attn, nm: str, with_context: bool, ln_stats: Optional[tuple] = None, want_stats: bool = False,
                   stop_at_attention: bool = False):
        """attn(tok) + tok for the (R, 1, N, C) token view `tok` (ip.py:502-591, 1017): LayerNorm -> q | k | v in one GEMM ->
        K^/V^T rows behind the conditioning and null rows -> flash attention -> to_out -> LayerNorm + residual."""
        W, R = self.W, self.R
        N, C = tok.H * tok.W, tok.C
        heads, dh = attn.heads, attn.dim_head
        inner = heads * dh
        # q | k | v from ONE GEMM (to_q and to_kv are both bias-free on the same normalised input, ip.py:539)
        wqkv = W.raw(nm + ".qkv", torch.cat((attn.to_q.weight.detach().float(), attn.to_kv.weight.detach().float())), None,
                      split=SPLIT_1X1 and self._split_small(C, 1, inner + 2 * dh, R * N))
        qkv = self.new(R, 1, N, inner + 2 * dh)
        ld = inner + 2 * dh
        n_ctx = self.NT if (with_context and attn.to_context is not None) else 0
        J = n_ctx + 1 + N
        Jp = ops._round_up(J, 32)
        khat = torch.zeros(R, Jp, dh, dtype=torch.float16, device=self.dev)
        vt = torch.zeros(R, dh, Jp, dtype=torch.float16, device=self.dev)
        k_strides, vt_strides = (Jp * dh, 0, dh), (dh * Jp, 0, Jp)
        site = dict(kind="self", name=nm, mod=attn, khat=khat, vt=vt, heads=1, Jp=Jp, n_ctx=n_ctx, dh=dh, k_strides=k_strides, vt_strides=vt_strides)
        self.attn_sites.append(site)
        g_norm = W.f32(nm + ".norm.g", lambda: _pad_vec(attn.norm.g, wqkv.Cin_pad))
        k_scale = W.f32(nm + ".k_scale", lambda: attn.k_scale)
        if ops.rowchain_ok(C, N, heads, dh, wqkv) and tok.ld == C:
            # LayerNorm -> q | k | v -> q rows, K^ rows, V^T columns as ONE launch (ROWCHAIN mode QKV); the statistics its producer emitted, if any
            ops.rowchain_qkv(plan, tok, qkv, wqkv, g_norm, khat, vt, k_scale, heads=heads, r0=n_ctx + 1, k_strides=k_strides, vt_strides=vt_strides,
                             rows_per_batch=N, ln_stats=ln_stats, label=nm + ".qkv.chain")
        else:
            if ln_stats is not None:
                mu, rs = ln_stats
            else:
                mu, rs = self.f32buf(R * N), self.f32buf(R * N)
                ops.rowstat(plan, tok, mode=1, rs=rs, mu=mu, eps=1e-5, label=nm + ".norm")
            ops.igemm(plan, tok, wqkv, qkv, mu=mu, rs=rs, pa=g_norm, label=nm + ".qkv")
            ops.kv_prep(plan, qkv.t, qkv.t, k_scale, khat, vt, B=R, heads=1, rows=N, r0=n_ctx + 1,
                        src_strides=(N * ld, ld, 0), k_strides=k_strides, vt_strides=vt_strides, k_off=inner, v_off=inner + dh,
                        label=nm + ".kv_self", head_dim=dh)
        o = self.new(R, 1, N, inner)
        ops.attention(plan, qkv.t, khat, vt, o.t, B=R, heads=heads, rows=N, J=J, q_strides=(N * ld, dh, ld), k_strides=k_strides,
                      vt_strides=vt_strides, o_strides=(N * inner, dh, inner), q_scale=W.f32(nm + ".q_scale", lambda: attn.q_scale),
                      q_mult=SIM_SCALE * LOG2E, label=nm + ".attn", head_dim=dh,
                      logit_bound=ops.attention_logit_bound(attn.q_scale, attn.k_scale, SIM_SCALE * LOG2E))
        if stop_at_attention:       # (the caller runs the out-projection, the LayerNorm + residual and the FeedForward as one ROWCHAIN launch)
            return o
        y = self.new(R, 1, N, C)
        ops.igemm(plan, o, W.conv(nm + ".to_out", attn.to_out[0], split=SPLIT_1X1 and self._split_small(inner, 1, C, R * N)), y, label=nm + ".to_out")
        x1 = self.new(R, 1, N, C)
        st = (self.f32buf(R * N), self.f32buf(R * N)) if (LN_STATS_FUSED and want_stats) else None   # statistics of x1 for a FeedForward's first LayerNorm (a second pass over the row: only where one follows)
        ops.ln_residual(plan, y, W.f32(nm + ".out_g", lambda: attn.to_out[1].g), x1, res=tok, eps=1e-5, ln_stats_out=st, label=nm + ".out_norm")
        return x1, st

    def _feed_forward(self, plan, x: Act, ff: nn.Sequential, name: str, ln_stats: Optional[tuple] = None, split: bool = False) -> Act:
        """ip.py:972-980 + residual (ip.py:1018): LN -> Linear -> GELU -> LN -> Linear, + x.  ln_stats: (mean, rstd) of x's rows where its
        producer emitted them."""
        W = self.W
        rows = x.rows
        hidden = ff[1].weight.shape[0]
        if ln_stats is not None:
            mu, rs = ln_stats
        else:
            mu, rs = self.f32buf(rows), self.f32buf(rows)
            ops.rowstat(plan, x, mode=1, rs=rs, mu=mu, eps=1e-5, label=name + ".ln0")
        w1 = W.conv(name + ".w1", ff[1], split=split)
        hid = self.new(x.B, x.H, x.W, hidden)
        ops.igemm(plan, x, w1, hid, mu=mu, rs=rs, pa=W.f32(name + ".g0", lambda: _pad_vec(ff[0].g, w1.Cin_pad)), act_out=ACT_GELU,
                  label=name + ".lin1")
        mu2, rs2 = self.f32buf(rows), self.f32buf(rows)
        ops.rowstat(plan, hid, mode=1, rs=rs2, mu=mu2, eps=1e-5, label=name + ".ln1")
        w2 = W.conv(name + ".w2", ff[4], split=split)
        out = self.new(x.B, x.H, x.W, x.C)
        out.ssq = self.f32buf(rows)
        op = ops.igemm(plan, hid, w2, out, mu=mu2, rs=rs2, pa=W.f32(name + ".g1", lambda: _pad_vec(ff[3].g, w2.Cin_pad)), res=x,
                       ssq_out=out.ssq, label=name + ".lin2")
        if not op.ssq_emitted:
            out.ssq = None
        return out

    # ---- resampling
    def _downsample(self, plan, x: Act, mod, name: str) -> Act:
        R = self.R
        if isinstance(mod, ParallelP):  # last level (ip.py:1366): conv3x3 + conv1x1 summed == one 3x3 conv with the 1x1 folded into its centre tap
            def make():
                w3 = mod.fns[0].weight.detach().float().clone()
                w3[:, :, 1, 1] += mod.fns[1].weight.detach().float()[:, :, 0, 0]
                return ops.pack_weight(w3, mod.fns[0].bias.detach().float() + mod.fns[1].bias.detach().float(), self.dev)
            w = self.W.get(name, make)
            out = self.new(R, x.H, x.W, w.Cout)
            out.ssq = self.f32buf(out.rows)
            if not ops.igemm(plan, x, w, out, ssq_out=out.ssq, label=name).ssq_emitted:
                out.ssq = None
            return out
        conv = mod[1]  # pixel-unshuffle + 1x1 conv (ip.py:633-640) == 2x2 stride-2 conv
        def make():
            w = conv.weight.detach().float()
            return ops.pack_weight(w.view(w.shape[0], x.C, 2, 2), conv.bias.detach().float(), self.dev)
        w = self.W.get(name, make)
        out = self.new(R, x.H // 2, x.W // 2, w.Cout)
        out.ssq = self.f32buf(out.rows)
        if not ops.igemm(plan, x, w, out, stride=2, pad=0, ssq_out=out.ssq, label=name).ssq_emitted:
            out.ssq = None
        return out

    def _upsample(self, plan, x: Act, mod: PixelShuffleUpsampleP, name: str) -> Act:
        conv = mod.net[0]
        c4 = conv.weight.shape[0]
        cq = c4 // 4
        perm = torch.arange(c4).view(cq, 4).t().reshape(-1)  # PixelShuffle channel c*4 + s -> packed order (s, c)
        w = self.W.conv(name, conv, out_perm=perm)
        out = self.new(self.R, 2 * x.H, 2 * x.W, cq)
        ops.igemm(plan, x, w, out, act_out=ACT_SILU, out_mode=OUT_PIXEL_SHUFFLE, label=name)
        return out

    def _upsample_nearest_conv(self, plan, x: Act, mod: nn.Sequential, name: str) -> Act:
        """`Upsample` (ip.py:595-601, pixel_shuffle_upsample=False): nearest x2 as four strided row copies (one per output parity; a
        "batch" of the copy is one input row), then the 3x3 conv."""
        R, H, Wd, C = self.R, x.H, x.W, x.C
        assert x.ld == C and x.bs == H * Wd * C
        up = self.new(R, 2 * H, 2 * Wd, C)
        for dy in range(2):
            for dx in range(2):
                ops.rows_copy(plan, x.t, up.t, B=R * H, rows=Wd, C=C, src_bs=Wd * C, src_rs=C, dst_bs=4 * Wd * C, dst_rs=2 * C,
                              src_off=x.off, dst_off=(dy * 2 * Wd + dx) * C, label=f"{name}.nearest{dy}{dx}")
        w = self.W.conv(name + ".conv", mod[1])
        out = self.new(R, 2 * H, 2 * Wd, w.Cout)
        out.ssq = self.f32buf(out.rows)
        if not ops.igemm(plan, up, w, out, ssq_out=out.ssq, label=name + ".conv").ssq_emitted:
            out.ssq = None
        return out

    # ------------------------------------------------------------------------------------------ conditioning K/V
    def _ctx_weights(self):
        """Batched projections of the conditioning tokens for every attention site:
        self-attention `to_context` = LayerNorm(affine) + Linear (ip.py:527): the LN affine is folded into the Linear, so all
        sites share one normalised input; cross-attention `to_kv` (ip.py:783) consumes c directly."""
        selfs = [s for s in self.attn_sites if s["kind"] == "self" and s["n_ctx"] > 0]
        crosses = [s for s in self.attn_sites if s["kind"] == "cross"]

        def make_self():
            ws, bs = [], []
            for s in selfs:
                ln, lin = s["mod"].to_context[0], s["mod"].to_context[1]
                w = lin.weight.detach().float()
                ws.append(w * ln.weight.detach().float()[None, :])
                bs.append(lin.bias.detach().float() + w @ ln.bias.detach().float())
            return ops.pack_weight(torch.cat(ws), torch.cat(bs), self.dev, split=bool(SPLIT_STATIC))

        def make_cross():
            return ops.pack_weight(torch.cat([s["mod"].to_kv.weight.detach().float() for s in crosses]), None, self.dev, split=bool(SPLIT_STATIC))

        ws = self.W.get("ctx.self", make_self) if selfs else None
        wc = self.W.get("ctx.cross", make_cross) if crosses else None
        return selfs, crosses, ws, wc

    def _emit_context_kv(self, plan, c_rows: Act, rows_per_batch: int, k_row0_self: int, k_row0_cross: int, tag: str):
        """Project `c_rows` ([1,1,R*rpb,cond], already norm_cond'ed) for every site and write its K^/V^T rows."""
        R, W = self.R, self.W
        selfs, crosses, ws, wc = self._ctx_weights()
        n = rows_per_batch
        jobs = []   # one K^/V^T job per site, all run by a single launch after the two projections
        proj = {}   # the projections' output buffers (the per-step table of the sampler refills them: enable_time_table)
        if selfs:
            mu, rs = self.f32buf(R * n), self.f32buf(R * n)
            ops.rowstat(plan, c_rows, mode=1, rs=rs, mu=mu, eps=1e-5, label=f"ctx.{tag}.ln")
            st = self.new(1, 1, R * n, ws.Cout)
            ops.igemm(plan, c_rows, ws, st, mu=mu, rs=rs, label=f"ctx.{tag}.self")
            proj["self"] = st
            col = 0   # to_context of site i yields (k | v) = 2 * dim_head columns
            for s in selfs:
                d_ = s["dh"]
                ops.kv_prep(plan, st.t, st.t, W.f32(s["name"] + ".k_scale", lambda s=s: s["mod"].k_scale), s["khat"], s["vt"], B=R, heads=1,
                            rows=n, r0=k_row0_self, src_strides=(n * ws.Cout, ws.Cout, 0), k_strides=s["k_strides"], vt_strides=s["vt_strides"],
                            k_off=col, v_off=col + d_, batch=jobs, head_dim=d_)
                col += 2 * d_
            assert col == ws.Cout
        if crosses:
            st = self.new(1, 1, R * n, wc.Cout)
            ops.igemm(plan, c_rows, wc, st, label=f"ctx.{tag}.cross")
            proj["cross"] = st
            col = 0  # sites differ in head count / head dim (the mid blocks are always 8 x 64, ip.py:1380-1382): cumulative column offsets
            for s in crosses:
                d_ = s["dh"]
                inner = s["heads"] * d_
                ops.kv_prep(plan, st.t, st.t, W.f32(s["name"] + ".k_scale", lambda s=s: s["mod"].k_scale), s["khat"], s["vt"], B=R,
                            heads=s["heads"], rows=n, r0=k_row0_cross, src_strides=(n * wc.Cout, wc.Cout, d_), k_strides=s["k_strides"],
                            vt_strides=s["vt_strides"], k_off=col, v_off=col + inner, batch=jobs, head_dim=d_)
                col += 2 * inner
            assert col == wc.Cout
        if jobs:
            ops.kv_prep_multi(plan, jobs, self.dev, label=f"kv_ctx.{tag}")
        return proj

    def _emit_null_kv(self, plan):
        """learned null key/value (ip.py:545-547 self: after the context; ip.py:805-808 cross: first)."""
        R, W = self.R, self.W
        for s in self.attn_sites:
            nk = W.f32(s["name"] + ".null_kv", lambda s=s: s["mod"].null_kv)
            r0 = s["n_ctx"] if s["kind"] == "self" else 0
            ops.kv_prep(plan, nk, nk, W.f32(s["name"] + ".k_scale", lambda s=s: s["mod"].k_scale), s["khat"], s["vt"], B=R, heads=s["heads"],
                        rows=1, r0=r0, src_strides=(0, 0, 0), k_strides=s["k_strides"], vt_strides=s["vt_strides"], k_off=0, v_off=s["dh"],
                        label=s["name"] + ".kv_null", head_dim=s["dh"])

    # ------------------------------------------------------------------------------------------ static plan
    def _build_static_plan(self, n_tok: int):
        """Timestep-invariant conditioning (ip.py:1583-1660 minus the time tokens)."""
        u, R, W, cd, Tc = self.unet, self.R, self.W, self.cond_dim, self.Tc
        plan = Plan("unet-static")
        L = u.max_text_len
        src_batch = self.src_batch
        te16 = mask_u8 = None
        t_parts: List[Act] = []
        static_tokens: List[Act] = []   # pieces of c after the time tokens: [lowres time tokens][text tokens]
        # ---- low-res noise-level conditioning (ip.py:1583-1589)
        if self.lowres:
            hid_l = self.new(1, 1, R, Tc)
            ops.time_embed(plan, times=self.lowres_times, coef=None, step_ptr=None,
                           freqs=W.f32("ltime.freqs", lambda: u.to_lowres_time_hiddens[0].weights),
                           w=W.f32("ltime.w", lambda: u.to_lowres_time_hiddens[1].weight),
                           bias=W.f32("ltime.b", lambda: u.to_lowres_time_hiddens[1].bias), hid=hid_l, label="lowres_time_embed")
            t_l = self.new(1, 1, R, Tc)
            ops.igemm(plan, hid_l, W.conv("ltime.cond", u.to_lowres_time_cond[0], split=SPLIT_STATIC), t_l, label="to_lowres_time_cond")
            t_parts.append(t_l)
            tok_l = self.new(1, 1, R, self.ntt * cd)
            ops.igemm(plan, hid_l, W.conv("ltime.tokens", u.to_lowres_time_tokens[0], split=SPLIT_STATIC), tok_l, label="to_lowres_time_tokens")
            static_tokens.append(Act(tok_l.t, R, 1, self.ntt, cd, cd, self.ntt * cd))
        # ---- text conditioning (ip.py:1595-1652)
        if self.has_text:
            assert n_tok > 0
            ted = u.text_to_cond.weight.shape[1]
            te16 = torch.zeros(src_batch, n_tok, ted, dtype=torch.float16, device=self.dev)
            mask_u8 = torch.ones(src_batch, L, dtype=torch.uint8, device=self.dev)
            tok = self.new(src_batch, 1, L, cd, zero=True)          # rows >= n_tok stay zero (F.pad, ip.py:1617)
            te = Act(te16, src_batch, 1, n_tok, ted, ted, n_tok * ted)
            tok_head = Act(tok.t, src_batch, 1, n_tok, cd, cd, L * cd)
            ops.igemm(plan, te, W.conv("text_to_cond", u.text_to_cond, split=SPLIT_STATIC), tok_head, label="text_to_cond")
            xt = self.new(R, 1, L, cd)
            ops.select_rows(plan, tok.t, W.f16("null_text_embed", lambda: u.null_text_embed[0]), mask_u8, self.src_idx, self.keep_u8, xt.t,
                            R=R, L=L, C=cd, label="text_keep_select")
            text_tokens = self._perceiver(plan, xt, u.attn_pool) if u.attn_pool is not None else xt
            static_tokens.append(text_tokens)
            pooled = self.new(1, 1, R, cd)
            ops.mean_rows(plan, text_tokens, pooled, label="text_mean_pool")
            nc = u.to_text_non_attn_cond
            mu, rs = self.f32buf(R), self.f32buf(R)
            ops.rowstat(plan, pooled, mode=1, rs=rs, mu=mu, eps=1e-5, label="text_hidden.ln")
            w1 = W.conv("text_hidden.w1", nc[1], split=SPLIT_STATIC)
            h1 = self.new(1, 1, R, Tc)
            ops.igemm(plan, pooled, w1, h1, mu=mu, rs=rs, pa=W.f32("text_hidden.lnw", lambda: _pad_vec(nc[0].weight, w1.Cin_pad)),
                      ps=W.f32("text_hidden.lnb", lambda: _pad_vec(nc[0].bias, w1.Cin_pad)), act_out=ACT_SILU, label="text_hidden.lin1")
            h2 = self.new(1, 1, R, Tc)
            ops.igemm(plan, h1, W.conv("text_hidden.w2", nc[3], split=SPLIT_STATIC), h2, label="text_hidden.lin2")
            th = self.new(1, 1, R, Tc)
            ops.select_rows(plan, h2.t, W.f16("null_text_hidden", lambda: u.null_text_hidden), None, self.arange_idx, self.keep_u8, th.t,
                            R=R, L=1, C=Tc, label="text_hidden_select")
            t_parts.append(th)
        # ---- t_const = text hiddens + lowres t  (added to to_time_cond(hid) each step, ip.py:1588, 1652)
        if len(t_parts) == 2:
            ops.gate_residual(plan, t_parts[0], None, t_parts[1], self.t_const, label="t_const")
        elif len(t_parts) == 1:
            ops.rows_copy(plan, t_parts[0].t, self.t_const.t, B=1, rows=R, C=Tc, src_bs=0, src_rs=Tc, dst_bs=0, dst_rs=Tc, label="t_const")
        else:
            ops.memset32(plan, self.t_const.t.view(torch.int32), 0, label="t_const=0")
        # ---- static part of c: norm_cond per token (ip.py:1656-1660), then K/V rows for every attention site
        ns = sum(a.W * a.H for a in static_tokens)
        assert ns == self.NS, f"static conditioning tokens: built {ns}, planned {self.NS}"
        if ns > 0:
            c_static = self.new(R, 1, ns, cd)
            r0 = 0
            for a in static_tokens:
                n = a.H * a.W
                dst = Act(c_static.t, R, 1, n, cd, cd, ns * cd, off=r0 * cd)
                ops.ln_residual(plan, Act(a.t, R, 1, n, cd, a.ld, a.bs, a.off), W.f32("norm_cond.w", lambda: u.norm_cond.weight), dst,
                                beta=W.f32("norm_cond.b", lambda: u.norm_cond.bias), eps=1e-5, label="norm_cond(static)")
                r0 += n
            flat = Act(c_static.t, 1, 1, R * ns, cd, cd, R * ns * cd)
            self._emit_context_kv(plan, flat, rows_per_batch=ns, k_row0_self=self.ntt, k_row0_cross=1 + self.ntt, tag="static")
        self._emit_null_kv(plan)
        return plan, te16, mask_u8

    def _perceiver(self, plan, xt: Act, pr) -> Act:
        """PerceiverResampler (ip.py:447-498), depth x [PerceiverAttention (ip.py:408-445) + FeedForward(mult 4)]."""
        R, W, cd = self.R, self.W, self.cond_dim
        L = xt.W
        nl, nm = pr.latents.shape[0], pr.num_latents_mean_pooled
        NL = nl + nm
        # x + positional embedding
        posb = self.new(R, 1, L, cd)
        ops.rows_copy(plan, W.f16("attn_pool.pos", lambda: pr.pos_emb.weight[:L]), posb.t, B=R, rows=L, C=cd, src_bs=0, src_rs=cd,
                      dst_bs=L * cd, dst_rs=cd, label="attn_pool.pos_bcast")
        xpos = self.new(R, 1, L, cd)
        flat = lambda a: Act(a.t, 1, 1, a.B * a.H * a.W, a.C, a.ld, a.B * a.bs, a.off)
        ops.gate_residual(plan, flat(xt), None, flat(posb), flat(xpos), label="attn_pool.x_plus_pos")
        lat = self.new(R, 1, NL, cd)
        if nm > 0:
            pooled = self.new(1, 1, R, cd)
            ops.mean_rows(plan, xt, pooled, label="attn_pool.mean_pool")
            seq = pr.to_latents_from_mean_pooled_seq
            mu, rs = self.f32buf(R), self.f32buf(R)
            ops.rowstat(plan, pooled, mode=1, rs=rs, mu=mu, eps=1e-5, label="attn_pool.mp.ln")
            wm = W.conv("attn_pool.mp", seq[1], split=SPLIT_STATIC)
            dst = Act(lat.t, R, 1, 1, nm * cd, NL * cd, NL * cd)  # (R, nm*cd) written as the first nm rows of each row's latents
            src = Act(pooled.t, R, 1, 1, cd, cd, cd)
            ops.igemm(plan, src, wm, dst, mu=mu, rs=rs, pa=W.f32("attn_pool.mp.g", lambda: _pad_vec(seq[0].g, wm.Cin_pad)), label="attn_pool.mp.lin")
        ops.rows_copy(plan, W.f16("attn_pool.latents", lambda: pr.latents), lat.t, B=R, rows=nl, C=cd, src_bs=0, src_rs=cd, dst_bs=NL * cd,
                      dst_rs=cd, dst_off=nm * cd, label="attn_pool.latents")
        Jk = L + NL
        Jp = ops._round_up(Jk, 32)
        for i, (pa, ff) in enumerate(pr.layers):
            nm_ = f"attn_pool.layers.{i}"
            heads, dh = pa.heads, pa.dim_head
            inner = heads * dh
            kv = self.new(R, 1, Jk, 2 * inner)
            wkv = W.conv(nm_ + ".to_kv", pa.to_kv, split=SPLIT_STATIC)
            # k/v of the normalised sequence ...
            mu, rs = self.f32buf(R * L), self.f32buf(R * L)
            ops.rowstat(plan, xpos, mode=1, rs=rs, mu=mu, eps=1e-5, label=nm_ + ".norm")
            ops.igemm(plan, xpos, wkv, Act(kv.t, R, 1, L, 2 * inner, 2 * inner, Jk * 2 * inner), mu=mu, rs=rs,
                      pa=W.f32(nm_ + ".norm.w", lambda pa=pa: _pad_vec(pa.norm.weight, wkv.Cin_pad)),
                      ps=W.f32(nm_ + ".norm.b", lambda pa=pa: _pad_vec(pa.norm.bias, wkv.Cin_pad)), label=nm_ + ".kv_x")
            # ... and of the normalised latents (ip.py:417-418), which also give q
            mul, rsl = self.f32buf(R * NL), self.f32buf(R * NL)
            ops.rowstat(plan, lat, mode=1, rs=rsl, mu=mul, eps=1e-5, label=nm_ + ".norm_latents")
            lnw = W.f32(nm_ + ".norml.w", lambda pa=pa: _pad_vec(pa.norm_latents.weight, wkv.Cin_pad))
            lnb = W.f32(nm_ + ".norml.b", lambda pa=pa: _pad_vec(pa.norm_latents.bias, wkv.Cin_pad))
            ops.igemm(plan, lat, wkv, Act(kv.t, R, 1, NL, 2 * inner, 2 * inner, Jk * 2 * inner, off=L * 2 * inner), mu=mul, rs=rsl, pa=lnw, ps=lnb,
                      label=nm_ + ".kv_lat")
            q = self.new(R, 1, NL, inner)
            ops.igemm(plan, lat, W.conv(nm_ + ".to_q", pa.to_q, split=SPLIT_STATIC), q, mu=mul, rs=rsl, pa=lnw, ps=lnb, label=nm_ + ".to_q")
            khat = torch.zeros(R, heads, Jp, dh, dtype=torch.float16, device=self.dev)
            vt = torch.zeros(R, heads, dh, Jp, dtype=torch.float16, device=self.dev)
            ks, vs = (heads * Jp * dh, Jp * dh, dh), (heads * dh * Jp, dh * Jp, Jp)
            ops.kv_prep(plan, kv.t, kv.t, W.f32(nm_ + ".k_scale", lambda pa=pa: pa.k_scale), khat, vt, B=R, heads=heads, rows=Jk, r0=0,
                        src_strides=(Jk * 2 * inner, 2 * inner, dh), k_strides=ks, vt_strides=vs, k_off=0, v_off=inner, label=nm_ + ".kv_prep",
                        head_dim=dh)
            o = self.new(R, 1, NL, inner)
            ops.attention(plan, q.t, khat, vt, o.t, B=R, heads=heads, rows=NL, J=Jk, q_strides=(NL * inner, dh, inner), k_strides=ks,
                          vt_strides=vs, o_strides=(NL * inner, dh, inner), q_scale=W.f32(nm_ + ".q_scale", lambda pa=pa: pa.q_scale),
                          q_mult=SIM_SCALE * LOG2E, label=nm_ + ".attn", head_dim=dh,
                          logit_bound=ops.attention_logit_bound(pa.q_scale, pa.k_scale, SIM_SCALE * LOG2E))
            y = self.new(R, 1, NL, cd)
            ops.igemm(plan, o, W.conv(nm_ + ".to_out", pa.to_out[0], split=SPLIT_STATIC), y, label=nm_ + ".to_out")
            lat2 = self.new(R, 1, NL, cd)
            ops.ln_residual(plan, y, W.f32(nm_ + ".out.w", lambda pa=pa: pa.to_out[1].weight), lat2,
                            beta=W.f32(nm_ + ".out.b", lambda pa=pa: pa.to_out[1].bias), res=lat, eps=1e-5, label=nm_ + ".out_norm")
            lat = self._feed_forward(plan, lat2, ff, nm_ + ".ff", split=SPLIT_STATIC)
        return lat

    # ------------------------------------------------------------------------------------------ run-time API
    def set_conditioning(self, *, text_embeds, text_mask, keep, lowres_noise_times):
        """Stage the timestep-invariant inputs and run the static plan.  `keep`: bool [R] (True = conditional row)."""
        R, u, src_batch = self.R, self.unet, self.src_batch
        n_tok = 0
        if self.has_text:
            assert text_embeds is not None, "this engine was planned with text conditioning"
            text_embeds = text_embeds[:, : u.max_text_len]
            n_tok = text_embeds.shape[1]
            assert text_embeds.shape[0] == src_batch
        if n_tok not in self._static_plans:
            self._static_plans[n_tok] = self._build_static_plan(n_tok)
        plan, te16, mask_u8 = self._static_plans[n_tok]
        self.keep_u8.copy_(keep.to(torch.uint8))
        self.src_idx.copy_(torch.arange(R, dtype=torch.int32) % src_batch)
        if self.lowres:
            assert lowres_noise_times is not None
            lt = lowres_noise_times.float().reshape(-1)
            self.lowres_times.copy_(lt.repeat(R // lt.numel()))
        if self.has_text:
            te16.copy_(text_embeds.to(torch.float16))
            L = u.max_text_len
            m = torch.zeros(src_batch, L, dtype=torch.uint8)
            if text_mask is not None:
                tm = text_mask[:, :L].to(torch.uint8).cpu()
                m[:, : tm.shape[1]] = tm
            else:
                m[:, :] = 1      # no mask: the zero-padded positions stay zero tokens, they are NOT replaced by null_text_embed (ip.py:1619-1632)
            mask_u8.copy_(m)
        if not self.dry:
            plan.run()
            if self._tt_plan is not None:
                self._tt_plan.run()
        self._cond_ready = True

    # labels of the launches of the step plan that depend on the timestep and the conditioning only (not on x_t); the planner records the
    # params structs themselves while it emits them (self._time_chain_ops): the fast plan drops exactly those, whatever a later op is called
    _TIME_CHAIN = ("time_embed", "to_time_cond", "to_time_tokens", "norm_cond(time)", "time_mlps", "scale_shift", "ctx.dyn.ln", "ctx.dyn.self",
                   "ctx.dyn.cross")

    def enable_time_table(self, coef: torch.Tensor, step_ptr: torch.Tensor) -> Optional[Plan]:
        """Sampler mode: evaluate the timestep-only part of the denoiser — time embedding -> time conditioning / time tokens -> every
        ResnetBlock's scale / shift, the time tokens' K / V projections of every attention site (nine launches of ~8 us each at the head of
        every step, 1 workgroup each) — for ALL rows of the sampler's coefficient table in one batched pass per request (`_tt_plan`, run by
        set_conditioning), and return the step plan in which one STEP_SLICE launch copies the current step's rows instead.  The batched
        pass is the same kernels on `rows * steps` rows; the results are what the per-step launches compute."""
        R, W, u = self.R, self.W, self.unet
        NR = coef.shape[0]
        if self._tt_plan is not None:
            return self.step_plan_tt
        rows_all = NR * R
        # footprint of the tables and of the batched pass's intermediates, per stage and lane (fp32 scale / shift rows, fp16 hiddens, time
        # tokens and K / V projections).  Above the cap (large batches, long schedules, small-memory parts) the per-step chain stays: same
        # results, nine small launches per step.
        selfs_, crosses_, ws_, wc_ = self._ctx_weights()
        proj_c = (ws_.Cout if selfs_ else 0) + (wc_.Cout if crosses_ else 0)
        chain_bytes = (8 * self.Tc + 8 * self.total_c) if TIME_CHAIN_F32 else 2 * (3 * self.Tc + 2 * self.total_c)   # hid, t_const, t, ss rows (fp32 t / ss: LINEAR_F32)
        # per row of the coefficient table: what STAYS for the life of the request (the four tables STEP_SLICE reads) and what the batched pass needs
        # while it runs (hiddens, time tokens, the time MLPs' output — as large as both scale / shift tables together).  The pass runs in chunks of
        # steps over ONE set of intermediates sized for a chunk (round 6, 1000 steps x 16 rows: BASELINE C2 3.98 -> 2.42 GiB per stage and lane in 8 chunks, README unet2 1.88 -> 1.24 in 4, unet1 1.44 -> 1.20 in 2)
        table_row = 2 * 4 * self.total_c + self.ntt * 2 * proj_c
        work_row = chain_bytes + self.ntt * (2 * 2 * self.cond_dim + 8)
        nchunks = max(1, min(NR, -(-rows_all * work_row // TIME_TABLE_CHUNK_BYTES)))
        steps_c = -(-NR // nchunks)                 # steps per chunk
        nchunks = -(-NR // steps_c)
        nmax = steps_c * R
        tt_bytes = rows_all * table_row + nmax * work_row
        self.time_table_bytes = tt_bytes
        self.time_table_layout = dict(chunks=nchunks, steps_per_chunk=steps_c, work_bytes_unchunked=rows_all * work_row, table_bytes=rows_all * table_row)
        if tt_bytes > TIME_TABLE_MAX_BYTES:
            return None
        tt = Plan("unet-time-table")
        times_all = coef[:, 6].to(self.dev).float().repeat_interleave(R).contiguous()       # the log-SNR every step's time_embed reads
        pre = lambda a, n: Act(a.t, 1, 1, n, a.C, a.ld, n * a.ld, a.off)                      # the first n rows of a row tensor
        rows_of = lambda a, r0, n: Act(a.t, 1, 1, n, a.C, a.ld, n * a.ld, a.off + r0 * a.ld)  # rows [r0, r0 + n)
        tc_all = self.new(1, 1, nmax, self.Tc)               # the conditioning rows of one step, repeated: the same for every chunk
        ops.rows_copy(tt, self.t_const.t, tc_all.t, B=steps_c, rows=R, C=self.Tc, src_bs=0, src_rs=self.Tc, dst_bs=R * self.Tc, dst_rs=self.Tc,
                      label="tt.t_const")
        hid = self.new(1, 1, nmax, self.Tc)
        t_all = self.f32buf(nmax, self.Tc) if TIME_CHAIN_F32 else self.new(1, 1, nmax, self.Tc)
        tok_raw = self.new(1, 1, nmax, self.ntt * self.cond_dim)
        c_time = self.new(1, 1, nmax * self.ntt, self.cond_dim)
        tw, tb, gam, isc, ish, _, total_c = W.get("timemlp.tables", lambda: self._time_mlp_tables(self._all_resnet_blocks()))
        ss = self.f32buf(nmax, tw.shape[0]) if TIME_CHAIN_F32 else self.new(1, 1, nmax, tw.shape[0])
        tab_pa, tab_ps = self.f32buf(rows_all, total_c), self.f32buf(rows_all, total_c)
        segments = [(tab_pa, self.pa2), (tab_ps, self.ps2)]
        selfs, crosses, ws, wc = selfs_, crosses_, ws_, wc_
        n_tok_rows = rows_all * self.ntt
        tab_self = tab_cross = mu = rs = None
        if selfs:
            mu, rs = self.f32buf(nmax * self.ntt), self.f32buf(nmax * self.ntt)
            tab_self = self.new(1, 1, n_tok_rows, ws.Cout)
            segments.append((tab_self.t, self._dyn_proj["self"].t))
        if crosses:
            tab_cross = self.new(1, 1, n_tok_rows, wc.Cout)
            segments.append((tab_cross.t, self._dyn_proj["cross"].t))
        for ch in range(nchunks):
            r0 = ch * nmax
            n = min(nmax, rows_all - r0)
            tag = "tt." if nchunks == 1 else f"tt{ch}."
            ops.time_embed(tt, times=times_all[r0:r0 + n], coef=None, step_ptr=None, freqs=W.f32("time.freqs", lambda: u.to_time_hiddens[0].weights),
                           w=W.f32("time.w", lambda: u.to_time_hiddens[1].weight), bias=W.f32("time.b", lambda: u.to_time_hiddens[1].bias), hid=pre(hid, n),
                           label=tag + "time_embed")
            if TIME_CHAIN_F32:
                ops.linear_f32(tt, pre(hid, n), W.f32("time.cond.wt32", lambda: u.to_time_cond[0].weight.t()), W.f32("time.cond.b32", lambda: u.to_time_cond[0].bias),
                               t_all[:n], res=pre(tc_all, n), label=tag + "to_time_cond")
            else:
                ops.igemm(tt, pre(hid, n), W.conv("time.cond", u.to_time_cond[0], split=SPLIT_STATIC), pre(t_all, n), res=pre(tc_all, n), label=tag + "to_time_cond")
            ops.igemm(tt, pre(hid, n), W.conv("time.tokens", u.to_time_tokens[0], split=SPLIT_STATIC), pre(tok_raw, n), label=tag + "to_time_tokens")
            tok_rows = Act(tok_raw.t, 1, 1, n * self.ntt, self.cond_dim, self.cond_dim, n * self.ntt * self.cond_dim)
            ct = pre(c_time, n * self.ntt)
            ops.ln_residual(tt, tok_rows, W.f32("norm_cond.w", lambda: u.norm_cond.weight), ct, beta=W.f32("norm_cond.b", lambda: u.norm_cond.bias),
                            eps=1e-5, label=tag + "norm_cond(time)")
            if TIME_CHAIN_F32:
                ops.linear_f32(tt, t_all[:n], W.f32("timemlp.wt32", lambda: tw.t()), W.f32("timemlp.b32", lambda: tb), ss[:n], act_in=ACT_SILU, label=tag + "time_mlps")
                ss_c = ss[:n]
            else:
                ops.igemm(tt, pre(t_all, n), W.raw("timemlp.w", tw, tb, split=SPLIT_STATIC), pre(ss, n), act_in=ACT_SILU, label=tag + "time_mlps")
                ss_c = pre(ss, n)
            ops.scale_shift(tt, ss_c, W.f32("timemlp.gam", lambda: gam), W.get("timemlp.isc", lambda: isc.to(self.dev)),
                            W.get("timemlp.ish", lambda: ish.to(self.dev)), tab_pa[r0:r0 + n], tab_ps[r0:r0 + n], label=tag + "scale_shift")
            if selfs:
                ops.rowstat(tt, ct, mode=1, rs=rs[:n * self.ntt], mu=mu[:n * self.ntt], eps=1e-5, label=tag + "ctx.ln")
                ops.igemm(tt, ct, ws, rows_of(tab_self, r0 * self.ntt, n * self.ntt), mu=mu[:n * self.ntt], rs=rs[:n * self.ntt], label=tag + "ctx.self")
            if crosses:
                ops.igemm(tt, ct, wc, rows_of(tab_cross, r0 * self.ntt, n * self.ntt), label=tag + "ctx.cross")
        tt.keep += [tab_pa, tab_ps, times_all, hid, t_all, tok_raw, c_time, ss, tc_all.t, mu, rs] + ([tab_self.t] if selfs else []) + ([tab_cross.t] if crosses else [])
        # the step plan with the chain replaced by the copy of the current step's rows
        one = Plan("slice")
        ops.step_slice(one, segments, step_ptr, label="time_table_rows")
        fast = Plan("unet-step-tt")
        placed = False
        chain = {id(st) for st in self._time_chain_ops}
        assert NR == coef.shape[0] and len(chain) == sum(l in self._TIME_CHAIN for _, _, l in self.step_plan.ops), \
            "the recorded timestep chain and the labelled one differ"
        for kind, st, label in self.step_plan.ops:
            if id(st) in chain:
                if not placed:
                    fast.ops.append(one.ops[0])
                    placed = True
                continue
            fast.ops.append((kind, st, label))
        assert placed and len(fast.ops) == len(self.step_plan.ops) - len(chain) + 1
        fast.keep = list(self.step_plan.keep) + list(one.keep) + [times_all]
        self.step_plan_tt = fast
        self._tt_plan = tt
        if self._cond_ready and not self.dry:
            tt.run()
        return fast

    def bind_step_counter(self, coef: torch.Tensor, step_ptr: torch.Tensor):
        """Sampler mode: the time embedding reads log-SNR of the current step from the coef table (graph replay)."""
        p = self._time_embed_op
        p.times, p.coef, p.step_ptr = None, coef.data_ptr(), step_ptr.data_ptr()
        self.coef, self.step_ptr = coef, step_ptr

    def set_cond_images(self, cond_images: torch.Tensor):
        """The conditioning image of the following forward / sampling calls: (src_batch, cond_images_channels, h, w), values as given
        (the reference does not normalise it); nearest-resized to this engine's resolution (ip.py:1558-1559) and packed to fp16 NHWC."""
        assert self.cond_in is not None, 'this unet was built without cond_images_channels'
        assert cond_images.shape[1] == self.cond_in.shape[1], \
            'the number of channels on the conditioning image you are passing in does not match what you specified on initialiation of the unet'
        assert cond_images.shape[0] == self.src_batch, f'cond_images batch {cond_images.shape[0]} != {self.src_batch}'
        ci = cond_images.to(self.dev).float()
        if ci.shape[-2:] != self.cond_in.shape[-2:]:
            ci = torch.nn.functional.interpolate(ci, self.cond_in.shape[-1], mode=getattr(self.unet, 'resize_mode', 'nearest'))   # ip.py:1559
        self.cond_in.copy_(ci)
        if self._self_cond_op is not None:
            return                                   # packed together with the self-conditioning input at the start of every step
        if self._cond_pack is None:
            self._cond_pack = Plan("cond-image")
            ops.pack_image(self._cond_pack, self.cond_in, None, self.cimg, brep=self.R // self.src_batch, label="pack_cond_image")
        if not self.dry:
            self._cond_pack.run()

    def set_self_cond(self, self_cond: Optional[torch.Tensor]):
        """Plain-forward mode: the self-conditioning image of the next forward() (None = zeros, ip.py:1542)."""
        assert self.self_cond_in is not None, 'this unet was built without self_cond'
        if self_cond is None:
            self.self_cond_in.zero_()
        else:
            self.self_cond_in.copy_(self_cond)

    def bind_self_cond(self, x0_thr: torch.Tensor):
        """Sampler mode: every step self-conditions on `x0_thr`, the buffer DDPM_UPDATE leaves the previous step's thresholded x0 in
        (ip.py:2249: `self_cond = x_start if unet.self_cond else None`)."""
        assert self._self_cond_op is not None and x0_thr.numel() == self.self_cond_in.numel() and x0_thr.dtype == torch.float32
        self._self_cond_op.a = x0_thr.data_ptr()
        self._self_cond_src = x0_thr

    def forward(self, x: torch.Tensor, time: torch.Tensor, lowres_cond_img: Optional[torch.Tensor] = None) -> torch.Tensor:
        assert self._cond_ready, "set_conditioning() first"
        R = self.R
        self.x_in.copy_(x)
        if self.lowres:
            self.lowres_in.copy_(lowres_cond_img)
        t = time.float().reshape(-1)
        self.times.copy_(t.repeat(R // t.numel()))
        if not self.dry:
            self.step_plan.run()
        return self.out
