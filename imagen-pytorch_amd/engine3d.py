"""UnetEngine3D — kernel planner of the drop-in `Unet3D` (Imagen-Video denoiser, imagen_pytorch/imagen_video.py:1650-1941 = iv.py).

The two kernels it needs beyond the image path live in csrc/temporal.hip.  Checked on CPU against oracle/unet3d_oracle.py through
tests/plan_interp.py and on MI355X by tests/test_video_gpu.py (DESIGN.md §8 NEXT-2).

Layout: a clip is fp16 [R, F, H, W, C] — F consecutive NHWC frames per row — so the same memory serves three views:
  frames  Act(R*F, H, W, C)     per-frame ops of the image path (3x3 conv, down / up-sampling, CrossEmbed) with batch R*F
  clip    Act(R, F*H, W, C)     per-clip ops: GlobalContext, cross attention and space-time attention over all F*H*W tokens,
                                1x1 convolutions whose gate / affine is per clip
  time    Act(R, F, H*W, C)     ops along the frame axis: "image" rows = frames, columns = pixels

The pseudo-3D convolution (iv.py:397-451) = the image path's fused ChanRMSNorm -> SiLU -> 3x3 conv per frame, followed by the causal
temporal conv1d (k = 3).  The temporal conv is three accumulating 1x1 GEMMs over frame-shifted views of the time layout
(y[f] = W2 x[f] + b;  y[1:] += W1 x[:-1];  y[2:] += W0 x[:-2]): only IGEMM features the image path already uses.  That costs three
launches and ~2x the minimal traffic per temporal conv; a dedicated (3 x 1)-tap staging path is the obvious next optimisation.
Everything input-independent is prepared at plan build: the relative position bias table of every temporal attention
(DynamicPositionBias MLP on the 2F-1 frame distances, iv.py:1182-1223) and the depthwise PEG taps.

Prompt frames (`cond_video_frames` / `post_cond_video_frames`, iv.py:1682-1718, 1933-1939): the network then runs on
F = len(post) + len(pre) + Fx frames per clip — the reference concatenates BOTH prompts in front of the Fx frames being denoised —
while the sampler state `x_in` / `out` keep Fx frames.  The prompt slots of the packed input clip are static (written once by
set_cond_video_frames); per step the Fx packed frames are copied behind them and the final conv's F output frames are cut back to
frames [len(pre), len(pre) + Fx) — two strided row copies around the unchanged plan.
"""
from __future__ import annotations

import math
from types import SimpleNamespace
from typing import List, Optional

import torch
import torch.nn.functional as TF

from . import ops
from .engine import SIM_SCALE, UnetEngine, _pad_vec
from .modules3d import CrossEmbed3dP, Parallel3dP, PixelShuffleUpsample3dP, TransformerBlock3dP
from .ops import ACT_GELU, ACT_SILU, OUT_NCHW_F32, OUT_PIXEL_SHUFFLE, Act, Plan


def _w2d(mod):
    """View of a conv module whose weight carries singleton frame / spatial dims, as the (w [o, i, kh, kw], bias) pair the image-path
    weight packer expects: Conv3d (o, i, 1, kh, kw) and Conv1d (o, i, 1)."""
    w = mod.weight.detach().float()
    if w.ndim == 5:
        w = w[:, :, 0]
    elif w.ndim == 3:
        w = w[:, :, :, None]
    return SimpleNamespace(weight=w, bias=None if mod.bias is None else mod.bias.detach().float())


# Split-precision weights (engine.py, SPLIT_*) on the video path's OUTPUT stage: `final_conv` and block1 of `final_res_block` — the two convs
# whose weight rounding reaches the output unattenuated (plan interpreter, BASELINE C5: cond 1.008e-3 -> 0.970e-3, null 0.921e-3 -> 0.901e-3;
# block1 of the two outer down levels on top of them buys nothing).  A video clip puts R * F frames through every conv, so the image path's
# executed-FLOP bound (SPLIT_SMALL_FLOPS) never admits them: the output stage gets its own, wider, bound; two launches per step grow by
# ~20 us each (1 % of the C5 step).
SPLIT_OUTPUT_STAGE = 1
TEMPORAL_CONV_FUSED = 1   # (round 6) the causal temporal Conv1d as one (3 x 1)-tap launch (ABI 10: ImagenIgemmParams.pad_x1) instead of three 1x1 GEMMs
TIME_CHAIN_F32 = 0        # (round 6, call A) 1 = to_time_cond and the batched time MLPs on fp32 rows (IMAGEN_OP_LINEAR_F32) as the image planner runs them: priced in the
                          # plan interpreter at +-0.5 % on C5 (cond 0.970 -> 0.965e-3, null 0.901 -> 0.905e-3) and measured on MI355X at 9.67e-4 / 9.97e-4 against 9.64e-4 /
                          # 9.76e-4 without — no gain on the video denoiser, so its chain stays on the fp16 GEMMs
SPLIT_OUTPUT_MAX_K = 640          # taps * input channels of the unsplit weight (dim 64: 576)
SPLIT_OUTPUT_FLOPS = 4.0e10       # 2 * pixels * Cout * (2 K) of the split launch (C5: 1.9e10)


class UnetEngine3D(UnetEngine):

    def enable_time_table(self, coef, step_ptr):
        """(The per-request time table of the image engine is not built for the video plan: its step keeps the per-step chain.)"""
        return None

    def __init__(self, unet, rows: int, src_batch: int, frames: int, size: int, device, with_text: bool = True, ignore_time: bool = False,
                 dry: bool = False, pre_frames: int = 0, post_frames: int = 0):
        self.Fx, self.Fpre, self.Fpost = frames, pre_frames, post_frames   # frames of the sampler state / of the two prompts
        self.F = frames + pre_frames + post_frames                         # frames the network runs on
        self.ignore_time = ignore_time
        assert self.F <= 32, "the temporal attention kernel holds at most 32 frames per pixel"
        div = getattr(unet, 'total_temporal_divisor', 1)
        assert pre_frames % div == 0 and post_frames % div == 0, \
            f'the number of conditioning frames must be divisible by {div}'                     # iv.py:1700, 1713
        super().__init__(unet, rows, src_batch, size, device, with_text=with_text, dry=dry)

    # ------------------------------------------------------------------------------------------ views
    def frames(self, a: Act, F: int) -> Act:
        """clip / time view -> per-frame view."""
        n = a.B * a.H * a.W
        RF = self.R * F
        P = n // RF
        S = int(round(math.sqrt(P)))
        assert S * S == P and a.ld == a.C
        return Act(a.t, RF, S, S, a.C, a.C, P * a.C, a.off, ssq=a.ssq)

    def clip(self, a: Act, F: int) -> Act:
        assert a.ld == a.C and a.B == self.R * F
        return Act(a.t, self.R, F * a.H, a.W, a.C, a.C, F * a.H * a.W * a.C, a.off, ssq=a.ssq)

    def _alloc_io(self):
        super()._alloc_io()
        R, S, u, F, Fx = self.R, self.S, self.unet, self.F, self.Fx
        # frame-major fp32 images: (b, f, c, h, w); the sampler's state and the prediction it reads hold the Fx denoised frames only
        self.x_in = self.f32buf(self.src_batch, Fx, u.channels, S, S, zero=True)
        self.lowres_in = self.f32buf(self.src_batch, Fx, u.channels, S, S, zero=True) if self.lowres else None
        self.out = self.f32buf(R, Fx, u.channels_out, S, S)
        self.out_full = self.f32buf(R, F, u.channels_out, S, S) if F != Fx else self.out   # what final_conv writes
        # conditioning image (Unet3D(cond_images_channels=...), iv.py:1722-1731): ONE image per sample, repeated over every frame the
        # network runs on; the init conv reads it as a second, channel-concatenated input like the image Unet does
        cc = getattr(u, 'cond_images_channels', 0)
        self.cimg = self.new(R * F, S, S, (cc + 7) // 8 * 8, zero=True) if cc else None

    def set_cond_images(self, cond_images: torch.Tensor):
        """(src_batch, cond_images_channels, h, w), values as given: resized to this engine's resolution with the unet's resize_mode
        (resize_video_to after the repeat over frames, iv.py:1728-1729: the frames are identical, so one 2-D resize), packed to fp16
        NHWC and written to every frame of every row — static over the timesteps."""
        u, R, S, F = self.unet, self.R, self.S, self.F
        cc = u.cond_images_channels
        assert self.cimg is not None, 'this unet was built without cond_images_channels'
        assert cond_images.ndim == 4 and cond_images.shape[0] == self.src_batch and cond_images.shape[1] == cc, \
            'the number of channels on the conditioning image you are passing in does not match what you specified on initialiation of the unet'
        ci = cond_images.to(self.dev).float()
        if tuple(ci.shape[-2:]) != (S, S):
            ci = TF.interpolate(ci, S, mode=getattr(u, 'resize_mode', 'nearest'))
        packed = torch.zeros(self.src_batch, S, S, self.cimg.C, device=self.dev)
        packed[..., :cc] = ci.permute(0, 2, 3, 1)
        packed = packed.to(torch.float16).repeat(R // self.src_batch, 1, 1, 1)                       # rows: [cond..., null...]
        self.cimg.t.view(R, F, S, S, self.cimg.C).copy_(packed[:, None].expand(-1, F, -1, -1, -1))

    # ------------------------------------------------------------------------------------------ step plan
    def _build_step_plan(self) -> Plan:
        u, R, S, W, F = self.unet, self.R, self.S, self.W, self.F
        plan = Plan("unet3d-step")
        self._plan = plan
        it = self.ignore_time
        cin = u.channels * (2 if self.lowres else 1)
        assert cin <= 8, "init conv packs the input frames into 8 channels"
        Fx = self.Fx
        self.img = self.new(R * F, S, S, 8, zero=True)
        xin = self.x_in.view(self.src_batch * Fx, u.channels, S, S)
        lin = self.lowres_in.view(self.src_batch * Fx, u.channels, S, S) if self.lowres else None
        self.img_x = self.img if F == Fx else self.new(R * Fx, S, S, 8)       # the packed Fx frames [x | lowres | zero pad]
        self._pack_op = ops.pack_image(plan, xin, lin, self.img_x, brep=R // self.src_batch, label="pack_frames")
        plan.keep += [self.x_in, self.lowres_in] if self.lowres else [self.x_in]
        self.fin2 = None
        if F != Fx:
            # prompt frames: clip = [post | pre | x] (iv.py:1703, 1716); the prompt slots are static, the Fx frames land behind them
            fr = S * S * 8
            ops.rows_copy(plan, self.img_x.t, self.img.t, B=R, rows=1, C=Fx * fr, src_bs=Fx * fr, src_rs=0, dst_bs=F * fr, dst_rs=0,
                          dst_off=(self.Fpost + self.Fpre) * fr, label="place_frames")
            if self.lowres and self.Fpost:
                # final_conv reads the low-res clip extended as [pre | lowres | post] (iv.py:1687, 1691) — another frame order than
                # the input clip's once there are succeeding prompt frames: its own buffer, same channel slots as `img`
                self.fin2 = self.new(R * F, S, S, 8, zero=True)
                ops.rows_copy(plan, self.img_x.t, self.fin2.t, B=R, rows=1, C=Fx * fr, src_bs=Fx * fr, src_rs=0, dst_bs=F * fr, dst_rs=0,
                              dst_off=self.Fpre * fr, label="place_lowres_frames")

        # ---- time conditioning: identical to the image Unet (iv.py:1764-1781)
        self.hid = self.new(1, 1, R, self.Tc)
        self._time_embed_op = ops.time_embed(
            plan, times=self.times, coef=None, step_ptr=None,
            freqs=W.f32("time.freqs", lambda: u.to_time_hiddens[0].weights), w=W.f32("time.w", lambda: u.to_time_hiddens[1].weight),
            bias=W.f32("time.b", lambda: u.to_time_hiddens[1].bias), hid=self.hid, label="time_embed")
        # the timestep-conditioning chain in fp32 (engine.TIME_CHAIN_F32): these are PER-SAMPLE rows, so an fp16 rounding of them is a
        # bias of the whole clip, not noise that averages over pixels
        if TIME_CHAIN_F32:
            self.t = self.f32buf(R, self.Tc)
            ops.linear_f32(plan, self.hid, W.f32("time.cond.wt32", lambda: u.to_time_cond[0].weight.t()), W.f32("time.cond.b32", lambda: u.to_time_cond[0].bias),
                           self.t, res=self.t_const, label="to_time_cond")
        else:
            self.t = self.new(1, 1, R, self.Tc)
            ops.igemm(plan, self.hid, W.conv("time.cond", u.to_time_cond[0]), self.t, res=self.t_const, label="to_time_cond")
        tok_raw = self.new(1, 1, R, self.ntt * self.cond_dim)
        ops.igemm(plan, self.hid, W.conv("time.tokens", u.to_time_tokens[0]), tok_raw, label="to_time_tokens")
        self.c_time = self.new(1, 1, R * self.ntt, self.cond_dim)
        tok_rows = Act(tok_raw.t, 1, 1, R * self.ntt, self.cond_dim, self.cond_dim, R * self.ntt * self.cond_dim)
        ops.ln_residual(plan, tok_rows, W.f32("norm_cond.w", lambda: u.norm_cond.weight), self.c_time,
                        beta=W.f32("norm_cond.b", lambda: u.norm_cond.bias), eps=1e-5, label="norm_cond(time)")

        # ---- every ResnetBlock's time-MLP in one GEMM.  The per-frame convs index their affine by frame (batch R*F), so t is
        # first replicated over the frames of its clip (max over the levels' frame counts: temporal strides only shrink F)
        blocks = self._all_resnet_blocks()
        self._blk_index = {id(rb): i for i, rb in enumerate(blocks)}
        tw, tb, gam, isc, ish, self._blk_off, total_c = W.get("timemlp.tables", lambda: self._time_mlp_tables(blocks))
        self.total_c = total_c
        if TIME_CHAIN_F32:     # (fp32 rows copied as pairs of halves)
            t_rep = self.f32buf(R * F, self.Tc)
            ops.rows_copy(plan, self.t.view(torch.float16), t_rep.view(torch.float16), B=R, rows=F, C=2 * self.Tc, src_bs=2 * self.Tc, src_rs=0,
                          dst_bs=2 * F * self.Tc, dst_rs=2 * self.Tc, label="t_per_frame")
            ss = self.f32buf(R * F, tw.shape[0])
            ops.linear_f32(plan, t_rep, W.f32("timemlp.wt32", lambda: tw.t()), W.f32("timemlp.b32", lambda: tb), ss, act_in=ACT_SILU, label="time_mlps")
        else:
            t_rep = self.new(1, 1, R * F, self.Tc)
            ops.rows_copy(plan, self.t.t, t_rep.t, B=R, rows=F, C=self.Tc, src_bs=self.Tc, src_rs=0, dst_bs=F * self.Tc, dst_rs=self.Tc,
                          label="t_per_frame")
            ss = self.new(1, 1, R * F, tw.shape[0])
            ops.igemm(plan, t_rep, W.raw("timemlp.w", tw, tb), ss, act_in=ACT_SILU, label="time_mlps")
        self.pa2 = self.f32buf(R * F, total_c)
        self.ps2 = self.f32buf(R * F, total_c)
        ops.scale_shift(plan, ss, W.f32("timemlp.gam", lambda: gam), W.get("timemlp.isc", lambda: isc.to(self.dev)),
                        W.get("timemlp.ish", lambda: ish.to(self.dev)), self.pa2, self.ps2)
        self._kv_dynamic_anchor = len(plan.ops)

        # ---- traversal (iv.py:1751-1941)
        f = F                                         # current number of frames
        x = self._init_conv3d(plan)
        if not it:
            x = self._temporal_peg(plan, x, u.init_temporal_peg, "init_temporal_peg", f)
            x = self._temporal_attn(plan, x, u.init_temporal_attn, "init_temporal_attn", f)
        self.taps['init'] = x
        if u.init_resnet_block is not None:
            x = self._resnet3d(plan, x, None, u.init_resnet_block, "init_resnet", f, with_cond=False)
        hiddens: List[Act] = []
        strides = self.lc["temporal_strides"]
        n_levels = len(self.lc["in_out"])
        for i, lvl in enumerate(u.downs):
            pre, init_block, res_blocks, attn_block, peg, tattn, tdown, post = lvl
            if pre is not None:
                x = self._downsample3d(plan, x, pre, f"downs.{i}.0")
            x = self._resnet3d(plan, x, None, init_block, f"downs.{i}.1", f, with_cond=True)
            for j, rb in enumerate(res_blocks):
                x = self._resnet3d(plan, x, None, rb, f"downs.{i}.2.{j}", f, with_cond=False)
                hiddens.append(x)
            if isinstance(attn_block, TransformerBlock3dP):
                x = self._transformer3d(plan, x, attn_block, f"downs.{i}.3", f)
            if not it:
                x = self._temporal_peg(plan, x, peg, f"downs.{i}.4", f)
                x = self._temporal_attn(plan, x, tattn, f"downs.{i}.5", f)
            hiddens.append(x)
            self.taps[f'down{i}'] = x
            if tdown is not None and not it:
                x = self._temporal_down(plan, x, tdown, f"downs.{i}.6", f)
                f //= strides[i]
            if post is not None:
                x = self._downsample3d(plan, x, post, f"downs.{i}.7")
        self.taps['mid_in'] = x
        x = self._resnet3d(plan, x, None, u.mid_block1, "mid_block1", f, with_cond=True)
        self.taps['mid_block1'] = x
        if u.mid_attn is not None:                    # Residual(Attention) over all f*h*w tokens, no context, no feed-forward
            tok = self.clip(x, f).tokens()
            y, _ = self._self_attn(plan, tok, u.mid_attn.fn, "mid_attn.fn", with_context=False)
            x = Act(y.t, x.B, x.H, x.W, x.C, x.C, x.H * x.W * x.C)
        self.taps['mid_attn'] = x
        if not it:
            x = self._temporal_peg(plan, x, u.mid_temporal_peg, "mid_temporal_peg", f)
            self.taps['mid_peg'] = x
            x = self._temporal_attn(plan, x, u.mid_temporal_attn, "mid_temporal_attn", f)
            self.taps['mid_tattn'] = x
        x = self._resnet3d(plan, x, None, u.mid_block2, "mid_block2", f, with_cond=True)
        self.taps['mid'] = x
        for i, lvl in enumerate(u.ups):
            init_block, res_blocks, attn_block, peg, tattn, tup, upsample = lvl
            lv = n_levels - 1 - i
            if tup is not None and not it:
                x = self._temporal_up(plan, x, tup, f"ups.{i}.5", f)
                f *= strides[lv]
            x = self._resnet3d(plan, x, hiddens.pop(), init_block, f"ups.{i}.0", f, with_cond=True)
            for j, rb in enumerate(res_blocks):
                x = self._resnet3d(plan, x, hiddens.pop(), rb, f"ups.{i}.1.{j}", f, with_cond=False)
            if isinstance(attn_block, TransformerBlock3dP):
                x = self._transformer3d(plan, x, attn_block, f"ups.{i}.2", f)
            if not it:
                x = self._temporal_peg(plan, x, peg, f"ups.{i}.3", f)
                x = self._temporal_attn(plan, x, tattn, f"ups.{i}.4", f)
            if isinstance(upsample, PixelShuffleUpsample3dP):
                x = self._upsample3d(plan, x, upsample, f"ups.{i}.6")
            self.taps[f'up{i}'] = x
        assert not hiddens and f == F
        if u.final_res_block is not None:
            x = self._resnet3d(plan, x, None, u.final_res_block, "final_res_block", f, with_cond=False)
        self._final_conv3d(plan, x)

        dyn = Plan("kv-dynamic")
        self._emit_context_kv(dyn, self.c_time, rows_per_batch=self.ntt, k_row0_self=0, k_row0_cross=1, tag="dyn")
        plan.ops[self._kv_dynamic_anchor:self._kv_dynamic_anchor] = dyn.ops
        plan.keep.extend(dyn.keep)
        plan._arr = None
        return plan

    # ------------------------------------------------------------------------------------------ convolutions
    def _init_conv3d(self, plan) -> Act:
        """CrossEmbedLayer per frame (iv.py:1121-1146) as ONE kmax x kmax conv, or the plain init conv."""
        u = self.unet

        cc = getattr(u, 'cond_images_channels', 0)
        auxp = self.cimg.C if self.cimg is not None else 0

        def spread(w):
            """Reference input channels [cond image | x | lowres] (iv.py:1685, 1731) -> ours [x | lowres | 0.. (8)] ++ [cond image | 0.. (auxp)]."""
            wp = torch.zeros(w.shape[0], 8 + auxp, *w.shape[2:])
            wp[:, : w.shape[1] - cc] = w[:, cc:]
            wp[:, 8: 8 + cc] = w[:, :cc]
            return wp

        def make():
            if isinstance(u.init_conv, CrossEmbed3dP):
                kmax = max(u.init_conv.kernel_sizes)
                ws, bs = [], []
                for conv, k in zip(u.init_conv.convs, u.init_conv.kernel_sizes):
                    cw = _w2d(conv)
                    w = torch.zeros(*cw.weight.shape[:2], kmax, kmax)
                    p = (kmax - k) // 2
                    w[:, :, p:p + k, p:p + k] = cw.weight
                    ws.append(spread(w))
                    bs.append(cw.bias)
                return ops.pack_weight(torch.cat(ws), torch.cat(bs), self.dev, G=1)
            cw = _w2d(u.init_conv)
            return ops.pack_weight(spread(cw.weight), cw.bias, self.dev, G=1)

        out = self.new(self.R * self.F, self.S, self.S, self.lc["init_dim"])
        out.ssq = self.f32buf(out.rows)
        if not ops.igemm(plan, self.img, self.W.get("init_conv", make), out, x2=self.cimg, ssq_out=out.ssq, label="init_conv").ssq_emitted:
            out.ssq = None
        return out

    def _final_conv3d(self, plan, x: Act):
        """final_conv over cat(x, lowres_cond_img) per frame (iv.py:1928-1931) -> fp32 (R, F, C, H, W)."""
        u = self.unet
        extra = (self.fin2 if self.fin2 is not None else self.img) if self.lowres else None
        cw = _w2d(u.final_conv)

        kh_ = cw.weight.shape[-1]
        split = extra is None and self._split_output(x.C, kh_ * kh_, cw.weight.shape[0], x.rows)

        def make():
            w = cw.weight
            co, ci, kh, kw = w.shape
            if extra is None:
                return ops.pack_weight(w, cw.bias, self.dev, split=split)
            wp = torch.zeros(co, x.C + 8, kh, kw)
            wp[:, : x.C] = w[:, : x.C]
            wp[:, x.C + u.channels: x.C + 2 * u.channels] = w[:, x.C:]      # packed frame = [x | lowres | zero pad]
            return ops.pack_weight(wp, cw.bias, self.dev)

        out4 = self.out_full.view(self.R * self.F, u.channels_out, self.S, self.S)
        ops.igemm(plan, x, self.W.get("final_conv|split" if split else "final_conv", make), out4, x2=extra, out_mode=OUT_NCHW_F32, label="final_conv")
        plan.keep.append(self.out_full)
        if self.F != self.Fx:
            # out[:, :, len(pre):][:, :, :-len(post)] (iv.py:1933-1939): frames [Fpre, Fpre + Fx) of every clip; fp32 moved as fp16 pairs
            fr = 2 * u.channels_out * self.S * self.S
            ops.rows_copy(plan, self.out_full, self.out, B=self.R, rows=1, C=self.Fx * fr, src_bs=self.F * fr, src_rs=0,
                          dst_bs=self.Fx * fr, dst_rs=0, src_off=self.Fpre * fr, label="cut_frames")
            plan.keep.append(self.out)

    def set_cond_video_frames(self, cond_video_frames: Optional[torch.Tensor], post_cond_video_frames: Optional[torch.Tensor]):
        """The prompt frames of the following forward / sampling calls, each (src_batch, c, f', h, w) with values as given (the
        reference does not normalise them): nearest-resized to this engine's resolution, frame count unchanged (resize_video_to,
        iv.py:1702, 1715), and written to the static slots of the packed input clip — and of the final conv's low-res clip, where a
        low-res stage sees them as extra low-res frames (iv.py:1686-1692; there the reference needs them at the stage's size)."""
        u, R, S, F = self.unet, self.R, self.S, self.F
        c = u.channels
        img = self.img.t.view(R, F, S, S, 8)
        fin2 = self.fin2.t.view(R, F, S, S, 8) if self.fin2 is not None else None
        for v, n, at_in, at_fin in ((cond_video_frames, self.Fpre, self.Fpost, 0), (post_cond_video_frames, self.Fpost, 0, self.Fpre + self.Fx)):
            assert (v is None) == (n == 0), 'this engine was planned for another number of conditioning frames'
            if v is None:
                continue
            assert v.ndim == 5 and v.shape[0] == self.src_batch and v.shape[1] == c and v.shape[2] == n, \
                f'conditioning frames must be ({self.src_batch}, {c}, {n}, h, w), got {tuple(v.shape)}'
            v = v.to(self.dev).float()
            if self.lowres:
                assert v.shape[-1] == S and v.shape[-2] == S, \
                    'a low-res-conditioned Unet3D concatenates the conditioning frames with its low-res clip: they must have its size'
            elif tuple(v.shape[-2:]) != (S, S):
                v = TF.interpolate(v, (n, S, S), mode='nearest')
            fm = v.permute(0, 2, 3, 4, 1)                                   # (b, f', h, w, c)
            packed = torch.zeros(self.src_batch, n, S, S, 8, device=self.dev)
            packed[..., :c] = fm
            if self.lowres:
                packed[..., c:2 * c] = fm                                   # cat((frames, frames), dim = 1), iv.py:1688, 1692
            packed = packed.to(torch.float16).repeat(R // self.src_batch, 1, 1, 1, 1)
            img[:, at_in:at_in + n] = packed
            if fin2 is not None:
                fin2[:, at_fin:at_fin + n] = packed

    def _split_output(self, Cin: int, taps: int, Cout: int, pixels: int) -> bool:
        """A split-precision weight for a launch of the output stage (SPLIT_OUTPUT_* above)?"""
        return bool(SPLIT_OUTPUT_STAGE and Cin % 8 == 0 and taps * Cin <= SPLIT_OUTPUT_MAX_K
                    and 4.0 * pixels * Cout * taps * Cin <= SPLIT_OUTPUT_FLOPS)

    def _temporal_conv(self, plan, x: Act, conv, name: str, f: int) -> Act:
        """Causal Conv1d(k = 3) over the frames of every pixel (iv.py:436-449).  TEMPORAL_CONV_FUSED: ONE launch — in the (clip, frame, pixel) view it is
        a 3 x 1 window over rows = frames with two zero rows in front (ops.igemm(causal_rows=True), kernel family 0): one fp32 accumulation
        instead of three launches that pass their partial sums through fp16 (rounds 1-5: three accumulating 1x1 GEMMs on frame-shifted views,
        144 of the 320 launches of a C5 step)."""
        R = self.R
        C, P = x.C, x.H * x.W
        w = conv.weight.detach().float()                       # (C_out, C_in, 3): tap k multiplies frame f - 2 + k
        K = w.shape[-1]
        y = self.new(x.B, x.H, x.W, w.shape[0])
        Co = w.shape[0]
        if TEMPORAL_CONV_FUSED and C % 8 == 0:
            pw = self.W.raw(f"{name}.taps", w.unsqueeze(-1), conv.bias.detach().float())       # (C_out, C_in, 3, 1)
            ops.igemm(plan, Act(x.t, R, f, P, C, C, f * P * C, x.off), pw, Act(y.t, R, f, P, Co, Co, f * P * Co, y.off), causal_rows=True,
                      label=f"{name}.taps")
            return y
        for shift in range(min(K, f)):                         # shift 0: the current frame (with the bias), 1: f-1, 2: f-2
            tap = K - 1 - shift
            pw = self.W.raw(f"{name}.tap{tap}", w[:, :, tap], conv.bias.detach().float() if shift == 0 else None)
            n = f - shift
            xin = Act(x.t, R, n, P, C, C, f * P * C, x.off)
            yout = Act(y.t, R, n, P, Co, Co, f * P * Co, y.off + shift * P * Co)
            ops.igemm(plan, xin, pw, yout, res=yout if shift else None, label=f"{name}.tap{tap}")
        return y

    # ------------------------------------------------------------------------------------------ ResnetBlock (iv.py:743-815)
    def _resnet3d(self, plan, x: Act, skip: Optional[Act], rb, name: str, f: int, with_cond: bool) -> Act:
        W, R = self.W, self.R
        C1, C2 = x.C, (skip.C if skip is not None else 0)
        Cin, Cout = C1 + C2, rb.dim_out
        assert Cin == rb.dim, f"{name}: {Cin} input channels, block expects {rb.dim}"
        s = self.unet.skip_connect_scale
        H, Wd = x.H, x.W
        temporal = not self.ignore_time
        in_scale = None
        if skip is not None:
            in_scale = torch.ones(Cin)
            in_scale[C1:] = s
        sx = self._ssq_of(plan, x, name + ".block1.stat_x")
        ss = self._ssq_of(plan, skip, name + ".block1.stat_skip") if skip is not None else None
        # (the output stage's block1 — no skip input, batch-shared affine — takes a split-precision weight: _split_output)
        w1 = W.conv(name + ".block1", rb.block1.project.spatial_conv,
                    split=name == "final_res_block" and skip is None and self._split_output(Cin, 9, Cout, R * f * H * Wd))
        pa1 = W.f32(name + ".block1.pa", lambda: _pad_vec(rb.block1.norm.gamma.detach().float().flatten().cpu() * math.sqrt(Cin)
                                                         * (in_scale if in_scale is not None else 1.0), w1.Cin_pad))
        off = self._blk_off[self._blk_index[id(rb)]]
        pa2, ps2 = self.pa2[:, off:], self.ps2[:, off:]
        h1 = self.new(R * f, H, Wd, Cout)
        h1.ssq = self.f32buf(h1.rows)
        op = ops.igemm(plan, x, w1, h1, x2=skip, ssq_a=sx, ssq_b=ss, ssq_wb=s * s, pa=pa1, pstride=0, act_in=ACT_SILU,
                       ssq_out=None if temporal else h1.ssq, label=name + ".block1")
        if temporal or not op.ssq_emitted:
            h1.ssq = None
        if temporal:
            h1 = self._temporal_conv(plan, h1, rb.block1.project.temporal_conv, name + ".block1.temporal", f)
        if rb.cross_attn is not None:
            assert with_cond
            h1 = self.frames(self._cross_attn(plan, self.clip(h1, f), rb.cross_attn, name + ".cross_attn"), f)
        s1 = self._ssq_of(plan, h1, name + ".block2.stat")
        h2 = self.new(R * f, H, Wd, Cout)
        # pa2 / ps2 hold one row per frame of the FULL clip (R*F rows, equal within a clip); at a level with f < F frames, frame
        # b' = r*f + i reads row b' * (F/f) = r*F + i*(F/f), which lies inside clip r
        ops.igemm(plan, h1, W.conv(name + ".block2", rb.block2.project.spatial_conv), h2, ssq_a=s1, pa=pa2, ps=ps2,
                  pstride=self.total_c * (self.F // f), act_in=ACT_SILU, label=name + ".block2")
        if temporal:
            h2 = self._temporal_conv(plan, h2, rb.block2.project.temporal_conv, name + ".block2.temporal", f)
        gate = None
        h2c = self.clip(h2, f)
        if rb.gca is not None:                                  # softmax pooling over ALL f*h*w positions of the clip (iv.py:1022-1027)
            g = rb.gca
            hidden = g.net[0].weight.shape[0]
            gate = self.f32buf(R, Cout)
            chunks = ops.gca_chunks(f * H * Wd, R, Cout)
            part = self.f32buf(R, chunks, Cout + 2)
            ops.gca(plan, h2c, W.f32(name + ".gca.wk", lambda: g.to_k.weight.reshape(-1)), float(g.to_k.bias.detach().float().item()),
                    W.f32(name + ".gca.w1t", lambda: g.net[0].weight.reshape(hidden, Cout).t()), W.f32(name + ".gca.b1", lambda: g.net[0].bias),
                    W.f32(name + ".gca.w2t", lambda: g.net[2].weight.reshape(Cout, hidden).t()), W.f32(name + ".gca.b2", lambda: g.net[2].bias),
                    part, gate, chunks, label=name + ".gca")
        out = self.new(R * f, H, Wd, Cout)
        out.ssq = self.f32buf(out.rows)
        outc, xc = self.clip(out, f), self.clip(x, f)
        if rb.res_conv is not None:                             # 1x1: the clip view keeps the gate per clip
            wr = W.conv(name + ".res_conv", _w2d(rb.res_conv), in_scale=in_scale)
            skc = self.clip(skip, f) if skip is not None else None
            if gate is not None:
                op = ops.igemm(plan, xc, wr, outc, x2=skc, addend=h2c, gate=gate, ssq_out=out.ssq, label=name + ".res_conv")
            else:
                op = ops.igemm(plan, xc, wr, outc, x2=skc, res=h2c, ssq_out=out.ssq, label=name + ".res_conv")
            if not op.ssq_emitted:
                out.ssq = None
        else:
            assert skip is None
            ops.gate_residual(plan, h2c, gate, xc, outc, rs_out=out.ssq, raw_ssq=True, label=name + ".tail")
        return out

    # ------------------------------------------------------------------------------------------ temporal PEG / attention
    def _temporal_peg(self, plan, x: Act, mod, name: str, f: int) -> Act:
        conv = mod.fn[1]
        C = x.C
        out = self.new(x.B, x.H, x.W, C)
        ops.temporal_peg(plan, x, self.W.f32(name + ".w", lambda: conv.weight.reshape(C, 3)), self.W.f32(name + ".b", lambda: conv.bias), out,
                         B=self.R, F=f, causal=self.unet.time_causal_attn, label=name)
        return out

    def _position_bias(self, attn, f: int) -> torch.Tensor:
        """[heads, f, f+1] fp32: column 0 = the learned null-key bias, columns 1.. = DynamicPositionBias(i - j) (iv.py:1208-1223,
        547-552).  Input-independent: evaluated once from the parameters when the plan is built."""
        rp = attn.rel_pos_bias
        with torch.no_grad():
            pos = torch.arange(-f + 1, f, dtype=torch.float32).reshape(-1, 1)
            for layer in list(rp.mlp)[:-1]:
                lin, norm = layer[0], layer[1]
                h = TF.linear(pos, lin.weight.detach().float().cpu(), lin.bias.detach().float().cpu())
                h = (h - h.mean(-1, keepdim=True)) * torch.rsqrt(h.var(-1, unbiased=False, keepdim=True) + 1e-5) * norm.g.detach().float().cpu()
                pos = TF.silu(h)
            last = rp.mlp[-1]
            pos = TF.linear(pos, last.weight.detach().float().cpu(), last.bias.detach().float().cpu())       # (2f-1, heads)
            idx = torch.arange(f).reshape(-1, 1) - torch.arange(f).reshape(1, -1) + (f - 1)
            bias = pos[idx].permute(2, 0, 1)                                                               # (heads, f, f)
            null = attn.null_attn_bias.detach().float().cpu().reshape(-1, 1, 1).expand(-1, f, 1)
            return torch.cat((null, bias), dim=-1).contiguous()

    def _temporal_attn(self, plan, x: Act, mod, name: str, f: int) -> Act:
        """x + Attention(causal, relative position bias) along the frames of every pixel (iv.py:257-270, 1416)."""
        W, R = self.W, self.R
        attn = mod.fn.fn
        nm = name + ".fn.fn"
        C, P = x.C, x.H * x.W
        heads, dh = attn.heads, attn.dim_head
        assert dh == 64
        inner = heads * dh
        rows = R * f * P
        tok = Act(x.t, 1, 1, rows, C, C, rows * C, x.off)
        mu, rs = self.f32buf(rows), self.f32buf(rows)
        ops.rowstat(plan, tok, mode=1, rs=rs, mu=mu, eps=1e-5, label=nm + ".norm")
        wqkv = W.raw(nm + ".qkv", torch.cat((attn.to_q.weight.detach().float(), attn.to_kv.weight.detach().float())), None)
        qkv = self.new(1, 1, rows, inner + 2 * dh)
        ops.igemm(plan, tok, wqkv, qkv, mu=mu, rs=rs, pa=W.f32(nm + ".norm.g", lambda: _pad_vec(attn.norm.g, wqkv.Cin_pad)), label=nm + ".qkv")
        o = self.new(1, 1, rows, inner)
        ops.temporal_attention(plan, qkv, W.f32(nm + ".null_kv", lambda: attn.null_kv), W.f32(nm + ".q_scale", lambda: attn.q_scale),
                               W.f32(nm + ".k_scale", lambda: attn.k_scale), W.get(f"{nm}.bias.{f}", lambda: self._position_bias(attn, f).to(self.dev)),
                               o, B=R, F=f, P=P, heads=heads, causal=attn.causal, scale=SIM_SCALE, label=nm + ".attn")
        y = self.new(1, 1, rows, C)
        ops.igemm(plan, o, W.conv(nm + ".to_out", attn.to_out[0]), y, label=nm + ".to_out")
        out = self.new(x.B, x.H, x.W, C)
        ops.ln_residual(plan, y, W.f32(nm + ".out_g", lambda: attn.to_out[1].g), Act(out.t, 1, 1, rows, C, C, rows * C), res=tok, eps=1e-5,
                        label=nm + ".out_norm")
        return out

    # ------------------------------------------------------------------------------------------ TransformerBlock (iv.py:1059-1091)
    def _transformer3d(self, plan, x: Act, tb: TransformerBlock3dP, name: str, f: int) -> Act:
        cur = x
        for d, (attn, ff) in enumerate(tb.layers):
            nm = f"{name}.layers.{d}"
            y, _ = self._self_attn(plan, self.clip(cur, f).tokens(), attn, nm + ".0", with_context=True)
            cur = Act(y.t, x.B, x.H, x.W, x.C, x.C, x.H * x.W * x.C)
            cur = self._chan_feed_forward(plan, cur, ff, nm + ".1", f)
        return cur

    def _chan_feed_forward(self, plan, x: Act, ff, name: str, f: int) -> Act:
        """x + ChanFeedForward(x) (iv.py:1048-1057): per-position LayerNorm over C -> 1x1 -> GELU -> [second half of the hidden channels
        taken from the previous frame] -> LayerNorm -> 1x1."""
        W, R = self.W, self.R
        C, P = x.C, x.H * x.W
        rows = R * f * P
        shift = getattr(ff, "time_token_shift", True)
        j = 4 if shift else 3
        tok = Act(x.t, 1, 1, rows, C, C, rows * C, x.off)
        mu, rs = self.f32buf(rows), self.f32buf(rows)
        ops.rowstat(plan, tok, mode=1, rs=rs, mu=mu, eps=1e-5, label=name + ".ln0")
        w1 = W.conv(name + ".w1", _w2d(ff[1]))
        hidden = w1.Cout
        hid = self.new(1, 1, rows, hidden)
        ops.igemm(plan, tok, w1, hid, mu=mu, rs=rs, pa=W.f32(name + ".g0", lambda: _pad_vec(ff[0].g, w1.Cin_pad)), act_out=ACT_GELU,
                  label=name + ".conv1")
        if shift and x.B == R * f:                                # 5-D input only (iv.py:1041-1042); here always
            half = hidden // 2 + hidden % 2                       # torch.chunk(2): the first chunk takes the ceiling
            assert half % 8 == 0 and (hidden - half) % 8 == 0, "time token shift needs 8-channel aligned halves"
            sh = self.new(1, 1, rows, hidden)
            fr = P * hidden                                       # elements per frame
            ops.rows_copy(plan, hid.t, sh.t, B=R, rows=f * P, C=half, src_bs=f * fr, src_rs=hidden, dst_bs=f * fr, dst_rs=hidden,
                          label=name + ".keep_half")
            if f > 1:
                ops.rows_copy(plan, hid.t, sh.t, B=R, rows=(f - 1) * P, C=hidden - half, src_bs=f * fr, src_rs=hidden, dst_bs=f * fr,
                              dst_rs=hidden, src_off=half, dst_off=fr + half, label=name + ".shift_half")
            zeros = W.get(("zeros16", hidden), lambda: torch.zeros(hidden, dtype=torch.float16, device=self.dev))
            ops.rows_copy(plan, zeros, sh.t, B=R, rows=P, C=hidden - half, src_bs=0, src_rs=0, dst_bs=f * fr, dst_rs=hidden, dst_off=half,
                          label=name + ".shift_zero")
            hid = sh
        mu2, rs2 = self.f32buf(rows), self.f32buf(rows)
        ops.rowstat(plan, hid, mode=1, rs=rs2, mu=mu2, eps=1e-5, label=name + ".ln1")
        w2 = W.conv(name + ".w2", _w2d(ff[j + 1]))
        out = self.new(x.B, x.H, x.W, C)
        out.ssq = self.f32buf(rows)
        op = ops.igemm(plan, hid, w2, Act(out.t, 1, 1, rows, C, C, rows * C), mu=mu2, rs=rs2,
                       pa=W.f32(name + ".g1", lambda: _pad_vec(ff[j].g, w2.Cin_pad)), res=tok, ssq_out=out.ssq, label=name + ".conv2")
        if not op.ssq_emitted:
            out.ssq = None
        return out

    # ------------------------------------------------------------------------------------------ resampling
    def _downsample3d(self, plan, x: Act, mod, name: str) -> Act:
        if isinstance(mod, Parallel3dP):      # last level: conv3x3 + conv1x1 summed == one 3x3 conv (iv.py:1465)
            def make():
                a, b = _w2d(mod.fns[0]), _w2d(mod.fns[1])
                w3 = a.weight.clone()
                w3[:, :, 1, 1] += b.weight[:, :, 0, 0]
                return ops.pack_weight(w3, a.bias + b.bias, self.dev)
            w = self.W.get(name, make)
            out = self.new(x.B, x.H, x.W, w.Cout)
            out.ssq = self.f32buf(out.rows)
            if not ops.igemm(plan, x, w, out, ssq_out=out.ssq, label=name).ssq_emitted:
                out.ssq = None
            return out
        cw = _w2d(mod[1])                     # pixel-unshuffle + 1x1 conv per frame (iv.py:640-645) == 2x2 stride-2 conv
        w = self.W.get(name, lambda: ops.pack_weight(cw.weight.reshape(cw.weight.shape[0], x.C, 2, 2), cw.bias, self.dev))
        out = self.new(x.B, x.H // 2, x.W // 2, w.Cout)
        out.ssq = self.f32buf(out.rows)
        if not ops.igemm(plan, x, w, out, stride=2, pad=0, ssq_out=out.ssq, label=name).ssq_emitted:
            out.ssq = None
        return out

    def _upsample3d(self, plan, x: Act, mod: PixelShuffleUpsample3dP, name: str) -> Act:
        cw = _w2d(mod.net[0])
        c4 = cw.weight.shape[0]
        cq = c4 // 4
        perm = torch.arange(c4).view(cq, 4).t().reshape(-1)
        w = self.W.conv(name, cw, out_perm=perm)
        out = self.new(x.B, 2 * x.H, 2 * x.W, cq)
        ops.igemm(plan, x, w, out, act_out=ACT_SILU, out_mode=OUT_PIXEL_SHUFFLE, label=name)
        return out

    def _temporal_down(self, plan, x: Act, mod, name: str, f: int) -> Act:
        """'b c (f p) h w -> b (c p) f h w' + 1x1 conv (iv.py:681-686), p = 2: a two-input GEMM over the even / odd frames."""
        stride = mod.stride
        assert stride == 2, "temporal stride 2 only (two-input GEMM)"
        cw = _w2d(mod[1])
        C = x.C
        w = cw.weight[:, :, 0, 0]                                           # (o, c*2) with input channel index c*2 + p

        def make():
            return ops.pack_weight(torch.cat((w[:, 0::2], w[:, 1::2]), dim=1), cw.bias, self.dev)
        pw = self.W.get(name, make)
        n = x.B // 2
        fr = x.H * x.W * C
        even = Act(x.t, n, x.H, x.W, C, C, 2 * fr, x.off)
        odd = Act(x.t, n, x.H, x.W, C, C, 2 * fr, x.off + fr)
        out = self.new(n, x.H, x.W, pw.Cout)
        out.ssq = self.f32buf(out.rows)
        if not ops.igemm(plan, even, pw, out, x2=odd, ssq_out=out.ssq, label=name).ssq_emitted:
            out.ssq = None
        return out

    def _temporal_up(self, plan, x: Act, mod, name: str, f: int) -> Act:
        """Conv1d(C -> C*r, 1) + SiLU + 'b (c r) n -> b c (n r)' (iv.py:649-679): one GEMM per phase j < r writing frames j::r."""
        r = mod.stride
        cw = _w2d(mod.net[0])
        w, b = cw.weight[:, :, 0, 0], cw.bias                                # rows c*r + j
        co = w.shape[0] // r
        out = self.new(x.B * r, x.H, x.W, co)
        fr = x.H * x.W * co
        for j in range(r):
            pw = self.W.get(f"{name}.phase{j}", lambda j=j: ops.pack_weight(w[j::r], b[j::r], self.dev))
            dst = Act(out.t, x.B, x.H, x.W, co, co, r * fr, out.off + j * fr)
            ops.igemm(plan, x, pw, dst, act_out=ACT_SILU, label=f"{name}.phase{j}")
        return out
