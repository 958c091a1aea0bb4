"""CPU interpreter of kernel plans (TEST INFRASTRUCTURE — never imported by the product).

The planners (`imagen_pytorch_amd/engine.py`, ...) only emit `Imagen*Params` structs: raw pointers, strides, op order.  With
`dry=True` they do so over CPU memory.  This module executes such a plan on the CPU by implementing, in plain torch, the
CONTRACT of every op kind exactly as include/imagen_hip.h documents it (fp16 storage, fp32 arithmetic) and reading / writing the
same buffers through their raw addresses.  It lets the host logic — which buffer feeds which op, strides, offsets, folded
weights, the order of launches — be checked against the oracle without a GPU.  It says nothing about the HIP kernels themselves:
those are checked against fp32 torch and the oracle by the `-m gpu` tests.

Weights: `ops.KEEP_REFERENCE_WEIGHTS = True` makes `ops.pack_weight` remember the unpacked (fp16-rounded) weight behind every
packed buffer, because the packed MFMA-fragment order is a kernel-private detail.
"""
from __future__ import annotations

import math
from typing import Dict

import torch
import torch.nn.functional as F

from imagen_pytorch_amd import _abi, ops

K = _abi.ENUMS
f16, f32, i32, u8 = torch.float16, torch.float32, torch.int32, torch.uint8


class Mem:
    """Raw address -> flat typed view of the torch storage that contains it."""

    def __init__(self):
        self.storages: Dict[int, torch.UntypedStorage] = {}

    def register(self, t):
        if isinstance(t, torch.Tensor) and t.numel() > 0:
            st = t.untyped_storage()
            self.storages[st.data_ptr()] = st

    def register_plan(self, plan):
        for t in plan.keep:
            self.register(t)

    def view(self, addr, dtype) -> torch.Tensor:
        assert addr, "null pointer dereferenced by the interpreter"
        for base, st in self.storages.items():
            if base <= addr < base + st.nbytes():
                isz = torch.empty(0, dtype=dtype).element_size()
                assert (addr - base) % isz == 0
                n = (st.nbytes() - (addr - base)) // isz
                return torch.empty(0, dtype=dtype).set_(st, (addr - base) // isz, (n,), (1,))
        raise KeyError(f"address {addr:#x} is not inside any registered buffer")

    def strided(self, addr, dtype, size, stride) -> torch.Tensor:
        return self.view(addr, dtype).as_strided(size, stride)


def _act(x, kind):
    if kind == ops.ACT_SILU:
        return F.silu(x)
    if kind == ops.ACT_GELU:
        return F.gelu(x)
    return x


class Interpreter:
    def __init__(self):
        self.mem = Mem()
        self.gca_ctx = {}       # part pointer -> pooled context (GCA_PARTIAL -> GCA_FINAL hand-over)
        self.trace = []

    # ------------------------------------------------------------------------------------------------ driver
    def run(self, plan):
        self.mem.register_plan(plan)
        for kind, p, label in plan.ops:
            fn = self.DISPATCH.get(kind)
            if fn is None:
                raise NotImplementedError(f"plan interpreter: op kind {kind} ({label})")
            fn(self, p)
            self.trace.append(label)

    # ------------------------------------------------------------------------------------------------ STEP_SLICE
    def step_slice(self, p):
        m = self.mem
        step = int(m.view(p.step_ptr, torch.int32)[0])
        if p.steps > 0:
            step = min(max(step, 0), p.steps - 1)
        for k in range(4):
            w = getattr(p, f"words{k}")
            if not w:
                continue
            n = w * 16
            src = m.view(getattr(p, f"src{k}") + step * n, torch.uint8)[:n]
            m.view(getattr(p, f"dst{k}"), torch.uint8)[:n].copy_(src)

    # ------------------------------------------------------------------------------------------------ ACT_PREP
    def act_prep(self, p):
        m = self.mem
        B, n = p.rows // p.rows_per_batch, p.rows_per_batch
        x = m.strided(p.x1, f16, (B, n, p.C1), (p.bs1, p.ld1, 1)).float()
        if p.x2:
            x = torch.cat((x, m.strided(p.x2, f16, (B, n, p.C2), (p.bs2, p.ld2, 1)).float()), dim=-1)
        C = x.shape[-1]
        a = x
        if p.mu:
            a = a - m.view(p.mu, f32)[:p.rows].reshape(B, n, 1)
        if p.rs:
            a = a * m.view(p.rs, f32)[:p.rows].reshape(B, n, 1)
        elif p.self_stat:
            ssq = (x[..., :p.C1] ** 2).sum(-1).reshape(-1)
            if p.ssq_b:
                ssq = ssq + p.ssq_wb * m.view(p.ssq_b, f32)[:p.rows]
            a = a * (1.0 / ssq.sqrt().clamp(min=1e-12)).reshape(B, n, 1)
        elif p.ssq_a:
            ssq = m.view(p.ssq_a, f32)[:p.rows].clone()
            if p.ssq_b:
                ssq = ssq + p.ssq_wb * m.view(p.ssq_b, f32)[:p.rows]
            a = a * (1.0 / ssq.sqrt().clamp(min=1e-12)).reshape(B, n, 1)
        if p.pa:
            a = a * m.strided(p.pa, f32, (B, C), (p.pstride, 1)).reshape(B, 1, C)
        if p.ps:
            a = a + m.strided(p.ps, f32, (B, C), (p.pstride, 1)).reshape(B, 1, C)
        m.strided(p.y, f16, (B, n, C), (p.bsy, p.ldy, 1)).copy_(_act(a, p.act_in).half())

    # ------------------------------------------------------------------------------------------------ IGEMM
    def igemm(self, p):
        m = self.mem
        B, H, W, C1, C2 = p.B, p.H, p.W, p.C1, p.C2
        x = m.strided(p.x1, f16, (B, H, W, C1), (p.bs1, W * p.ld1, p.ld1, 1)).float()
        if p.x2:
            x = torch.cat((x, m.strided(p.x2, f16, (B, H, W, C2), (p.bs2, W * p.ld2, p.ld2, 1)).float()), dim=-1)
        C = C1 + C2
        npx = B * H * W
        a = x
        if p.mu:
            a = a - m.view(p.mu, f32)[:npx].reshape(B, H, W, 1)
        rs = None
        if p.rs:
            rs = m.view(p.rs, f32)[:npx]
        elif p.ssq_a:
            ssq = m.view(p.ssq_a, f32)[:npx].clone()
            if p.ssq_b:
                ssq = ssq + p.ssq_wb * m.view(p.ssq_b, f32)[:npx]
            rs = 1.0 / ssq.sqrt().clamp(min=1e-12)
        if rs is not None:
            a = a * rs.reshape(B, H, W, 1)
        if p.pa:
            a = a * m.strided(p.pa, f32, (B, C), (p.pstride, 1)).reshape(B, 1, 1, C)
        if p.ps:
            a = a + m.strided(p.ps, f32, (B, C), (p.pstride, 1)).reshape(B, 1, 1, C)
        a = _act(a, p.act_in).half().float()
        wref, bias = ops.REFERENCE_WEIGHTS[p.w]
        assert wref.shape[1] >= C and wref.shape[2:] == (p.KH, p.KW) and wref.shape[0] == p.Cout
        if p.pad_x1:   # an x padding of its own; rows / columns past the input are zero (OH / OW as given)
            px = p.pad_x1 - 1
            need_h, need_w = (p.OH - 1) * p.stride + p.KH, (p.OW - 1) * p.stride + p.KW
            ap = F.pad(a.permute(0, 3, 1, 2), (px, max(0, need_w - px - a.shape[2]), p.pad, max(0, need_h - p.pad - a.shape[1])))
            acc = F.conv2d(ap, wref[:, :C], None, stride=p.stride)[:, :, :p.OH, :p.OW]
        else:
            acc = F.conv2d(a.permute(0, 3, 1, 2), wref[:, :C], None, stride=p.stride, padding=p.pad)[:, :, :p.OH, :p.OW]
        assert acc.shape[2:] == (p.OH, p.OW), (acc.shape, p.OH, p.OW)
        v = acc.permute(0, 2, 3, 1)                                   # (B, OH, OW, Cout)
        if p.bias:
            v = v + m.view(p.bias, f32)[:p.Cout]
        OH, OW, Co = p.OH, p.OW, p.Cout
        if p.post_pa:
            nrm = v.norm(dim=-1, keepdim=True).clamp(min=1e-12)
            pa = m.strided(p.post_pa, f32, (B, Co), (p.post_pstride, 1)).reshape(B, 1, 1, Co)
            ps = m.strided(p.post_ps, f32, (B, Co), (p.post_pstride, 1)).reshape(B, 1, 1, Co)
            v = F.silu(v / nrm * pa + ps)
        v = _act(v, p.act_out)
        if p.addend:
            add = m.strided(p.addend, f16, (B, OH, OW, Co), (p.bs_add, OW * p.ld_add, p.ld_add, 1)).float()
            v = v + add * m.strided(p.gate, f32, (B, Co), (p.gate_stride, 1)).reshape(B, 1, 1, Co)
        if p.res:
            v = v + m.strided(p.res, f16, (B, OH, OW, Co), (p.bs_res, OW * p.ld_res, p.ld_res, 1)).float()
        if p.out_mode == ops.OUT_NCHW_F32:
            m.strided(p.y, f32, (B, Co, OH, OW), (Co * OH * OW, OH * OW, OW, 1)).copy_(v.permute(0, 3, 1, 2))
        elif p.out_mode == ops.OUT_PIXEL_SHUFFLE:
            # packed output channel o = s * (Co/4) + c with s = dy*2 + dx (the engine permutes the conv's output channels)
            cq = Co // 4
            y = m.strided(p.y, f16, (B, 2 * OH, 2 * OW, cq), (p.bsy, 2 * OW * p.ldy, p.ldy, 1))
            vv = v.reshape(B, OH, OW, 2, 2, cq)
            for dy in range(2):
                for dx in range(2):
                    y[:, dy::2, dx::2, :] = vv[:, :, :, dy, dx, :].half()
        else:
            y = m.strided(p.y, f16, (B, OH, OW, Co), (p.bsy, OW * p.ldy, p.ldy, 1))
            y.copy_(v.half())
            if p.ssq_out:
                m.view(p.ssq_out, f32)[:B * OH * OW].copy_((v.half().float() ** 2).sum(-1).reshape(-1))
            if p.gca_part:   # per output tile: (max logit, sum exp, sum exp * y) -> part[b][tile][Cout + 2]
                hq = v.half().float()
                wk = m.view(p.gca_wk, f32)[:Co]
                ty, tx = -(-OH // p.TH), -(-OW // p.TW)
                part = m.view(p.gca_part, f32)[:B * ty * tx * (Co + 2)].reshape(B, ty * tx, Co + 2)
                for iy in range(ty):
                    for ix in range(tx):
                        t = hq[:, iy * p.TH:(iy + 1) * p.TH, ix * p.TW:(ix + 1) * p.TW, :].reshape(B, -1, Co)
                        k = t @ wk + p.gca_bk
                        mx = k.max(dim=1).values
                        e = torch.exp(k - mx[:, None])
                        part[:, iy * tx + ix, 0] = mx
                        part[:, iy * tx + ix, 1] = e.sum(1)
                        part[:, iy * tx + ix, 2:] = torch.einsum("bp,bpc->bc", e, t)

    # ------------------------------------------------------------------------------------------------ statistics / glue
    def _rows(self, addr, rows, rpb, C, ld, bs):
        nb = rows // rpb
        return self.mem.strided(addr, f16, (nb, rpb, C), (bs, ld, 1))

    def rowstat(self, p):
        m = self.mem
        x = self._rows(p.x1, p.rows, p.rows_per_batch, p.C1, p.ld1, p.bs1).float().reshape(p.rows, p.C1)
        x2 = self._rows(p.x2, p.rows, p.rows_per_batch, p.C2, p.ld2, p.bs2).float().reshape(p.rows, p.C2) if p.x2 else None
        if p.mode in (0, 2):
            tot = (x ** 2).sum(-1) + (p.w2 * (x2 ** 2).sum(-1) if x2 is not None else 0.0)
            m.view(p.rs, f32)[:p.rows].copy_(tot if p.mode == 2 else 1.0 / tot.sqrt().clamp(min=1e-12))
        else:
            full = x if x2 is None else torch.cat((x, x2), dim=-1)
            m.view(p.mu, f32)[:p.rows].copy_(full.mean(-1))
            m.view(p.rs, f32)[:p.rows].copy_(torch.rsqrt(full.var(-1, unbiased=False) + p.eps))

    def gate_residual(self, p):
        m = self.mem
        h = m.strided(p.h, f16, (p.rows, p.C), (p.ld_h, 1)).float()
        res = m.strided(p.res, f16, (p.rows, p.C), (p.ld_res, 1)).float()
        if p.gate:
            nb = p.rows // p.rows_per_batch
            g = m.strided(p.gate, f32, (nb, p.C), (p.C, 1)).repeat_interleave(p.rows_per_batch, dim=0)
            h = h * g
        out = (h + res).half()
        m.strided(p.out, f16, (p.rows, p.C), (p.ld_out, 1)).copy_(out)
        if p.rs_out:
            ssq = (out.float() ** 2).sum(-1)
            m.view(p.rs_out, f32)[:p.rows].copy_(ssq if p.raw_ssq else 1.0 / ssq.sqrt().clamp(min=1e-12))

    def gca_tail(self, p):
        """GCA_TAIL: GlobalContext finalisation (GCA_FINAL's contract) + out = h * gate + res + optional per-row outputs."""
        m = self.mem
        B, HW, C = p.B, p.HW, p.C
        h = m.strided(p.h, f16, (B, HW, C), (HW * p.ld_h, p.ld_h, 1)).float()
        res = m.strided(p.res, f16, (B, HW, C), (HW * p.ld_res, p.ld_res, 1)).float()
        if p.part:
            if p.part in self.gca_ctx:
                ctx = self.gca_ctx.pop(p.part)
            else:
                rows = m.view(p.part, f32)[:B * p.chunks * (C + 2)].reshape(B, p.chunks, C + 2)
                w = torch.exp(rows[:, :, 0] - rows[:, :, 0].max(dim=1, keepdim=True).values)
                ctx = torch.einsum("bk,bkc->bc", w, rows[:, :, 2:]) / (w * rows[:, :, 1]).sum(1, keepdim=True)
            gate = self._gca_gate(ctx, p.w1t, p.b1, p.w2t, p.b2, C, p.hidden)
            if p.gate:
                m.view(p.gate, f32)[:B * C].copy_(gate.reshape(-1))
        elif p.gate_in:
            gate = m.view(p.gate_in, f32)[:B * C].reshape(B, C)
        else:
            gate = torch.ones(B, C)
        out = (h * gate.reshape(B, 1, C) + res).half()
        m.strided(p.out, f16, (B, HW, C), (HW * p.ld_out, p.ld_out, 1)).copy_(out)
        of = out.float()
        ssq = (of ** 2).sum(-1)
        if p.ssq_out:
            m.view(p.ssq_out, f32)[:B * HW].copy_(ssq.reshape(-1))
        if p.mu_out:
            m.view(p.mu_out, f32)[:B * HW].copy_(of.mean(-1).reshape(-1))
            m.view(p.rs_out, f32)[:B * HW].copy_(torch.rsqrt(of.var(-1, unbiased=False) + p.eps).reshape(-1))
        if p.act_out:
            pa = m.view(p.act_pa, f32)[:C]
            a = F.silu(of * (1.0 / ssq.sqrt().clamp(min=1e-12)).unsqueeze(-1) * pa)
            m.strided(p.act_out, f16, (B, HW, C), (HW * p.ld_act, p.ld_act, 1)).copy_(a.half())

    def ln_residual(self, p):
        m = self.mem
        y = self._rows(p.y, p.rows, p.rows_per_batch, p.C, p.ld_y, p.bs_y).float()
        out = (y - y.mean(-1, keepdim=True)) * torch.rsqrt(y.var(-1, unbiased=False, keepdim=True) + p.eps) * m.view(p.g, f32)[:p.C]
        if p.beta:
            out = out + m.view(p.beta, f32)[:p.C]
        if p.res:
            out = out + self._rows(p.res, p.rows, p.rows_per_batch, p.C, p.ld_res, p.bs_res).float()
        out = out.half()
        self._rows(p.out, p.rows, p.rows_per_batch, p.C, p.ld_out, p.bs_out).copy_(out)
        if p.ssq_out:
            m.view(p.ssq_out, f32)[:p.rows].copy_((out.float() ** 2).sum(-1).reshape(-1))
        if p.mu_out:
            of = out.float().reshape(p.rows, p.C)
            m.view(p.mu_out, f32)[:p.rows].copy_(of.mean(-1))
            m.view(p.rs_out, f32)[:p.rows].copy_(torch.rsqrt(of.var(-1, unbiased=False) + p.eps_out))

    def time_embed(self, p):
        m = self.mem
        nf = 2 * p.half_dim + 1
        if p.step_ptr:
            step = int(m.view(p.step_ptr, i32)[0])
            if p.steps > 0:
                step = min(max(step, 0), p.steps - 1)
            x = m.view(p.coef, f32)[step * 8 + 6].expand(p.B)
        else:
            x = m.view(p.times, f32)[:p.B]
        fr = x[:, None] * m.view(p.freqs, f32)[:p.half_dim][None, :] * (2 * math.pi)
        feats = torch.cat((x[:, None], fr.sin(), fr.cos()), dim=-1)
        w = m.view(p.w, f32)[:p.out_dim * nf].reshape(p.out_dim, nf)
        hid = F.silu(feats @ w.t() + m.view(p.bias, f32)[:p.out_dim])
        m.strided(p.hid, f16, (p.B, p.out_dim), (p.ld_hid, 1)).copy_(hid.half())

    def linear_f32(self, p):
        """LINEAR_F32: y = bias + f(x) @ wt (+ res), fp32 rows in (or fp16), fp32 rows out."""
        m = self.mem
        x = m.strided(p.x, f32 if p.x_f32 else f16, (p.rows, p.K), (p.ld_x, 1)).float()
        wt = m.strided(p.wt, f32, (p.K, p.Cout), (p.Cout, 1))
        v = _act(x, p.act_in) @ wt
        if p.bias:
            v = v + m.view(p.bias, f32)[:p.Cout]
        if p.res:
            v = v + m.strided(p.res, f16, (p.rows, p.Cout), (p.ld_res, 1)).float()
        m.strided(p.y, f32, (p.rows, p.Cout), (p.ld_y, 1)).copy_(v)

    def scale_shift(self, p):
        m = self.mem
        ss = m.strided(p.ss, f32 if p.ss_f32 else f16, (p.B, p.ld_ss), (p.ld_ss, 1)).float()
        isc = m.view(p.idx_scale, i32)[:p.total_c].long()
        ish = m.view(p.idx_shift, i32)[:p.total_c].long()
        m.view(p.pa, f32)[:p.B * p.total_c].copy_((m.view(p.gamma_s, f32)[:p.total_c] * (ss[:, isc] + 1.0)).reshape(-1))
        m.view(p.ps, f32)[:p.B * p.total_c].copy_(ss[:, ish].reshape(-1))

    def pack_image(self, p):
        m = self.mem
        HW = p.H * p.W
        a = m.view(p.a, f32)[:p.B * p.Ca * HW].reshape(p.B, p.Ca, p.H, p.W)
        out = torch.zeros(p.B, p.H, p.W, p.Cpad)
        out[..., :p.Ca] = a.permute(0, 2, 3, 1)
        if p.b:
            out[..., p.Ca:p.Ca + p.Cb] = m.view(p.b, f32)[:p.B * p.Cb * HW].reshape(p.B, p.Cb, p.H, p.W).permute(0, 2, 3, 1)
        m.view(p.out, f16)[:p.B * p.Brep * HW * p.Cpad].copy_(out.repeat(p.Brep, 1, 1, 1).half().reshape(-1))

    def rows_copy(self, p):
        m = self.mem
        src = m.strided(p.src, f16, (p.B, p.rows, p.C), (p.src_bs, p.src_rs, 1))
        m.strided(p.dst, f16, (p.B, p.rows, p.C), (p.dst_bs, p.dst_rs, 1)).copy_(src.clone())

    def memset32(self, p):
        self.mem.view(p.dst, i32)[:p.count].fill_(p.value if p.value < 2 ** 31 else p.value - 2 ** 32)

    def select_rows(self, p):
        m = self.mem
        src = m.view(p.src, i32)[:p.R].long()
        keep = m.view(p.keep, u8)[:p.R].bool()
        nsrc = int(src.max()) + 1
        a = m.view(p.a, f16)[:nsrc * p.L * p.C].reshape(nsrc, p.L, p.C)
        nul = m.view(p.nul, f16)[:p.L * p.C].reshape(1, p.L, p.C)
        sel = keep[:, None].expand(p.R, p.L)
        if p.mask:
            sel = sel & m.view(p.mask, u8)[:nsrc * p.L].reshape(nsrc, p.L).bool()[src]
        m.view(p.dst, f16)[:p.R * p.L * p.C].copy_(torch.where(sel[..., None], a[src], nul.expand(p.R, -1, -1)).reshape(-1))

    def mean_rows(self, p):
        m = self.mem
        x = m.strided(p.x, f16, (p.B, p.rows, p.C), (p.bs_x, p.ld_x, 1)).float()
        m.strided(p.out, f16, (p.B, p.C), (p.ld_out, 1)).copy_(x.mean(1).half())

    # ------------------------------------------------------------------------------------------------ attention
    def kv_prep(self, p):
        m = self.mem
        dt = f32 if p.src_is_f32 else f16
        D = 32 if p.head_dim == 32 else 64
        sz = (p.B, p.heads, p.rows, D)
        st = (p.src_bs, p.src_hs, p.src_rs, 1)
        k = m.strided(p.k_src, dt, sz, st).float()
        v = m.strided(p.v_src, dt, sz, st).float()
        khat = F.normalize(k, dim=-1, eps=1e-12) * m.view(p.k_scale, f32)[:D]
        m.strided(p.khat + 2 * p.r0 * p.k_rs, f16, sz, (p.k_bs, p.k_hs, p.k_rs, 1)).copy_(khat.half())
        m.strided(p.vt + 2 * p.r0, f16, sz, (p.vt_bs, p.vt_hs, 1, p.vt_ds)).copy_(v.half())

    def kv_prep_multi(self, p):
        st = _abi.STRUCTS["ImagenKvPrepParams"]
        import ctypes
        for i in range(p.n):
            job = st.from_address(p.jobs + i * ctypes.sizeof(st))
            self.kv_prep(job)

    def qnorm(self, p):
        m = self.mem
        D = 32 if p.head_dim == 32 else 64
        q = m.strided(p.q, f16, (p.rows, p.heads, D), (p.ld, D, 1))
        q.copy_((F.normalize(q.float(), dim=-1, eps=1e-12) * m.view(p.q_scale, f32)[:D] * p.mult).half())

    def attention(self, p):
        m = self.mem
        D = 32 if p.head_dim == 32 else 64
        q = m.strided(p.q, f16, (p.B, p.heads, p.rows, D), (p.q_bs, p.q_hs, p.q_rs, 1)).float()
        if p.q_scale:
            q = (F.normalize(q, dim=-1, eps=1e-12) * m.view(p.q_scale, f32)[:D] * p.q_mult).half().float()
        k = m.strided(p.k, f16, (p.B, p.heads, p.J, D), (p.k_bs, p.k_hs, p.k_rs, 1)).float()
        v = m.strided(p.vt, f16, (p.B, p.heads, p.J, D), (p.vt_bs, p.vt_hs, 1, p.vt_ds)).float()
        sim = torch.einsum("bhid,bhjd->bhij", q, k) * math.log(2.0)       # the kernel's exponent base is 2
        o = torch.einsum("bhij,bhjd->bhid", sim.softmax(-1), v)
        m.strided(p.o, f16, (p.B, p.heads, p.rows, D), (p.o_bs, p.o_hs, p.o_rs, 1)).copy_(o.half())

    # ------------------------------------------------------------------------------------------------ ROWCHAIN
    def rowchain(self, p):
        """include/imagen_hip.h, ImagenRowchainParams: the chain's layers in fp32 with fp16 roundings where the contract states them."""
        m = self.mem
        rows, C, inner = p.rows, p.C, p.inner
        r16 = lambda t: t.half().float()

        def ln(t, g, n):
            return (t - t.mean(-1, keepdim=True)) * torch.rsqrt(t.var(-1, unbiased=False, keepdim=True) + p.eps) * m.view(g, f32)[:n]

        def lin(t, w, cin, cout):
            wref, bias = ops.REFERENCE_WEIGHTS[w]
            assert bias is None and tuple(wref.shape[:2]) == (cout, cin), (wref.shape, cout, cin)
            return t @ wref[:, :, 0, 0].t()

        B, n = rows // p.rows_per_batch, p.rows_per_batch
        if p.mode == K["IMAGEN_CHAIN_RESPREP"]:
            xin = m.strided(p.x, f16, (rows, inner), (p.ld_x, 1)).float()
            if p.C2:
                xin = torch.cat((xin, m.strided(p.x2, f16, (rows, p.C2), (p.ld_x2, 1)).float()), dim=-1)
            wref, bias = ops.REFERENCE_WEIGHTS[p.w0]
            v = xin @ wref[:, :inner + p.C2, 0, 0].t()
            if p.bias:
                v = v + m.view(p.bias, f32)[:C]
            if p.addend:
                gate = m.strided(p.gate, f32, (B, C), (p.gate_stride, 1)).repeat_interleave(n, dim=0) if p.gate else 1.0
                v = v + m.strided(p.addend, f16, (rows, C), (p.ld_add, 1)).float() * gate
            out = v.half()
            m.strided(p.out, f16, (rows, C), (p.ld_out, 1)).copy_(out)
            ssq = (out.float() ** 2).sum(-1)
            if p.ssq_out:
                m.view(p.ssq_out, f32)[:rows].copy_(ssq)
            if p.prep_out:
                cat = out.float()
                tot = ssq.clone()
                if p.prep_C2:
                    cat = torch.cat((cat, m.strided(p.prep_x2, f16, (rows, p.prep_C2), (p.ld_prep_x2, 1)).float()), dim=-1)
                    if p.prep_ssq_b:
                        tot = tot + p.prep_ssq_wb * m.view(p.prep_ssq_b, f32)[:rows]
                a = F.silu(cat * torch.rsqrt(tot.clamp(min=1e-24))[:, None] * m.view(p.prep_pa, f32)[:C + p.prep_C2])
                m.strided(p.prep_out, f16, (rows, C + p.prep_C2), (p.ld_prep, 1)).copy_(a.half())
            return
        x = m.strided(p.x, f16, (rows, inner if p.mode == K["IMAGEN_CHAIN_FF"] else C), (p.ld_x, 1)).float()
        if p.mode == K["IMAGEN_CHAIN_FF"]:
            res = m.strided(p.res, f16, (rows, C), (p.ld_res, 1)).float()
            x1 = r16(ln(r16(lin(x, p.w0, inner, C)), p.g0, C) + res)
            hid = r16(F.gelu(lin(r16(ln(x1, p.g1, C)), p.w1, C, p.hidden)))
            out = (lin(r16(ln(hid, p.g2, p.hidden)), p.w2, p.hidden, C) + x1).half()
        else:
            if p.mu:
                a = (x - m.view(p.mu, f32)[:rows, None]) * m.view(p.rs, f32)[:rows, None] * m.view(p.g0, f32)[:C]
            else:
                a = ln(x, p.g0, C)
            a = r16(a)
            if p.mode == K["IMAGEN_CHAIN_XATTN"]:
                H = p.heads
                q = r16(lin(a, p.w0, C, inner)).reshape(B, n, H, 64).permute(0, 2, 1, 3)
                qh = r16(F.normalize(q, dim=-1, eps=1e-12) * m.view(p.q_scale, f32)[:64] * p.q_mult)
                k = m.strided(p.khat, f16, (B, H, p.J, 64), (p.k_bs, p.k_hs, p.k_rs, 1)).float()
                v = m.strided(p.vt, f16, (B, H, p.J, 64), (p.vt_bs, p.vt_hs, 1, p.vt_ds)).float()
                sim = torch.einsum("bhid,bhjd->bhij", qh, k) * math.log(2.0)
                o = r16(torch.einsum("bhij,bhjd->bhid", sim.softmax(-1), v)).permute(0, 2, 1, 3).reshape(rows, inner)
                res = m.strided(p.res, f16, (rows, C), (p.ld_res, 1)).float() if p.res else x
                out = (ln(r16(lin(o, p.w1, inner, C)), p.g1, C) + res).half()
            else:
                y = r16(lin(a, p.w0, C, inner + 128))
                m.strided(p.out, f16, (rows, inner), (p.ld_out, 1)).copy_(y[:, :inner].half())
                kk = (F.normalize(y[:, inner:inner + 64], dim=-1, eps=1e-12) * m.view(p.k_scale, f32)[:64]).half().reshape(B, n, 64)
                m.strided(p.khat + 2 * p.r0 * p.k_rs, f16, (B, n, 64), (p.k_bs, p.k_rs, 1)).copy_(kk)
                m.strided(p.vt + 2 * p.r0, f16, (B, n, 64), (p.vt_bs, 1, p.vt_ds)).copy_(y[:, inner + 64:].half().reshape(B, n, 64))
                return
        m.strided(p.out, f16, (rows, C), (p.ld_out, 1)).copy_(out)
        if p.ssq_out:
            m.view(p.ssq_out, f32)[:rows].copy_((out.float() ** 2).sum(-1))

    # ------------------------------------------------------------------------------------------------ GlobalContext
    def _gca_gate(self, ctx, w1t, b1, w2t, b2, C, hidden):
        m = self.mem
        W1 = m.view(w1t, f32)[:C * hidden].reshape(C, hidden)
        W2 = m.view(w2t, f32)[:hidden * C].reshape(hidden, C)
        hid = F.silu(ctx @ W1 + m.view(b1, f32)[:hidden])
        return torch.sigmoid(hid @ W2 + m.view(b2, f32)[:C])

    def gca_partial(self, p):
        m = self.mem
        h = m.strided(p.h, f16, (p.B, p.HW, p.C), (p.HW * p.ld, p.ld, 1)).float()
        logits = h @ m.view(p.wk, f32)[:p.C] + p.bk
        ctx = torch.einsum("bn,bnc->bc", logits.softmax(-1), h)
        if p.gate:
            m.view(p.gate, f32)[:p.B * p.C].copy_(self._gca_gate(ctx, p.w1t, p.b1, p.w2t, p.b2, p.C, p.hidden).reshape(-1))
        else:
            self.gca_ctx[p.part] = ctx

    def gca_final(self, p):
        if p.phase == 1:     # two-phase finalisation of a wide block: the gate is complete after phase 2, computed there from the same inputs
            return
        if p.part in self.gca_ctx:
            ctx = self.gca_ctx.pop(p.part)
        else:   # partial rows in memory (written by a conv epilogue): merge (max, sum exp, sum exp * h) over the chunks
            rows = self.mem.view(p.part, f32)[:p.B * p.chunks * (p.C + 2)].reshape(p.B, p.chunks, p.C + 2)
            w = torch.exp(rows[:, :, 0] - rows[:, :, 0].max(dim=1, keepdim=True).values)
            ctx = torch.einsum("bk,bkc->bc", w, rows[:, :, 2:]) / (w * rows[:, :, 1]).sum(1, keepdim=True)
        self.mem.view(p.gate, f32)[:p.B * p.C].copy_(self._gca_gate(ctx, p.w1t, p.b1, p.w2t, p.b2, p.C, p.hidden).reshape(-1))

    # ------------------------------------------------------------------------------------------------ Imagen-Video ops
    def temporal_peg(self, p):
        m = self.mem
        x = m.strided(p.x, f16, (p.B, p.F, p.P, p.C), (p.F * p.P * p.C, p.P * p.C, p.C, 1)).float()
        w = m.view(p.w, f32)[:p.C * 3].reshape(p.C, 3)
        pad = (2, 0) if p.causal else (1, 1)
        xp = F.pad(x, (0, 0, 0, 0, pad[0], pad[1]))
        out = x + m.view(p.bias, f32)[:p.C]
        for k in range(3):
            out = out + w[:, k] * xp[:, k:k + p.F]
        m.strided(p.out, f16, (p.B, p.F, p.P, p.C), (p.F * p.P * p.C, p.P * p.C, p.C, 1)).copy_(out.half())

    def temporal_attention(self, p):
        m = self.mem
        Fr, P, H = p.F, p.P, p.heads
        rows = m.strided(p.qkv, f16, (p.B, Fr, P, H * 64 + 128), (Fr * P * p.ld, P * p.ld, p.ld, 1)).float()
        q = rows[..., :H * 64].reshape(p.B, Fr, P, H, 64).permute(0, 2, 3, 1, 4)          # b p h i d
        k, v = rows[..., H * 64:H * 64 + 64].permute(0, 2, 1, 3), rows[..., H * 64 + 64:].permute(0, 2, 1, 3)   # b p j d
        nkv = m.view(p.null_kv, f32)[:128].reshape(2, 64)
        k = torch.cat((nkv[0].expand(p.B, P, 1, 64), k), dim=2)
        v = torch.cat((nkv[1].expand(p.B, P, 1, 64), v), dim=2)
        qh = F.normalize(q, dim=-1, eps=1e-12) * m.view(p.q_scale, f32)[:64] * p.scale
        kh = F.normalize(k, dim=-1, eps=1e-12) * m.view(p.k_scale, f32)[:64]
        sim = torch.einsum("bphid,bpjd->bphij", qh, kh) + m.view(p.bias, f32)[:H * Fr * (Fr + 1)].reshape(H, Fr, Fr + 1)
        if p.causal:
            sim = sim.masked_fill(torch.ones(Fr, Fr + 1, dtype=torch.bool).triu(2), -torch.finfo(sim.dtype).max)
        o = torch.einsum("bphij,bpjd->bphid", sim.softmax(-1), v)                       # b p h i d
        o = o.permute(0, 3, 1, 2, 4).reshape(p.B, Fr, P, H * 64)
        m.strided(p.o, f16, (p.B, Fr, P, H * 64), (Fr * P * p.ld_o, P * p.ld_o, p.ld_o, 1)).copy_(o.half())

    # ------------------------------------------------------------------------------------------------ sampler ops (injected noise only)
    def _coef_row(self, coef, step_ptr):
        step = int(self.mem.view(step_ptr, i32)[0])
        return step, self.mem.view(coef, f32)[step * 8: step * 8 + 8]

    def cfg_x0(self, p):
        m = self.mem
        n = p.B * p.n_per_sample
        _, cf = self._coef_row(p.coef, p.step_ptr)
        alpha, sigma = cf[0], cf[1]
        x = m.view(p.x, f32)[:n]
        pred = m.view(p.pred, f32)[: (2 * n if p.cfg else n)]
        out = pred[n:] + (pred[:n] - pred[n:]) * p.cond_scale if p.cfg else pred
        if p.objective == 0:
            x0 = (x - sigma * out) / alpha.clamp(min=1e-8)
        elif p.objective == 1:
            x0 = out
        else:
            x0 = alpha * x - sigma * out
        m.view(p.x0, f32)[:n].copy_(x0)
        m.view(p.absx0, f32)[:n].copy_(x0.abs())

    def quantile(self, p):
        a = self.mem.view(p.absx0, f32)[: p.B * p.n].reshape(p.B, p.n)
        self.mem.view(p.out, f32)[: p.B].copy_(torch.quantile(a, p.q, dim=-1))

    def ddpm_update(self, p):
        m = self.mem
        assert p.noise, "the interpreter only supports injected noise (the Philox stream is a kernel detail)"
        n = p.B * p.n_per_sample
        step, cf = self._coef_row(p.coef, p.step_ptr)
        alpha, alpha_next, sigma_next, c, nonzero = cf[0], cf[2], cf[3], cf[4], cf[5]
        x0 = m.view(p.x0, f32)[:n].reshape(p.B, -1)
        if p.dynamic_threshold:
            s = m.view(p.quant, f32)[: p.B].clamp(min=1.0).reshape(p.B, 1)
            x0 = x0.clamp(-s, s) / s
        else:
            x0 = x0.clamp(-1.0, 1.0)
        if p.x0_thr:
            m.view(p.x0_thr, f32)[:n].copy_(x0.reshape(-1))
        x = m.view(p.x, f32)[:n].reshape(p.B, -1)
        mean = alpha_next * (x * (1.0 - c) / alpha + c * x0)
        xn = mean + nonzero * (sigma_next * sigma_next * c).clamp(min=1e-20).sqrt() * m.view(p.noise, f32)[:n].reshape(p.B, -1)
        x.copy_(xn)
        if step + 1 >= p.total_steps and p.final_out:
            m.view(p.final_out, f32)[:n].copy_(((xn.clamp(-1.0, 1.0) + 1.0) * 0.5).reshape(-1))
        if not p.no_advance:
            m.view(p.step_ptr, i32)[0] += 1

    def lincomb(self, p):
        m = self.mem
        n = p.B * p.n_per_sample
        _, w = self._coef_row(p.coef, p.step_ptr)
        assert float(w[4]) == 0.0, "the interpreter only supports injected noise (pass it as t1)"
        thr = lambda t, q: t
        if p.thr_mode == 1:
            def thr(t, q):
                s = m.view(q, f32)[: p.B].clamp(min=1.0).reshape(p.B, 1)
                return (t.reshape(p.B, -1).clamp(-s, s) / s).reshape(-1)
        elif p.thr_mode == 2:
            thr = lambda t, q: t.clamp(-1.0, 1.0)
        v = w[0] * m.view(p.t0, f32)[:n]
        if p.t1:
            v = v + w[1] * thr(m.view(p.t1, f32)[:n], p.q1)
        if p.t2:
            v = v + w[2] * m.view(p.t2, f32)[:n]
        if p.t3:
            v = v + w[3] * thr(m.view(p.t3, f32)[:n], p.q3)
        if p.mask:
            v = torch.where(m.view(p.mask, f32)[:n] != 0, v, m.view(p.mask_else, f32)[:n])
        v = v.clone()
        m.view(p.out, f32)[:n].copy_(v)
        if p.out2:
            m.view(p.out2, f32)[:n].copy_(w[5] * v)
        if p.final and p.final_out:
            m.view(p.final_out, f32)[:n].copy_((v.clamp(-1.0, 1.0) + 1.0) * 0.5)
        if p.advance:
            m.view(p.step_ptr, i32)[0] += 1

    def lowres_prep(self, p):
        m = self.mem
        img = m.view(p.img, f32)[: p.B * p.C * p.Hin * p.Win].reshape(p.B, p.C, p.Hin, p.Win)
        up = img if (p.Hin, p.Win) == (p.Hout, p.Wout) else F.interpolate(img, (p.Hout, p.Wout), mode="nearest")
        n = p.B * p.C * p.Hout * p.Wout
        m.view(p.out, f32)[:n].copy_((p.alpha * (up * 2.0 - 1.0) + p.sigma * m.view(p.noise, f32)[:n].reshape(up.shape)).reshape(-1))

    DISPATCH = {}


Interpreter.DISPATCH = {
    K["IMAGEN_OP_IGEMM"]: Interpreter.igemm, K["IMAGEN_OP_ROWSTAT"]: Interpreter.rowstat, K["IMAGEN_OP_ATTENTION"]: Interpreter.attention,
    K["IMAGEN_OP_KV_PREP"]: Interpreter.kv_prep, K["IMAGEN_OP_KV_PREP_MULTI"]: Interpreter.kv_prep_multi, K["IMAGEN_OP_QNORM"]: Interpreter.qnorm,
    K["IMAGEN_OP_GCA_PARTIAL"]: Interpreter.gca_partial, K["IMAGEN_OP_GCA_FINAL"]: Interpreter.gca_final,
    K["IMAGEN_OP_GCA_TAIL"]: Interpreter.gca_tail,
    K["IMAGEN_OP_GATE_RESIDUAL"]: Interpreter.gate_residual, K["IMAGEN_OP_LN_RESIDUAL"]: Interpreter.ln_residual,
    K["IMAGEN_OP_TIME_EMBED"]: Interpreter.time_embed, K["IMAGEN_OP_SCALE_SHIFT"]: Interpreter.scale_shift,
    K["IMAGEN_OP_PACK_IMAGE"]: Interpreter.pack_image, K["IMAGEN_OP_ROWS_COPY"]: Interpreter.rows_copy, K["IMAGEN_OP_MEMSET32"]: Interpreter.memset32,
    K["IMAGEN_OP_SELECT_ROWS"]: Interpreter.select_rows, K["IMAGEN_OP_MEAN_ROWS"]: Interpreter.mean_rows,
    K["IMAGEN_OP_CFG_X0"]: Interpreter.cfg_x0, K["IMAGEN_OP_QUANTILE"]: Interpreter.quantile, K["IMAGEN_OP_DDPM_UPDATE"]: Interpreter.ddpm_update,
    K["IMAGEN_OP_LINCOMB"]: Interpreter.lincomb, K["IMAGEN_OP_LOWRES_PREP"]: Interpreter.lowres_prep,
    K["IMAGEN_OP_TEMPORAL_PEG"]: Interpreter.temporal_peg, K["IMAGEN_OP_TEMPORAL_ATTENTION"]: Interpreter.temporal_attention,
    K["IMAGEN_OP_ACT_PREP"]: Interpreter.act_prep,
    K["IMAGEN_OP_STEP_SLICE"]: Interpreter.step_slice,
    K["IMAGEN_OP_ROWCHAIN"]: Interpreter.rowchain,
    K["IMAGEN_OP_LINEAR_F32"]: Interpreter.linear_f32,
}
