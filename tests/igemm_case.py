"""Test infrastructure: one IGEMM launch described by a plain dict, run through the C ABI on the GPU and compared with a plain fp32
torch-CPU restatement of the op contract in include/imagen_hip.h (ImagenIgemmParams):

    in(p, c)  = concat(x1, x2)
    a(p, c)   = act_in((in - mu[p]) * rs[p] * pa[b, c] + ps[b, c]), 0 outside the image          rs may come from ssq_a (+ ssq_wb * ssq_b)
    acc       = conv(a, W, stride, pad)
    v         = act_out(acc + bias);  v += addend * gate[b, c]  |  v += res
    y         = v (NHWC fp16 | pixel-shuffle | NCHW fp32);  ssq_out = sum_c fp16(v)^2
    post:       y = silu(v / max(||v||, 1e-12) * post_pa[b, c] + post_ps[b, c])

Inputs are fp16-representable (the kernels' storage type); the weights are NOT pre-rounded: their fp16 rounding is inside the
tolerance.  Used by tests/test_igemm_cfgs_gpu.py (every tile configuration / code path) and tests/test_bench_shapes_gpu.py (every
distinct launch of the benchmark's denoiser plans, with the tile configuration the planner picked for it).
"""
import math

import torch
import torch.nn.functional as F


def h16(t):
    return t.half().float()


def nerr(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return ((a - b).norm() / b.norm().clamp(min=1e-12)).item()


DEFAULTS = dict(B=2, H=16, W=16, C1=32, C2=0, Cout=32, K=3, stride=1, pad=None, G=None, cfg=None,
                prologue="rs",      # none | rs | ssq | ln  (ln = mu + rs, LayerNorm statistics)
                affine=True,        # per-(batch, channel) pa / ps
                act_in="silu", act_out="none", epilogue="plain",   # plain | post | addend | res | shuffle | nchw
                ssq_out=False, bias=True, seed=0,
                gca=False)          # GlobalContext partials of the output from the epilogue (families 1 / 2, plain output, one cout tile)


def run_case(ops, dev, **kw):
    """Returns dict(err=..., err_ssq=..., cfg=(id, th, tw)).  Raises if the launcher refuses the combination."""
    c = dict(DEFAULTS)
    c.update(kw)
    g = torch.Generator().manual_seed(c["seed"])
    rn = lambda *s: torch.randn(*s, generator=g)
    B, H, W, C1, C2, Cout, K, stride = c["B"], c["H"], c["W"], c["C1"], c["C2"], c["Cout"], c["K"], c["stride"]
    C = C1 + C2
    pad = c["pad"] if c["pad"] is not None else ((K - 1) // 2 if stride == 1 else 0)
    x1 = h16(rn(B, C1, H, W) * 1.2 + 0.1)
    x2 = h16(rn(B, C2, H, W) * 0.8) if C2 else None
    w = rn(Cout, C, K, K) / math.sqrt(K * K * C)
    bias = rn(Cout) * 0.1 if c["bias"] else None
    xin = x1 if x2 is None else torch.cat((x1, x2), 1)
    # ---- prologue (reference)
    pro = c["prologue"]
    pa = (1 + 0.2 * rn(B, C)) if c["affine"] else (1 + 0.2 * rn(1, C)).expand(B, C)
    ps = (0.2 * rn(B, C)) if c["affine"] else torch.zeros(B, C)
    a = xin
    wb = 0.5
    if pro in ("rs", "ssq"):
        q = (x1 * x1).sum(1, keepdim=True) + (wb * (x2 * x2).sum(1, keepdim=True) if x2 is not None else 0.0)
        rs_ref = 1.0 / q.sqrt().clamp(min=1e-12) if pro == "ssq" else 1.0 / (q.sqrt() + 0.3)   # "rs": an arbitrary per-pixel scale
        a = a * rs_ref
    elif pro == "ln":
        mu_ref = xin.mean(1, keepdim=True)
        rs_ref = torch.rsqrt(xin.var(1, unbiased=False, keepdim=True) + 1e-5)
        a = (a - mu_ref) * rs_ref
    if pro != "none":
        a = a * pa.view(B, C, 1, 1) + ps.view(B, C, 1, 1)
    if c["act_in"] == "silu":
        a = F.silu(a)
    v = F.conv2d(a, w, bias, stride=stride, padding=pad)
    OH, OW = v.shape[2], v.shape[3]
    if c["act_out"] == "silu":
        v = F.silu(v)
    elif c["act_out"] == "gelu":
        v = F.gelu(v)
    ep = c["epilogue"]
    add = gate = res = None
    if ep == "addend":
        add, gate = h16(rn(B, Cout, OH, OW)), torch.rand(B, Cout, generator=g)
        v = v + add * gate.view(B, Cout, 1, 1)
    elif ep == "res":
        res = h16(rn(B, Cout, OH, OW))
        v = v + res
    post_pa = post_ps = None
    if ep == "post":
        post_pa, post_ps = 1 + 0.2 * rn(B, Cout), 0.2 * rn(B, Cout)
        v = F.silu(F.normalize(v, dim=1) * post_pa.view(B, Cout, 1, 1) + post_ps.view(B, Cout, 1, 1))
    ref = F.pixel_shuffle(v, 2) if ep == "shuffle" else v
    ref_ssq = (h16(v) ** 2).sum(1).reshape(-1)

    # ---- device side
    a1 = ops.act_from_nchw(x1.to(dev))
    a2 = ops.act_from_nchw(x2.to(dev)) if x2 is not None else None
    if ep == "shuffle":   # the kernel wants the output channels in (s1, s2, c) order (PixelShuffle reads channel c*4 + s1*2 + s2)
        perm = torch.arange(Cout).view(Cout // 4, 4).t().reshape(-1)
        pw = ops.pack_weight(w[perm], bias[perm] if bias is not None else None, dev, G=c["G"])
    else:
        pw = ops.pack_weight(w, bias, dev, G=c["G"])
    kwargs = {}
    if pro != "none":
        pad_c = lambda t: torch.cat((t, torch.zeros(t.shape[0], pw.Cin_pad - C)), 1).contiguous().to(dev)
        if c["affine"]:
            kwargs.update(pa=pad_c(pa), ps=pad_c(ps), pstride=pw.Cin_pad)
        else:
            kwargs.update(pa=pad_c(pa[:1]), pstride=0)
    if pro == "rs":
        kwargs["rs"] = rs_ref.reshape(-1).contiguous().to(dev)
    elif pro == "ssq":
        kwargs["ssq_a"] = (x1 * x1).sum(1).reshape(-1).contiguous().to(dev)
        if x2 is not None:
            kwargs["ssq_b"] = (x2 * x2).sum(1).reshape(-1).contiguous().to(dev)
            kwargs["ssq_wb"] = wb
    elif pro == "ln":
        kwargs["rs"] = rs_ref.reshape(-1).contiguous().to(dev)
        kwargs["mu"] = mu_ref.reshape(-1).contiguous().to(dev)
    act = dict(none=ops.ACT_NONE, silu=ops.ACT_SILU, gelu=ops.ACT_GELU)
    kwargs["act_in"] = act[c["act_in"]]
    kwargs["act_out"] = act[c["act_out"]]
    if ep == "addend":
        kwargs.update(addend=ops.act_from_nchw(add.to(dev)), gate=gate.to(dev))
    elif ep == "res":
        kwargs["res"] = ops.act_from_nchw(res.to(dev))
    ssq_t = None
    if c["ssq_out"]:
        ssq_t = torch.full((B * OH * OW,), -1.0, device=dev)
        kwargs["ssq_out"] = ssq_t
    if ep == "post":
        kwargs["post"] = dict(pa=post_pa.contiguous().to(dev), ps=post_ps.contiguous().to(dev), pstride=Cout)
    if ep == "nchw":
        y = torch.full((B, Cout, OH, OW), float("nan"), device=dev)
        kwargs["out_mode"] = ops.OUT_NCHW_F32
    elif ep == "shuffle":
        y = ops.new_act(B, 2 * OH, 2 * OW, Cout // 4, dev)
        kwargs["out_mode"] = ops.OUT_PIXEL_SHUFFLE
    else:
        y = ops.new_act(B, OH, OW, Cout, dev)
    if ep != "nchw":
        y.t.fill_(float("nan"))
    if c["gca"]:
        wk_ref, bk_ref = rn(Cout) * 0.3, 0.1
        kwargs["gca"] = dict(wk=wk_ref.to(dev), bk=bk_ref)
    plan = ops.Plan("case")
    p = ops.igemm(plan, a1, pw, y, x2=a2, stride=stride, pad=pad, cfg=c["cfg"], **kwargs)
    if ep == "post":
        assert p.post_applied, "post_pa epilogue was not applied (tile narrower than Cout?)"
    if c["ssq_out"]:
        assert p.ssq_emitted, "ssq_out was not emitted (tile narrower than Cout?)"
    plan.run()
    torch.cuda.synchronize()
    got = y if ep == "nchw" else ops.act_to_nchw(y)
    out = dict(err=nerr(got, ref), cfg=(p.cfg, p.TH, p.TW))
    if c["ssq_out"]:
        out["err_ssq"] = nerr(ssq_t, ref_ssq)
    if c["gca"]:
        assert p.gca_part_t is not None, "GlobalContext partials were not emitted (family 0 cfg or tile narrower than Cout?)"
        hq = h16(v).permute(0, 2, 3, 1).reshape(B, OH * OW, Cout)
        ctx_ref = torch.einsum("bn,bnc->bc", (hq @ wk_ref + bk_ref).softmax(-1), hq)
        rows = p.gca_part_t.cpu()
        w = torch.exp(rows[:, :, 0] - rows[:, :, 0].max(dim=1, keepdim=True).values)
        ctx = torch.einsum("bk,bkc->bc", w, rows[:, :, 2:]) / (w * rows[:, :, 1]).sum(1, keepdim=True)
        out["err_gca"] = nerr(ctx, ctx_ref)
    return out
