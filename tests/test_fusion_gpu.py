"""MI355X: epilogue-emitted ChanRMSNorm statistics (ssq_out -> ssq_a / ssq_b) give the same Block output as the
separate ROWSTAT pass, for every tile family (cross-wave LDS reduction included), and the fused tails emit the right sums."""
import math

import pytest
import torch

from conftest import gpu_device
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def insc_ref(C, C1, sk):
    v = torch.ones(1, C, 1, 1)
    v[:, C1:] = sk
    return v


def nerr(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return ((a - b).norm() / b.norm().clamp(min=1e-12)).item()


@pytest.mark.parametrize("B,H,W,C1,C2,Cmid,Cout", [(2, 32, 32, 32, 32, 32, 32), (2, 16, 16, 64, 64, 128, 128), (1, 64, 64, 64, 0, 64, 64),
                                                   (2, 8, 8, 128, 0, 128, 256), (1, 24, 24, 16, 8, 24, 16)])
def test_ssq_chain(B, H, W, C1, C2, Cmid, Cout):
    """conv_a emits ssq of its output; conv_b consumes (ssq_a, ssq_b over a concat) instead of a ROWSTAT pass."""
    from imagen_pytorch_amd import ops

    dev = gpu_device()
    torch.manual_seed(0)
    sk = 2 ** -0.5
    x1 = torch.randn(B, C1, H, W).half().float()
    x2 = torch.randn(B, C2, H, W).half().float() if C2 else None
    C = C1 + C2
    wa = torch.randn(Cmid, C, 3, 3) / math.sqrt(9 * C)
    wb = torch.randn(Cout, Cmid + C1, 3, 3) / math.sqrt(9 * (Cmid + C1))
    # reference: y = conv_a(cat(x1, x2*sk)) ; z = conv_b(silu(rms(cat(y, x1*sk))))
    xin = x1 if x2 is None else torch.cat((x1, x2 * sk), 1)
    y = F.conv2d(x1 if x2 is None else torch.cat((x1, x2), 1), (wa * insc_ref(C, C1, sk)).half().float(), None, padding=1)
    y16 = y.half().float()
    cat = torch.cat((y16, x1 * sk), 1)
    hn = F.silu(F.normalize(cat, dim=1) * math.sqrt(Cmid + C1))
    z = F.conv2d(hn, wb.half().float(), None, padding=1)

    a1 = ops.act_from_nchw(x1.to(dev))
    a2 = ops.act_from_nchw(x2.to(dev)) if x2 is not None else None
    insc = torch.ones(C)
    insc[C1:] = sk
    pwa, pwb = ops.pack_weight(wa, None, dev, in_scale=insc), ops.pack_weight(wb, None, dev)
    ya = ops.new_act(B, H, W, Cmid, dev)
    ssq_y = torch.full((B * H * W,), -1.0, device=dev)
    ssq_x1 = torch.empty(B * H * W, device=dev)
    plan = ops.Plan()
    op = ops.igemm(plan, a1, pwa, ya, x2=a2, ssq_out=ssq_y)
    if not op.ssq_emitted:   # tile narrower than Cout: the planner's fallback
        ops.rowstat(plan, ya, mode=2, rs=ssq_y)
    ops.rowstat(plan, a1, mode=2, rs=ssq_x1)
    pa = torch.ones(pwb.Cin_pad, device=dev) * math.sqrt(Cmid + C1)
    pa[Cmid:Cmid + C1] *= sk
    zb = ops.new_act(B, H, W, Cout, dev)
    ops.igemm(plan, ya, pwb, zb, x2=a1, ssq_a=ssq_y, ssq_b=ssq_x1, ssq_wb=sk * sk, pa=pa, act_in=ops.ACT_SILU)
    plan.run()
    torch.cuda.synchronize()
    ref_ssq = (y16 ** 2).sum(1).reshape(-1)
    assert nerr(ssq_y, ref_ssq) < 2e-3, "epilogue sum of squares"
    assert nerr(ops.act_to_nchw(ya), y) < 1e-3
    assert nerr(ops.act_to_nchw(zb), z) < 2e-3


def test_tail_kernels_emit_raw_ssq():
    from imagen_pytorch_amd import ops

    dev = gpu_device()
    torch.manual_seed(1)
    B, H, W, C = 2, 16, 16, 64
    h = torch.randn(B, C, H, W).half().float()
    x = torch.randn(B, C, H, W).half().float()
    gate = torch.rand(B, C)
    out = ops.new_act(B, H, W, C, dev)
    ssq = torch.empty(B * H * W, device=dev)
    plan = ops.Plan()
    ops.gate_residual(plan, ops.act_from_nchw(h.to(dev)), gate.to(dev), ops.act_from_nchw(x.to(dev)), out, rs_out=ssq, raw_ssq=True)
    g = 1 + 0.1 * torch.randn(C)
    out2 = ops.new_act(B, H, W, C, dev)
    ssq2 = torch.empty(B * H * W, device=dev)
    ops.ln_residual(plan, ops.act_from_nchw(h.to(dev)), g.to(dev), out2, res=ops.act_from_nchw(x.to(dev)), ssq_out=ssq2)
    plan.run()
    torch.cuda.synchronize()
    o = ops.act_to_nchw(out).cpu()
    assert nerr(ssq, (o ** 2).sum(1).reshape(-1)) < 1e-5
    o2 = ops.act_to_nchw(out2).cpu()
    assert nerr(ssq2, (o2 ** 2).sum(1).reshape(-1)) < 1e-5


@pytest.mark.parametrize("single", [0, 2])   # 2: the small-map path (one chunk per image, finalised by its own workgroup)
@pytest.mark.parametrize("B,S,C", [(2, 64, 32), (2, 32, 64), (3, 16, 128), (1, 24, 64), (2, 8, 512)])
def test_global_context_of_conv_output(B, S, C, single, monkeypatch):
    """GlobalContext (ip.py:945-970) of a conv output vs fp32 torch, as chunk partials + a finalisation kernel (default) and as
    ONE launch (one chunk per image, finalised by its own workgroup); run twice."""
    from imagen_pytorch_amd import ops

    dev = gpu_device()
    torch.manual_seed(3)
    x = torch.randn(B, C, S, S).half().float()
    w = torch.randn(C, C, 3, 3) / math.sqrt(9 * C)
    bias = torch.randn(C) * 0.1
    hidden = max(3, C // 2)
    wk, bk = torch.randn(1, C, 1, 1) / math.sqrt(C), torch.randn(1) * 0.1
    w1, b1 = torch.randn(hidden, C, 1, 1) / math.sqrt(C), torch.randn(hidden) * 0.1
    w2, b2 = torch.randn(C, hidden, 1, 1) / math.sqrt(hidden), torch.randn(C) * 0.1
    h = F.conv2d(x, w.half().float(), bias, padding=1).half().float()   # the gate is computed from the stored fp16 h
    ctx = F.conv2d(h, wk, bk).reshape(B, 1, S * S)
    pooled = torch.einsum("bin,bcn->bci", ctx.softmax(-1), h.reshape(B, C, S * S)).unsqueeze(-1)
    ref = torch.sigmoid(F.conv2d(F.silu(F.conv2d(pooled, w1, b1)), w2, b2)).reshape(B, C)
    gate = torch.zeros(B, C, device=dev)
    y = ops.new_act(B, S, S, C, dev)
    plan = ops.Plan()
    ops.igemm(plan, ops.act_from_nchw(x.to(dev)), ops.pack_weight(w, bias, dev), y)
    chunks = ops.gca_chunks(S * S, B, C if single == 2 else 0)
    part = torch.empty(B, chunks, C + 2, device=dev)
    ops.gca(plan, y, wk.reshape(C).to(dev), float(bk), w1.reshape(hidden, C).t().contiguous().to(dev), b1.to(dev),
            w2.reshape(C, hidden).t().contiguous().to(dev), b2.to(dev), part, gate, chunks)
    wide = bool(ops.GCA_FINAL_SPLIT) and ops.gca_final_is_wide(C, hidden)   # a wide block's squeeze MLP: GCA_FINAL phases 1 / 2 over many workgroups
    one_launch = chunks == 1 and not (C // 8) & (C // 8 - 1) and not wide   # power-of-two C/8: the in-kernel finalisation exists
    assert len(plan) == (4 if wide else 2 if one_launch else 3)
    for _ in range(2):
        gate.zero_()
        plan.run()
        torch.cuda.synchronize()
        assert nerr(gate, ref) < 1e-3, nerr(gate, ref)
    assert nerr(ops.act_to_nchw(y), h) < 1e-3


@pytest.mark.parametrize("B,S,C1,C", [(2, 32, 32, 32), (2, 16, 64, 64), (3, 8, 96, 128), (1, 24, 32, 64)])
def test_post_norm_epilogue_feeds_plain_conv(B, S, C1, C):
    """Block chaining (ip.py:671-691): conv1's epilogue applies the NEXT Block's ChanRMSNorm -> (scale+1, shift) -> SiLU to its own
    output (post_pa / post_ps), conv2 then stages that tensor with no prologue; vs fp32 torch with the norm on conv1's fp32 output."""
    from imagen_pytorch_amd import ops

    dev = gpu_device()
    torch.manual_seed(5)
    x = torch.randn(B, C1, S, S).half().float()
    w1 = (torch.randn(C, C1, 3, 3) / math.sqrt(9 * C1)).half().float()
    b1 = torch.randn(C) * 0.1
    w2 = (torch.randn(C, C, 3, 3) / math.sqrt(9 * C)).half().float()
    b2 = torch.randn(C) * 0.1
    gamma = 1 + 0.1 * torch.randn(C)
    scale, shift = 0.2 * torch.randn(B, C), 0.2 * torch.randn(B, C)
    h = F.conv2d(x, w1, b1, padding=1)
    a = F.silu(F.normalize(h, dim=1) * math.sqrt(C) * gamma.view(1, C, 1, 1) * (scale.view(B, C, 1, 1) + 1) + shift.view(B, C, 1, 1))
    ref = F.conv2d(a, w2, b2, padding=1)
    pa = ((gamma * math.sqrt(C)).view(1, C) * (scale + 1)).contiguous().to(dev)
    ps = shift.contiguous().to(dev)
    h1 = ops.new_act(B, S, S, C, dev)
    y = ops.new_act(B, S, S, C, dev)
    plan = ops.Plan()
    op = ops.igemm(plan, ops.act_from_nchw(x.to(dev)), ops.pack_weight(w1, b1, dev), h1, post=dict(pa=pa, ps=ps, pstride=C))
    assert op.post_applied
    ops.igemm(plan, h1, ops.pack_weight(w2, b2, dev), y)
    plan.run()
    torch.cuda.synchronize()
    assert nerr(ops.act_to_nchw(h1), a) < 1e-3, nerr(ops.act_to_nchw(h1), a)
    assert nerr(ops.act_to_nchw(y), ref) < 1e-3, nerr(ops.act_to_nchw(y), ref)
