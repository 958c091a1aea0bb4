"""ROWCHAIN (csrc/rowchain.hip): the token chains of the small maps as one launch each, against
  (a) the launch-per-op plan it replaces (same kernels' contracts, same rounding points: agreement to fp32 summation order), and
  (b) a plain fp32 torch restatement of the reference's layers (ip.py:502-591, 759-834, 972-1022) on fp16-rounded inputs.
Runs on MI355X (-m gpu) and on the CPU emulation (IMAGEN_EMUL_TESTS=1, tests/test_igemm_emulated.py keeps a slice in the CPU suite)."""
import math

import pytest
import torch
import torch.nn.functional as F

from conftest import gpu_device, record_parity

pytestmark = pytest.mark.gpu

TOL = 1e-3


def nerr(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return ((a - b).norm() / b.norm().clamp(min=1e-12)).item()


def h16(t):
    return t.half().float()


def _ln(x, g, eps=1e-5):
    mu = x.mean(-1, keepdim=True)
    var = x.var(-1, unbiased=False, keepdim=True)
    return (x - mu) * torch.rsqrt(var + eps) * g


def _act(t, dev, B, N):
    from imagen_pytorch_amd import ops
    C = t.shape[-1]
    return ops.Act(t.reshape(B, 1, N, C).half().contiguous().to(dev), B, 1, N, C, C, N * C)


def _sync():
    torch.cuda.synchronize()


@pytest.fixture(params=[False, True], ids=["tile32", "tile64"])
def tile64(request, monkeypatch):
    """Both tile heights on every case that allows 64 rows (the planner takes 64-row tiles from 16384 rows on)."""
    from imagen_pytorch_amd import ops
    monkeypatch.setattr(ops, "CHAIN_TILE64_MIN_ROWS", 1 if request.param else 1 << 30)
    return request.param


FF_CASES = [(2, 64, 256), (2, 64, 32), (1, 256, 128), (3, 32, 64), (16, 1024, 32), (16, 1024, 256)]


@pytest.mark.parametrize("B,N,C", FF_CASES)
def test_rowchain_ff(B, N, C, tile64):
    from conftest import EMULATED
    from imagen_pytorch_amd import ops
    if EMULATED and B * N > 2048:
        pytest.skip("the benchmark's row counts: hardware only (the emulation runs one workgroup after the other)")
    if tile64 and N % 64:
        pytest.skip("64-row tiles need N % 64 == 0")
    dev = gpu_device()
    torch.manual_seed(0)
    inner, hidden = 512, 2 * C
    o = h16(torch.randn(B, N, inner))
    tok = h16(torch.randn(B, N, C))
    w_out = torch.randn(C, inner) / math.sqrt(inner)
    w1 = torch.randn(hidden, C) / math.sqrt(C)
    w2 = torch.randn(C, hidden) / math.sqrt(hidden)
    g_out, g0, g1 = 1 + 0.1 * torch.randn(C), 1 + 0.1 * torch.randn(C), 1 + 0.1 * torch.randn(hidden)
    # fp32 reference (ip.py:529-532, 1017, 972-980, 1018) on fp16-rounded weights
    x1 = _ln(o @ h16(w_out).t(), g_out) + tok
    hid = F.gelu(_ln(x1, g0) @ h16(w1).t())
    ref = x1 + _ln(hid, g1) @ h16(w2).t()
    pw_out, pw1, pw2 = ops.pack_weight(w_out, None, dev), ops.pack_weight(w1, None, dev), ops.pack_weight(w2, None, dev)
    gd = [t.float().to(dev) for t in (g_out, g0, g1)]
    oa, toka = _act(o, dev, B, N), _act(tok, dev, B, N)
    assert ops.rowchain_ok(C, N, 8, 64, pw_out, pw1, pw2, hidden=hidden)
    # (a) the launch-per-op plan
    plan = ops.Plan("unfused")
    y = ops.new_act(B, 1, N, C, dev)
    ops.igemm(plan, oa, pw_out, y)
    x1a = ops.new_act(B, 1, N, C, dev)
    st = (torch.empty(B * N, device=dev), torch.empty(B * N, device=dev))
    ops.ln_residual(plan, y, gd[0], x1a, res=toka, ln_stats_out=st)
    hida = ops.new_act(B, 1, N, hidden, dev)
    pad = lambda v, n: torch.cat((v, torch.zeros(n - v.numel(), device=dev)))
    ops.igemm(plan, x1a, pw1, hida, mu=st[0], rs=st[1], pa=pad(gd[1], pw1.Cin_pad), act_out=ops.ACT_GELU)
    mu2, rs2 = torch.empty(B * N, device=dev), torch.empty(B * N, device=dev)
    ops.rowstat(plan, hida, mode=1, rs=rs2, mu=mu2)
    outa = ops.new_act(B, 1, N, C, dev)
    ssq_a = torch.empty(B * N, device=dev)
    opl = ops.igemm(plan, hida, pw2, outa, mu=mu2, rs=rs2, pa=pad(gd[2], pw2.Cin_pad), res=x1a, ssq_out=ssq_a)
    plan.run()
    # (b) one ROWCHAIN launch
    chain = ops.Plan("chain")
    outb = ops.new_act(B, 1, N, C, dev)
    ssq_b = torch.empty(B * N, device=dev)
    ops.rowchain_ff(chain, oa, toka, outb, pw_out, gd[0], pw1, gd[1], pw2, gd[2], rows_per_batch=N, ssq_out=ssq_b)
    chain.run()
    _sync()
    e_ref, e_old, e_pair = nerr(outb.t.reshape(B, N, C), ref), nerr(outa.t.reshape(B, N, C), ref), nerr(outb.t, outa.t)
    record_parity(f"rowchain_ff[{B}x{N}x{C}]", vs_fp32=e_ref, unfused_vs_fp32=e_old, vs_unfused=e_pair)
    assert e_ref < TOL and e_pair < 5e-4, (e_ref, e_old, e_pair)
    want_ssq = (outb.t.reshape(B * N, C).float() ** 2).sum(-1)
    assert nerr(ssq_b, want_ssq) < 1e-5
    if opl.ssq_emitted:
        assert nerr(ssq_b, ssq_a) < 1e-3


XA_CASES = [(2, 64, 256, 41), (2, 64, 32, 39), (1, 256, 128, 39), (2, 32, 64, 70), (16, 1024, 256, 41)]


@pytest.mark.parametrize("B,N,C,J", XA_CASES)
def test_rowchain_xattn(B, N, C, J, tile64):
    from conftest import EMULATED
    from imagen_pytorch_amd import ops
    if EMULATED and B * N > 2048:
        pytest.skip("the benchmark's row counts: hardware only")
    if tile64 and N % 64:
        pytest.skip("64-row tiles need N % 64 == 0")
    dev = gpu_device()
    torch.manual_seed(1)
    heads, dh, inner = 8, 64, 512
    x = h16(torch.randn(B, N, C))
    wq = torch.randn(inner, C) / math.sqrt(C)
    w_out = torch.randn(C, inner) / math.sqrt(inner)
    g_n, g_o = 1 + 0.1 * torch.randn(C), 1 + 0.1 * torch.randn(C)
    q_scale, k_scale = 1 + 0.1 * torch.randn(dh), 1 + 0.1 * torch.randn(dh)
    k = torch.randn(B, heads, J, dh)
    v = h16(torch.randn(B, heads, J, dh))
    khat_ref = h16(F.normalize(k, dim=-1) * k_scale)
    # fp32 reference (ip.py:759-834): q from the normalised rows, cosine-sim attention with scale 8, out-projection, LayerNorm, + x
    q = (_ln(x, g_n) @ h16(wq).t()).reshape(B, N, heads, dh).permute(0, 2, 1, 3)
    qh = F.normalize(q, dim=-1) * q_scale
    sim = torch.einsum("bhid,bhjd->bhij", qh, khat_ref) * 8.0
    o = torch.einsum("bhij,bhjd->bhid", sim.softmax(-1), v).permute(0, 2, 1, 3).reshape(B, N, inner)
    ref = _ln(o @ h16(w_out).t(), g_o) + x
    Jp = ops._round_up(J, 32)
    khat = torch.zeros(B, heads, Jp, dh, dtype=torch.float16)
    vt = torch.zeros(B, heads, dh, Jp, dtype=torch.float16)
    khat[:, :, :J] = khat_ref.half()
    vt[:, :, :, :J] = v.half().transpose(2, 3)
    khat, vt = khat.to(dev), vt.to(dev)
    ks, vs = (heads * Jp * dh, Jp * dh, dh), (heads * dh * Jp, dh * Jp, Jp)
    pwq, pwo = ops.pack_weight(wq, None, dev), ops.pack_weight(w_out, None, dev)
    gn, go, qs = g_n.to(dev), g_o.to(dev), q_scale.to(dev)
    xa = _act(x, dev, B, N)
    q_mult = 8.0 * ops.LOG2E
    # (a) launch per op
    plan = ops.Plan("unfused")
    mu, rs = torch.empty(B * N, device=dev), torch.empty(B * N, device=dev)
    ops.rowstat(plan, xa, mode=1, rs=rs, mu=mu)
    qa = ops.new_act(B, 1, N, inner, dev)
    pad = lambda t, n: torch.cat((t, torch.zeros(n - t.numel(), device=dev)))
    ops.igemm(plan, xa, pwq, qa, mu=mu, rs=rs, pa=pad(gn, pwq.Cin_pad))
    oa = ops.new_act(B, 1, N, inner, dev)
    ops.attention(plan, qa.t, khat, vt, oa.t, B=B, heads=heads, rows=N, J=J, q_strides=(N * inner, dh, inner), k_strides=ks, vt_strides=vs,
                  o_strides=(N * inner, dh, inner), q_scale=qs, q_mult=q_mult, head_dim=dh)
    ya = ops.new_act(B, 1, N, C, dev)
    ops.igemm(plan, oa, pwo, ya)
    outa = ops.new_act(B, 1, N, C, dev)
    ssq_a = torch.empty(B * N, device=dev)
    ops.ln_residual(plan, ya, go, outa, res=xa, ssq_out=ssq_a)
    plan.run()
    # (b) one launch, with the caller's statistics and with its own
    outs = []
    for stats in ((mu, rs), None):
        chain = ops.Plan("chain")
        outb = ops.new_act(B, 1, N, C, dev)
        ssq_b = torch.empty(B * N, device=dev)
        ops.rowchain_xattn(chain, xa, outb, pwq, gn, pwo, go, khat, vt, heads=heads, J=J, k_strides=ks, vt_strides=vs, q_scale=qs, q_mult=q_mult,
                           rows_per_batch=N, ln_stats=stats, ssq_out=ssq_b)
        chain.run()
        _sync()
        outs.append((outb, ssq_b))
    outb, ssq_b = outs[0]
    e_ref, e_old, e_pair = nerr(outb.t.reshape(B, N, C), ref), nerr(outa.t.reshape(B, N, C), ref), nerr(outb.t, outa.t)
    record_parity(f"rowchain_xattn[{B}x{N}x{C} J{J}]", vs_fp32=e_ref, unfused_vs_fp32=e_old, vs_unfused=e_pair)
    assert e_ref < TOL and e_pair < 5e-4, (e_ref, e_old, e_pair)
    assert nerr(outs[1][0].t, outb.t) < 1e-4 and nerr(ssq_b, ssq_a) < 1e-3


QKV_CASES = [(2, 64, 256, 41), (2, 64, 32, 0), (1, 256, 128, 39), (16, 1024, 256, 41)]


@pytest.mark.parametrize("B,N,C,n_ctx", QKV_CASES)
def test_rowchain_qkv(B, N, C, n_ctx, tile64):
    from conftest import EMULATED
    from imagen_pytorch_amd import ops
    if EMULATED and B * N > 2048:
        pytest.skip("the benchmark's row counts: hardware only")
    dev = gpu_device()
    torch.manual_seed(2)
    heads, dh, inner = 8, 64, 512
    ld = inner + 2 * dh
    x = h16(torch.randn(B, N, C))
    w = torch.randn(ld, C) / math.sqrt(C)
    g_n, k_scale = 1 + 0.1 * torch.randn(C), 1 + 0.1 * torch.randn(dh)
    pw = ops.pack_weight(w, None, dev)
    gn, ksc = g_n.to(dev), k_scale.to(dev)
    xa = _act(x, dev, B, N)
    r0 = n_ctx + 1
    Jp = ops._round_up(r0 + N, 32)
    ks, vs = (Jp * dh, 0, dh), (dh * Jp, 0, Jp)
    bufs = []
    for fused in (False, True):
        khat = torch.zeros(B, Jp, dh, dtype=torch.float16, device=dev)
        vt = torch.zeros(B, dh, Jp, dtype=torch.float16, device=dev)
        qkv = ops.new_act(B, 1, N, ld, dev, zero=True)
        plan = ops.Plan("qkv")
        if fused:
            ops.rowchain_qkv(plan, xa, qkv, pw, gn, khat, vt, ksc, heads=heads, r0=r0, k_strides=ks, vt_strides=vs, rows_per_batch=N)
        else:
            mu, rs = torch.empty(B * N, device=dev), torch.empty(B * N, device=dev)
            ops.rowstat(plan, xa, mode=1, rs=rs, mu=mu)
            ops.igemm(plan, xa, pw, qkv, mu=mu, rs=rs, pa=torch.cat((gn, torch.zeros(pw.Cin_pad - C, device=dev))))
            ops.kv_prep(plan, qkv.t, qkv.t, ksc, khat, vt, B=B, heads=1, rows=N, r0=r0, src_strides=(N * ld, ld, 0), k_strides=ks, vt_strides=vs,
                        k_off=inner, v_off=inner + dh, head_dim=dh)
        plan.run()
        _sync()
        bufs.append((qkv.t.reshape(B, N, ld)[..., :inner].float().cpu(), khat.float().cpu(), vt.float().cpu()))
    y = _ln(x, g_n) @ h16(w).t()
    ref_q, ref_k, ref_v = y[..., :inner], F.normalize(h16(y[..., inner:inner + dh]), dim=-1) * k_scale, y[..., inner + dh:]
    (q0, k0, v0), (q1, k1, v1) = bufs
    errs = dict(q=nerr(q1, ref_q), k=nerr(k1[:, r0:r0 + N], ref_k), v=nerr(v1[:, :, r0:r0 + N].transpose(1, 2), ref_v),
                q_pair=nerr(q1, q0), k_pair=nerr(k1, k0), v_pair=nerr(v1, v0))
    record_parity(f"rowchain_qkv[{B}x{N}x{C}]", **errs)
    assert max(errs["q"], errs["k"], errs["v"]) < TOL and max(errs["q_pair"], errs["k_pair"], errs["v_pair"]) < 5e-4, errs
    assert not k1[:, :r0].any() and not v1[:, :, :r0].any(), "rows in front of r0 belong to the conditioning: untouched"


RP_CASES = [(2, 64, 256, 128, 256, 128), (2, 256, 128, 64, 128, 64), (1, 64, 128, 0, 128, 0), (16, 1024, 256, 128, 256, 128), (16, 4096, 128, 64, 128, 64)]


@pytest.mark.parametrize("B,N,C1,C2,C,C2n", RP_CASES)
def test_rowchain_resprep(B, N, C1, C2, C, C2n, tile64):
    """res_conv + gate * h tail of a ResnetBlock + the next Block's activated input in one launch vs the IGEMM -> ACT_PREP pair it replaces."""
    from conftest import EMULATED
    from imagen_pytorch_amd import ops
    if EMULATED and B * N > 2048:
        pytest.skip("the benchmark's row counts: hardware only")
    if tile64 and N % 64:
        pytest.skip("64-row tiles need N % 64 == 0")
    dev = gpu_device()
    torch.manual_seed(3)
    x, skip = h16(torch.randn(B, N, C1)), (h16(torch.randn(B, N, C2) * 1.4) if C2 else None)
    h2 = h16(torch.randn(B, N, C))
    nskip = h16(torch.randn(B, N, C2n) * 0.8) if C2n else None
    w = torch.randn(C, C1 + C2) / math.sqrt(C1 + C2)
    bias = 0.1 * torch.randn(C)
    gate = torch.sigmoid(torch.randn(B, C))
    pa = (1 + 0.1 * torch.randn(C + C2n)) * math.sqrt(C + C2n)
    wb = 0.5
    cat = torch.cat((x, skip), -1) if C2 else x
    ref = cat @ h16(w).t() + bias + h2 * gate[:, None, :]
    pw = ops.pack_weight(w, bias, dev)
    xa_, h2a = _act(x, dev, B, N), _act(h2, dev, B, N)
    ska = _act(skip, dev, B, N) if C2 else None
    nska = _act(nskip, dev, B, N) if C2n else None
    gd, pad = gate.float().contiguous().to(dev), pa.float().to(dev)
    nssq = (nskip.float() ** 2).sum(-1).reshape(-1).to(dev) if C2n else None
    assert ops.resprep_ok(xa_, ska, pw, N) == (B * N <= ops.RESPREP_MAX_ROWS)      # (the planner's row limit; the kernel takes every case)
    # (a) IGEMM (+ gate * addend) -> ACT_PREP
    plan = ops.Plan("unfused")
    outa, ssq_a = ops.new_act(B, 1, N, C, dev), torch.empty(B * N, device=dev)
    opa = ops.igemm(plan, xa_, pw, outa, x2=ska, addend=h2a, gate=gd, ssq_out=ssq_a)
    ya = ops.new_act(B, 1, N, C + C2n, dev)
    if opa.ssq_emitted:
        ops.act_prep(plan, outa, ya, x2=nska, ssq_a=ssq_a, ssq_b=nssq, ssq_wb=wb, pa=pad, act_in=ops.ACT_SILU)
    else:
        ops.act_prep(plan, outa, ya, x2=nska, ssq_b=nssq, ssq_wb=wb, pa=pad, act_in=ops.ACT_SILU, self_stat=True)
    plan.run()
    # (b) one ROWCHAIN launch
    chain = ops.Plan("chain")
    outb, ssq_b = ops.new_act(B, 1, N, C, dev), torch.empty(B * N, device=dev)
    ops.rowchain_resprep(chain, xa_, ska, h2a, gd, outb, pw, rows_per_batch=N, ssq_out=ssq_b)
    yb = ops.request_prep(outb, nska, nssq, wb, pad)
    assert yb is not None and ops.request_prep(outb, nska, nssq, wb, pad) is None      # one consumer per producer
    chain.run()
    _sync()
    of = outb.t.reshape(B, N, C).float().cpu()
    tot = (of ** 2).sum(-1) + (wb * (nskip ** 2).sum(-1) if C2n else 0.0)
    ref_y = F.silu((torch.cat((of, nskip), -1) if C2n else of) * torch.rsqrt(tot)[..., None] * pa)
    errs = dict(out=nerr(of, ref), out_pair=nerr(outb.t, outa.t), prep=nerr(yb.t.reshape(B, N, -1), ref_y), prep_pair=nerr(yb.t, ya.t))
    record_parity(f"rowchain_resprep[{B}x{N} {C1}+{C2}->{C}|{C2n}]", **errs)
    assert errs["out"] < TOL and errs["prep"] < TOL and errs["out_pair"] < 5e-4 and errs["prep_pair"] < 1e-3, errs
    assert nerr(ssq_b, (of ** 2).sum(-1).reshape(-1)) < 1e-5
