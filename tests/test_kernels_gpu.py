"""Per-kernel parity tests (MI355X): every HIP op called through the C ABI vs a plain fp32 torch reference on CPU.

Inputs are rounded to fp16 first (the kernels' storage type), so the comparison isolates the kernel's own
arithmetic (fp32 accumulate, fp16 rounding of MFMA operands / outputs).  Tolerance: normwise relative error
<= 1e-3 (BASELINE.json north_star: "within 1e-3 rel fp16/bf16"), integer/selection work (quantile) exact.
"""
import math

import pytest
import torch

from conftest import gpu_device
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

TOL = 1e-3


def nerr(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return ((a - b).norm() / b.norm().clamp(min=1e-12)).item()


def h16(t):
    return t.half().float()


@pytest.fixture(scope="module")
def dev():
    return gpu_device()


@pytest.fixture(scope="module")
def ops():
    from imagen_pytorch_amd import ops as o

    return o


def test_single_hip_runtime(ops):
    from imagen_pytorch_amd import _abi

    _abi.load_library()
    torch.zeros(1, device="cuda")
    copies = _abi.hip_runtime_copies()
    assert len(copies) == 1, f"more than one HIP runtime mapped: {copies}"


def _run(plan):
    plan.run()
    torch.cuda.synchronize()


# ------------------------------------------------------------------------------------------------ igemm

CONV_CASES = [
    # B, H, W, C1, C2, Cout, with_affine
    (2, 64, 64, 32, 0, 32, True),
    (2, 32, 32, 32, 32, 32, True),      # concat (x, skip * 2^-1/2)
    (2, 16, 16, 128, 0, 128, True),
    (2, 8, 8, 256, 128, 256, False),
    (1, 24, 40, 64, 32, 64, True),      # ragged tiles
    (3, 8, 8, 16, 8, 24, True),         # 8-channel-chunk path (G = 1)
    (2, 16, 24, 96, 32, 128, True),     # 64-channel chunks (G = 8) with the concat boundary inside a chunk
    (1, 32, 32, 256, 256, 192, True),   # G = 8, Cout not a multiple of the 128 tile
]


@pytest.mark.parametrize("B,H,W,C1,C2,Cout,affine", CONV_CASES)
def test_igemm_block_conv3x3(ops, dev, B, H, W, C1, C2, Cout, affine):
    """Block: ChanRMSNorm -> (scale+1, shift) -> SiLU -> conv3x3 + bias (ip.py:671-691) with a two-tensor concat input."""
    torch.manual_seed(0)
    C = C1 + C2
    x1 = h16(torch.randn(B, C1, H, W))
    x2 = h16(torch.randn(B, C2, H, W) * 1.5) if C2 else None
    w = torch.randn(Cout, C, 3, 3) / math.sqrt(9 * C)
    bias = torch.randn(Cout) * 0.1
    gamma = 1 + 0.1 * torch.randn(C)
    scale = 0.2 * torch.randn(B, C) if affine else torch.zeros(B, C)
    shift = 0.2 * torch.randn(B, C) if affine else torch.zeros(B, C)
    sk = 2 ** -0.5
    # reference
    xin = x1 if x2 is None else torch.cat((x1, x2 * sk), dim=1)
    hn = F.normalize(xin, dim=1) * math.sqrt(C) * gamma.view(1, C, 1, 1)
    hn = F.silu(hn * (scale.view(B, C, 1, 1) + 1) + shift.view(B, C, 1, 1))
    ref = F.conv2d(hn, h16(w), bias, padding=1)
    # ours
    a1 = ops.act_from_nchw(x1.to(dev))
    a2 = ops.act_from_nchw(x2.to(dev)) if x2 is not None else None
    in_scale = torch.ones(C)
    in_scale[C1:] = sk
    pw = ops.pack_weight(w, bias, dev)  # the skip scale is applied in the prologue affine, not in the weights
    rs = torch.empty(B * H * W, device=dev)
    plan = ops.Plan("conv")
    ops.rowstat(plan, a1, mode=0, rs=rs, x2=a2, w2=sk * sk)
    pa = (gamma * math.sqrt(C) * in_scale).view(1, C) * (scale + 1)
    pa_d = torch.zeros(B, pw.Cin_pad, device=dev)
    ps_d = torch.zeros(B, pw.Cin_pad, device=dev)
    pa_d[:, :C] = pa.to(dev)
    ps_d[:, :C] = shift.to(dev)
    y = ops.new_act(B, H, W, Cout, dev)
    ops.igemm(plan, a1, pw, y, x2=a2, rs=rs, pa=pa_d, ps=ps_d, pstride=pw.Cin_pad, act_in=ops.ACT_SILU)
    _run(plan)
    e = nerr(ops.act_to_nchw(y), ref)
    assert e < TOL, f"normwise error {e:.2e}"


def test_igemm_cross_embed_15x15(ops, dev):
    """CrossEmbedLayer (ip.py:1051-1076) as ONE 15x15 conv: the 3x3 / 7x7 kernels zero-embedded in a 15x15 window."""
    torch.manual_seed(1)
    B, H, W, Cin = 2, 40, 40, 6
    dims = (16, 8, 8)
    x = h16(torch.randn(B, Cin, H, W))
    ws = [torch.randn(d, Cin, k, k) / math.sqrt(Cin * k * k) for d, k in zip(dims, (3, 7, 15))]
    bs = [torch.randn(d) * 0.1 for d in dims]
    ref = torch.cat([F.conv2d(x, h16(w), b, padding=w.shape[-1] // 2) for w, b in zip(ws, bs)], dim=1)
    big = torch.zeros(sum(dims), 8, 15, 15)
    o = 0
    for w in ws:
        k = w.shape[-1]
        p = (15 - k) // 2
        big[o:o + w.shape[0], :Cin, p:p + k, p:p + k] = w
        o += w.shape[0]
    pw = ops.pack_weight(big, torch.cat(bs), dev, G=1)
    xin = torch.zeros(B, 8, H, W)
    xin[:, :Cin] = x
    a = ops.act_from_nchw(xin.to(dev))
    y = ops.new_act(B, H, W, 32, dev)
    plan = ops.Plan()
    ops.igemm(plan, a, pw, y)
    _run(plan)
    assert nerr(ops.act_to_nchw(y), ref) < TOL


def test_igemm_downsample_and_pixel_shuffle(ops, dev):
    """Downsample (pixel-unshuffle + 1x1, ip.py:633-640) as a 2x2/stride-2 conv; PixelShuffleUpsample (ip.py:603-631)."""
    torch.manual_seed(2)
    B, H, W, C, Co = 2, 32, 32, 32, 64
    x = h16(torch.randn(B, C, H, W))
    w = torch.randn(Co, 4 * C, 1, 1) / math.sqrt(4 * C)
    b = torch.randn(Co) * 0.1
    ref = F.conv2d(F.pixel_unshuffle(x, 2), h16(w), b)
    w22 = w.view(Co, C, 2, 2)  # input channel order of pixel_unshuffle is (c, s1, s2)
    pw = ops.pack_weight(w22, b, dev)
    y = ops.new_act(B, H // 2, W // 2, Co, dev)
    plan = ops.Plan()
    ops.igemm(plan, ops.act_from_nchw(x.to(dev)), pw, y, stride=2, pad=0)
    _run(plan)
    assert nerr(ops.act_to_nchw(y), ref) < TOL

    Cq = 32
    w2 = torch.randn(4 * Cq, Co, 1, 1) / math.sqrt(Co)
    b2 = torch.randn(4 * Cq) * 0.1
    xin = h16(ref)
    ref2 = F.pixel_shuffle(F.silu(F.conv2d(xin, h16(w2), b2)), 2)
    # PixelShuffle reads channel c*4 + s1*2 + s2; the kernel wants (s1, s2, c) order
    perm = torch.arange(4 * Cq).view(Cq, 4).t().reshape(-1)
    pw2 = ops.pack_weight(w2[perm], b2[perm], dev)
    y2 = ops.new_act(B, H, W, Cq, dev)
    plan = ops.Plan()
    ops.igemm(plan, ops.act_from_nchw(xin.to(dev)), pw2, y2, act_out=ops.ACT_SILU, out_mode=ops.OUT_PIXEL_SHUFFLE)
    _run(plan)
    assert nerr(ops.act_to_nchw(y2), ref2) < TOL


def test_igemm_final_conv_nchw_f32(ops, dev):
    """final_conv (ip.py:1436, 1725): 3x3 conv over cat(features, lowres image) -> fp32 NCHW, 3 channels."""
    torch.manual_seed(3)
    B, H, W = 2, 32, 32
    x = h16(torch.randn(B, 32, H, W))
    lr = h16(torch.randn(B, 8, H, W))
    lr[:, 3:] = 0
    w = torch.randn(3, 35, 3, 3) * 0.05
    b = torch.randn(3) * 0.05
    ref = F.conv2d(torch.cat((x, lr[:, :3]), 1), h16(w), b, padding=1)
    wp = torch.zeros(3, 40, 3, 3)
    wp[:, :35] = w
    pw = ops.pack_weight(wp, b, dev, G=1)
    y = torch.empty(B, 3, H, W, device=dev)
    plan = ops.Plan()
    ops.igemm(plan, ops.act_from_nchw(x.to(dev)), pw, y, x2=ops.act_from_nchw(lr.to(dev)), out_mode=ops.OUT_NCHW_F32)
    _run(plan)
    assert nerr(y, ref) < TOL


@pytest.mark.parametrize("M,K,N", [(16, 256, 1024), (2 * 1024, 128, 512), (80, 512, 2048), (8192, 256, 128)])
def test_igemm_linear_ln_gelu_residual(ops, dev, M, K, N):
    """Linear with a fused LayerNorm prologue, GELU epilogue, and a second linear with residual (FeedForward ip.py:972-980)."""
    torch.manual_seed(4)
    x = h16(torch.randn(M, K) * 1.3 + 0.2)
    g = 1 + 0.1 * torch.randn(K)
    w = torch.randn(N, K) / math.sqrt(K)
    ref = F.gelu(F.linear(F.layer_norm(x, (K,)) * g, h16(w)))
    B = 2 if M % 2 == 0 else 1
    a = ops.Act(x.half().to(dev), B, 1, M // B, K, K, (M // B) * K)
    mu = torch.empty(M, device=dev)
    rs = torch.empty(M, device=dev)
    pw = ops.pack_weight(w, None, dev)
    y = ops.new_act(B, 1, M // B, N, dev)
    plan = ops.Plan()
    ops.rowstat(plan, a, mode=1, rs=rs, mu=mu, eps=1e-5)
    pa = torch.zeros(pw.Cin_pad, device=dev)
    pa[:K] = g.to(dev)
    ops.igemm(plan, a, pw, y, mu=mu, rs=rs, pa=pa, pstride=0, act_out=ops.ACT_GELU)
    _run(plan)
    assert nerr(y.t.reshape(M, N), ref) < TOL
    # second linear: N -> K with residual x
    w2 = torch.randn(K, N) / math.sqrt(N)
    hid = h16(y.t.reshape(M, N).float().cpu())
    ref2 = F.linear(hid, h16(w2)) + x
    pw2 = ops.pack_weight(w2, None, dev)
    y2 = ops.new_act(B, 1, M // B, K, dev)
    plan = ops.Plan()
    ops.igemm(plan, y, pw2, y2, res=a)
    _run(plan)
    assert nerr(y2.t.reshape(M, K), ref2) < TOL


def test_igemm_res_conv_gate_addend(ops, dev):
    """ResnetBlock tail with a 1x1 res_conv (ip.py:755-757): out = h*gate + conv1x1(cat(x, skip*s)) + b."""
    torch.manual_seed(5)
    B, H, W, C1, C2, Co = 2, 16, 16, 64, 32, 64
    x1, x2 = h16(torch.randn(B, C1, H, W)), h16(torch.randn(B, C2, H, W))
    hh = h16(torch.randn(B, Co, H, W))
    gate = torch.rand(B, Co)
    w = torch.randn(Co, C1 + C2, 1, 1) / math.sqrt(C1 + C2)
    b = torch.randn(Co) * 0.1
    sk = 2 ** -0.5
    insc = torch.ones(C1 + C2)
    insc[C1:] = sk
    weff = h16(w * insc.view(1, -1, 1, 1))
    ref = hh * gate.view(B, Co, 1, 1) + F.conv2d(torch.cat((x1, x2), 1), weff, b)
    pw = ops.pack_weight(w, b, dev, in_scale=insc)
    y = ops.new_act(B, H, W, Co, dev)
    plan = ops.Plan()
    ops.igemm(plan, ops.act_from_nchw(x1.to(dev)), pw, y, x2=ops.act_from_nchw(x2.to(dev)),
              addend=ops.act_from_nchw(hh.to(dev)), gate=gate.to(dev))
    _run(plan)
    assert nerr(ops.act_to_nchw(y), ref) < TOL


# ------------------------------------------------------------------------------------------------ attention

ATTN_TOL_BOUNDED = 1e-3   # (round 5, call A: 5.4e-4 at worst over the parametrisation, init-sized and sd = 0.2 scale draws alike)
ATTN_ERR = [0.0, 0.0]   # worst measured error over the parametrisation: [init-sized scales (bounded tiling), sd = 0.2 scale draws]


@pytest.mark.parametrize("D", [64, 32])
@pytest.mark.parametrize("fused_qnorm", [False, True])
@pytest.mark.parametrize("bounded", [False, True, "extreme"])
@pytest.mark.parametrize("B,heads,rows,J,shared", [(2, 1, 8 * 1024, 1065, True), (2, 8, 256, 41, False), (1, 8, 36, 292, False),
                                                   (2, 1, 8 * 64, 103, True), (2, 8, 256, 80, False), (1, 1, 8 * 64, 20, True),
                                                   (1, 2, 300, 129, False), (1, 1, 8 * 64, 64, True), (1, 1, 8 * 64, 65, True),
                                                   (1, 1, 8 * 64, 128, True), (1, 2, 256, 193, False)])
def test_attention(ops, dev, B, heads, rows, J, shared, fused_qnorm, D, bounded):
    """Cosine-sim attention (ip.py:559-590 / 812-833): QNORM (as its own op, or fused into the q load) + KV_PREP + ATTENTION vs
    softmax(8 * q^ k^T) v in fp32.  bounded: the plan passes the logit bound of the scale vectors (ops.attention_logit_bound) and the
    64-dim / >= 256-row launches take the bounded-logit softmax; "extreme": unit scales, every query parallel to its first key and
    antiparallel to its second — logits at + and - the bound, the two ends of the fp16 range the bounded kernel relies on."""
    if bounded and (D != 64 or rows < 256):
        pytest.skip("the bounded-logit softmax is a tiling of the 64-dim / >= 256-row launches")
    torch.manual_seed(6)
    q = h16(torch.randn(B, rows, heads, D))
    k = h16(torch.randn(B, J, heads, D))
    v = h16(torch.randn(B, J, heads, D))
    sd = 0.04 if bounded else 0.2   # (bounded: scales near the init value 1, the bound stays under ATTN_BOUND_MAX; 0.2: it does not)
    qs, ks = 1 + sd * torch.randn(D), 1 + sd * torch.randn(D)
    if bounded == "extreme":
        qs, ks = torch.ones(D), torch.ones(D)
        q = h16(k[:, :1].expand(B, rows, heads, D) * (1 + torch.rand(B, rows, 1, 1)))
        k[:, 1] = -k[:, 0]
    qn = F.normalize(q, dim=-1) * qs
    kn = F.normalize(k, dim=-1) * ks
    sim = torch.einsum("bihd,bjhd->bhij", qn, kn) * 8
    ref = torch.einsum("bhij,bjhd->bihd", sim.softmax(-1), v)
    Jp = (J + 31) // 32 * 32
    qd = q.half().to(dev).contiguous()
    khat = torch.zeros(B, heads, Jp, D, dtype=torch.float16, device=dev)
    vt = torch.zeros(B, heads, D, Jp, dtype=torch.float16, device=dev)
    o = torch.empty(B, rows, heads, D, dtype=torch.float16, device=dev)
    kd, vd = k.half().to(dev).contiguous(), v.half().to(dev).contiguous()
    plan = ops.Plan()
    qkw = dict(q_scale=qs.to(dev), q_mult=8 * ops.LOG2E) if fused_qnorm else {}
    if bounded:
        qkw["logit_bound"] = ops.attention_logit_bound(qs, ks, 8 * ops.LOG2E)
        assert qkw["logit_bound"] <= ops.ATTN_BOUND_MAX
    if not fused_qnorm:
        ops.qnorm(plan, qd, qs.to(dev), rows=B * rows, heads=heads, ld=heads * D, mult=8 * ops.LOG2E, head_dim=D)
    ops.kv_prep(plan, kd, vd, ks.to(dev), khat, vt, B=B, heads=heads, rows=J, r0=0,
                src_strides=(J * heads * D, heads * D, D), k_strides=(heads * Jp * D, Jp * D, D),
                vt_strides=(heads * D * Jp, D * Jp, Jp), head_dim=D)
    pa = ops.attention(plan, qd, khat, vt, o, B=B, heads=heads, rows=rows, J=J, head_dim=D,
                       q_strides=(rows * heads * D, D, heads * D), k_strides=(heads * Jp * D, Jp * D, D),
                       vt_strides=(heads * D * Jp, D * Jp, Jp), o_strides=(rows * heads * D, D, heads * D), **qkw)
    assert pa.softmax_mode == int(bool(bounded))
    _run(plan)
    e = nerr(o, ref)
    # the per-op bar of every other kernel test (round 4 asserted 2e-3 here and nobody had looked at the measured figure: 5.4e-4)
    tol = ATTN_TOL_BOUNDED
    ATTN_ERR[0] = max(ATTN_ERR[0], e) if bounded else ATTN_ERR[0]
    ATTN_ERR[1] = ATTN_ERR[1] if bounded else max(ATTN_ERR[1], e)
    from conftest import record_parity
    record_parity("attention", worst_bounded_scales=ATTN_ERR[0], worst_wide_scales=ATTN_ERR[1], tol=ATTN_TOL_BOUNDED)
    assert e < tol, f"attention normwise error {e:.2e}"


def test_kv_prep_multi_matches_individual_launches(ops, dev):
    """Jobs with different head counts / row counts / source dtypes in one launch: bit-identical to one launch per job."""
    torch.manual_seed(11)
    D, B = 64, 3
    specs = [(1, 2, torch.float16), (8, 2, torch.float16), (4, 5, torch.float16), (2, 1, torch.float32)]   # (heads, rows, dtype)
    outs = []
    for batched in (False, True):
        plan, jobs, res = ops.Plan(), [], []
        torch.manual_seed(12)
        for heads, J, dt in specs:
            Jp = 16
            src = torch.randn(B, J, 2 * heads * D).to(dt).to(dev)
            ks = (torch.rand(D) + 0.5).to(dev)
            khat = torch.zeros(B, heads, Jp, D, dtype=torch.float16, device=dev)
            vt = torch.zeros(B, heads, D, Jp, dtype=torch.float16, device=dev)
            ops.kv_prep(plan, src, src, ks, khat, vt, B=B, heads=heads, rows=J, r0=3, src_strides=(J * 2 * heads * D, 2 * heads * D, D),
                        k_strides=(heads * Jp * D, Jp * D, D), vt_strides=(heads * D * Jp, D * Jp, Jp), k_off=0, v_off=heads * D,
                        batch=jobs if batched else None)
            res += [khat, vt]
        if batched:
            ops.kv_prep_multi(plan, jobs, dev)
            assert len(plan) == 1
        _run(plan)
        outs.append([r.cpu() for r in res])
    for a, b in zip(*outs):
        assert a.abs().sum() > 0 and torch.equal(a, b)


# ------------------------------------------------------------------------------------------------ glue kernels

def test_rowstat_gate_ln(ops, dev):
    torch.manual_seed(7)
    B, H, W, C = 2, 16, 16, 96
    x = h16(torch.randn(B, C, H, W) + 0.3)
    a = ops.act_from_nchw(x.to(dev))
    rs = torch.empty(B * H * W, device=dev)
    mu = torch.empty(B * H * W, device=dev)
    plan = ops.Plan()
    ops.rowstat(plan, a, mode=0, rs=rs)
    _run(plan)
    ref = 1 / x.permute(0, 2, 3, 1).reshape(-1, C).norm(dim=-1)
    assert nerr(rs, ref) < 1e-5
    plan = ops.Plan()
    ops.rowstat(plan, a, mode=1, rs=rs, mu=mu, eps=1e-5)
    _run(plan)
    rows = x.permute(0, 2, 3, 1).reshape(-1, C)
    assert nerr(mu, rows.mean(-1)) < 1e-5
    assert nerr(rs, torch.rsqrt(rows.var(-1, unbiased=False) + 1e-5)) < 1e-5
    # gate_residual + rs_out
    hh = h16(torch.randn(B, C, H, W))
    gate = torch.rand(B, C)
    out = ops.new_act(B, H, W, C, dev)
    rso = torch.empty(B * H * W, device=dev)
    plan = ops.Plan()
    ops.gate_residual(plan, ops.act_from_nchw(hh.to(dev)), gate.to(dev), a, out, rs_out=rso)
    _run(plan)
    ref = hh * gate.view(B, C, 1, 1) + x
    assert nerr(ops.act_to_nchw(out), ref) < TOL
    ref_rs = 1 / ops.act_to_nchw(out).cpu().permute(0, 2, 3, 1).reshape(-1, C).norm(dim=-1)
    assert nerr(rso, ref_rs) < 1e-5
    # ln_residual
    g = 1 + 0.1 * torch.randn(C)
    beta = 0.1 * torch.randn(C)
    out2 = ops.new_act(B, H, W, C, dev)
    plan = ops.Plan()
    ops.ln_residual(plan, a, g.to(dev), out2, beta=beta.to(dev), res=ops.act_from_nchw(hh.to(dev)), eps=1e-5)
    _run(plan)
    ref = F.layer_norm(rows, (C,), g, beta, 1e-5) + hh.permute(0, 2, 3, 1).reshape(-1, C)
    assert nerr(out2.t.reshape(-1, C), ref) < TOL


@pytest.mark.parametrize("B,HW,C", [(2, 64 * 64, 32), (2, 16 * 16, 128), (1, 8 * 8, 256), (2, 24 * 24, 96), (2, 8 * 8, 512), (3, 16 * 16, 1024),
                                    (2, 8 * 8, 1024)])
def test_global_context(ops, dev, B, HW, C):
    """GlobalContext gate (ip.py:945-970).  The 512- and 1024-channel cases (C2's deep levels) take the two-phase finalisation (GCA_FINAL
    phase 1 / 2: the squeeze MLP over many workgroups) — one chunk per image and several."""
    torch.manual_seed(8)
    S = int(math.isqrt(HW))
    x = h16(torch.randn(B, C, S, S))
    hidden = max(3, C // 2)
    wk, bk = torch.randn(1, C, 1, 1) / math.sqrt(C), torch.randn(1) * 0.1
    w1, b1 = torch.randn(hidden, C, 1, 1) / math.sqrt(C), torch.randn(hidden) * 0.1
    w2, b2 = torch.randn(C, hidden, 1, 1) / math.sqrt(hidden), torch.randn(C) * 0.1
    ctx = F.conv2d(x, wk, bk).reshape(B, 1, HW)
    pooled = torch.einsum("bin,bcn->bci", ctx.softmax(-1), x.reshape(B, C, HW)).unsqueeze(-1)
    ref = torch.sigmoid(F.conv2d(F.silu(F.conv2d(pooled, w1, b1)), w2, b2)).reshape(B, C)
    a = ops.act_from_nchw(x.to(dev))
    chunks = ops.gca_chunks(HW)
    part = torch.empty(B, chunks, C + 2, device=dev)
    gate = torch.empty(B, C, device=dev)
    plan = ops.Plan()
    ops.gca(plan, a, wk.reshape(C).to(dev), float(bk), w1.reshape(hidden, C).t().contiguous().to(dev), b1.to(dev),
            w2.reshape(C, hidden).t().contiguous().to(dev), b2.to(dev), part, gate, chunks)
    if ops.gca_final_is_wide(C, hidden):
        assert [l for _, _, l in plan.ops][-2:] == ["gca.final1", "gca.final2"], [l for _, _, l in plan.ops]
    _run(plan)
    assert nerr(gate, ref) < 1e-4


@pytest.mark.parametrize("B,HW,C,mode", [(2, 64 * 64, 32, "part"), (2, 16 * 16, 128, "part"), (1, 8 * 8, 256, "gate_in"), (3, 20 * 20, 64, "part"),
                                         (2, 32 * 32, 64, "none"), (2, 12 * 12, 512, "part")])
def test_gca_tail(ops, dev, B, HW, C, mode):
    """GCA_TAIL: GlobalContext finalisation (ip.py:965-970) + h * gate + res (ip.py:755-757) in one launch, with every optional output —
    per-row sum of squares, LayerNorm statistics, the next Block's activated input silu(ChanRMSNorm(out) * gamma) (ip.py:683-690) — vs
    fp32 torch; the gate from GCA_PARTIAL rows merged in the kernel, from a ready gate, or absent."""
    torch.manual_seed(18)
    S = int(math.isqrt(HW))
    h, x = h16(torch.randn(B, C, S, S)), h16(torch.randn(B, C, S, S))
    hidden = max(4, C // 2)
    wk, bk = torch.randn(1, C, 1, 1) / math.sqrt(C), torch.randn(1) * 0.1
    w1, b1 = torch.randn(hidden, C, 1, 1) / math.sqrt(C), torch.randn(hidden) * 0.1
    w2, b2 = torch.randn(C, hidden, 1, 1) / math.sqrt(hidden), torch.randn(C) * 0.1
    ctx = F.conv2d(h, wk, bk).reshape(B, 1, HW)
    pooled = torch.einsum("bin,bcn->bci", ctx.softmax(-1), h.reshape(B, C, HW)).unsqueeze(-1)
    gate_ref = torch.sigmoid(F.conv2d(F.silu(F.conv2d(pooled, w1, b1)), w2, b2)).reshape(B, C)
    if mode == "none":
        gate_ref = torch.ones(B, C)
    gam = torch.rand(C) + 0.5
    ha, xa = ops.act_from_nchw(h.to(dev)), ops.act_from_nchw(x.to(dev))
    out = ops.new_act(B, S, S, C, dev)
    w1t, w2t = w1.reshape(hidden, C).t().contiguous().to(dev), w2.reshape(C, hidden).t().contiguous().to(dev)
    gate = torch.zeros(B, C, device=dev)
    ssq = torch.zeros(B * HW, device=dev)
    plan = ops.Plan()
    kw = {}
    if mode == "part":
        chunks = max(2, ops.gca_chunks(HW, B, C))       # several chunks: the merge is the kernel's job
        part = torch.empty(B, chunks, C + 2, device=dev)
        done = ops.gca(plan, ha, wk.reshape(C).to(dev), float(bk), w1t, b1.to(dev), w2t, b2.to(dev), part, gate, chunks, final=False)
        assert not done
        kw = dict(part=part, chunks=chunks, w1t=w1t, b1=b1.to(dev), w2t=w2t, b2=b2.to(dev), gate=gate)
    elif mode == "gate_in":
        kw = dict(gate_in=gate_ref.to(dev))
    ops.gca_tail(plan, ha, xa, out, ssq_out=ssq, label="tail", **kw)
    act = ops.request_act(out, (gam * math.sqrt(C)).to(dev))
    mu, rs = ops.request_ln_stats(out)
    assert act is not None and ops.request_act(out, gam.to(dev)) is None      # one consumer per producer
    _run(plan)
    ref = (h * gate_ref.reshape(B, C, 1, 1) + x)
    got = ops.act_to_nchw(out)
    assert nerr(got, ref) < TOL
    if mode == "part":
        assert nerr(gate, gate_ref) < 1e-4
    rows = out.t.float().reshape(B * HW, C).cpu()                      # statistics are those of the STORED fp16 values
    assert nerr(ssq, (rows ** 2).sum(-1)) < 1e-5
    assert nerr(mu, rows.mean(-1)) < 1e-4 and nerr(rs, torch.rsqrt(rows.var(-1, unbiased=False) + 1e-5)) < 1e-4
    act_ref = F.silu(F.normalize(rows, dim=-1) * gam * math.sqrt(C))
    assert nerr(act.t.reshape(B * HW, C), act_ref) < TOL


def test_step_slice_and_time_embed_clamp_the_step(ops, dev):
    """STEP_SLICE copies row *step_ptr of up to four per-step tables; TIME_EMBED reads coef[*step_ptr]: a counter outside [0, steps) — a warm-up
    launch, a counter left over from a longer schedule — is clamped to the last / first row instead of read out of bounds."""
    torch.manual_seed(13)
    T = 5
    tab_a, tab_b = torch.randn(T, 64).to(dev), torch.randn(T, 8, 16).half().to(dev)
    dst_a, dst_b = torch.zeros(64, device=dev), torch.zeros(8, 16, dtype=torch.float16, device=dev)
    step = torch.zeros(1, dtype=torch.int32, device=dev)
    coef = torch.randn(T, 8).to(dev)
    half, out = 4, 32
    freqs, w, b = torch.randn(half).to(dev), (torch.randn(out, 2 * half + 1) / 4).to(dev), (torch.randn(out) * 0.1).to(dev)
    hid = ops.new_act(2, 1, 1, out, dev)
    plan = ops.Plan()
    p = ops.step_slice(plan, [(tab_a, dst_a), (tab_b, dst_b)], step)
    assert p.steps == T
    ops.time_embed(plan, times=None, coef=coef, step_ptr=step, freqs=freqs, w=w, bias=b, hid=hid)
    for s, row in ((0, 0), (3, 3), (T - 1, T - 1), (T + 7, T - 1), (-2, 0)):
        step.fill_(s)
        _run(plan)
        assert torch.equal(dst_a.cpu(), tab_a[row].cpu()) and torch.equal(dst_b.cpu(), tab_b[row].cpu()), (s, row)
        x = coef[row, 6].cpu()
        f = x * freqs.cpu() * 2 * math.pi
        ref = F.silu(torch.cat((x.view(1), f.sin(), f.cos())) @ w.cpu().t() + b.cpu())
        assert nerr(hid.t.reshape(2, out)[0], ref) < 2e-3, (s, row)


def test_time_embed_scale_shift_pack_copy(ops, dev):
    torch.manual_seed(9)
    B, half, out = 4, 8, 256
    times = torch.randn(B) * 3
    freqs = torch.randn(half)
    w, b = torch.randn(out, 2 * half + 1) / 4, torch.randn(out) * 0.1
    f = times.view(-1, 1) * freqs.view(1, -1) * 2 * math.pi
    ref = F.silu(F.linear(torch.cat((times.view(-1, 1), f.sin(), f.cos()), -1), w, b))
    hid = ops.new_act(1, 1, B, out, dev)
    plan = ops.Plan()
    ops.time_embed(plan, times=times.to(dev), coef=None, step_ptr=None, freqs=freqs.to(dev), w=w.to(dev), bias=b.to(dev), hid=hid)
    _run(plan)
    assert nerr(hid.t.reshape(B, out), ref) < TOL
    # via the coef table + step counter (graph-replay mode)
    coef = torch.zeros(5, 8)
    coef[:, 6] = torch.arange(5) * 0.37 - 1
    step = torch.tensor([3], dtype=torch.int32, device=dev)
    plan = ops.Plan()
    ops.time_embed(plan, times=None, coef=coef.to(dev), step_ptr=step, freqs=freqs.to(dev), w=w.to(dev), bias=b.to(dev), hid=hid)
    _run(plan)
    t3 = torch.full((B,), float(coef[3, 6]))
    f = t3.view(-1, 1) * freqs.view(1, -1) * 2 * math.pi
    ref = F.silu(F.linear(torch.cat((t3.view(-1, 1), f.sin(), f.cos()), -1), w, b))
    assert nerr(hid.t.reshape(B, out), ref) < TOL
    # scale_shift
    Cs = [32, 64]
    ss = h16(torch.randn(B, 2 * sum(Cs)))
    gam = torch.randn(sum(Cs))
    idx_scale = torch.cat([torch.arange(0, 32), 64 + torch.arange(0, 64)]).int()
    idx_shift = torch.cat([32 + torch.arange(0, 32), 64 + 64 + torch.arange(0, 64)]).int()
    pa = torch.empty(B, sum(Cs), device=dev)
    ps = torch.empty(B, sum(Cs), device=dev)
    ssa = ops.Act(ss.half().to(dev), 1, 1, B, ss.shape[1], ss.shape[1], B * ss.shape[1])
    plan = ops.Plan()
    ops.scale_shift(plan, ssa, gam.to(dev), idx_scale.to(dev), idx_shift.to(dev), pa, ps)
    _run(plan)
    assert nerr(pa, gam * (ss[:, idx_scale.long()] + 1)) < 1e-6
    assert nerr(ps, ss[:, idx_shift.long()]) < 1e-6
    # pack_image with CFG replication
    x = torch.randn(2, 3, 16, 16)
    lr = torch.randn(2, 3, 16, 16)
    out_a = ops.new_act(4, 16, 16, 8, dev)
    plan = ops.Plan()
    ops.pack_image(plan, x.to(dev), lr.to(dev), out_a, brep=2)
    _run(plan)
    got = ops.act_to_nchw(out_a).cpu()
    ref = torch.cat((x, lr, torch.zeros(2, 2, 16, 16)), 1).half().float()
    assert torch.equal(got[:2], ref) and torch.equal(got[2:], ref)
    # rows_copy with broadcast source
    src = torch.randn(5, 64).half().to(dev)
    dst = torch.zeros(3, 9, 64, dtype=torch.float16, device=dev)
    plan = ops.Plan()
    ops.rows_copy(plan, src, dst, B=3, rows=5, C=64, src_bs=0, src_rs=64, dst_bs=9 * 64, dst_rs=64, dst_off=2 * 64)
    _run(plan)
    assert torch.equal(dst[:, 2:7].cpu(), src.cpu().expand(3, 5, 64)) and dst[:, :2].abs().sum() == 0


@pytest.mark.parametrize("rows,K,Cout,x32,act,res", [(16, 128, 300, False, False, True), (5, 128, 2048, True, True, False), (37, 200, 519, True, True, True),
                                                    (1, 512, 64, False, True, False)])
def test_linear_f32_time_chain(ops, dev, rows, K, Cout, x32, act, res):
    """LINEAR_F32 (the timestep-conditioning chain in fp32, ip.py:1575-1576, 738-741) against fp64 torch: fp32 rows out at fp32 accuracy —
    partial row blocks, a ragged last k chunk, a Cout that is no multiple of the 256-channel workgroup, fp16 / fp32 rows in, SiLU, the fp16
    residual rows — and SCALE_SHIFT reading the fp32 rows."""
    torch.manual_seed(rows + K)
    x = torch.randn(rows, K) * 2
    w, b = torch.randn(Cout, K) / K ** 0.5, torch.randn(Cout)
    r = h16(torch.randn(rows, Cout)) if res else None
    xin = x if x32 else h16(x)
    ref = F.linear(F.silu(xin.double()) if act else xin.double(), w.double(), b.double()) + (r.double() if res else 0)
    y = torch.full((rows, Cout), float("nan"), device=dev)
    xa = xin.to(dev).contiguous() if x32 else ops.Act(xin.half().to(dev), 1, 1, rows, K, K, rows * K)
    ra = ops.Act(r.half().to(dev), 1, 1, rows, Cout, Cout, rows * Cout) if res else None
    plan = ops.Plan()
    ops.linear_f32(plan, xa, w.t().contiguous().to(dev), b.to(dev), y, res=ra, act_in=ops.ACT_SILU if act else ops.ACT_NONE)
    _run(plan)
    assert nerr(y, ref.float()) < 2e-6, nerr(y, ref.float())
    if Cout >= 300:     # the fp32 rows as SCALE_SHIFT's input
        C = 100
        gam = torch.randn(C)
        isc, ish = torch.arange(0, C).int(), (torch.arange(0, C) + C).int()
        pa, ps = torch.empty(rows, C, device=dev), torch.empty(rows, C, device=dev)
        plan = ops.Plan()
        ops.scale_shift(plan, y, gam.to(dev), isc.to(dev), ish.to(dev), pa, ps)
        _run(plan)
        yc = y.cpu()
        assert torch.equal(pa.cpu(), gam * (yc[:, :C] + 1)) and torch.equal(ps.cpu(), yc[:, C:2 * C])


# ------------------------------------------------------------------------------------------------ sampler

@pytest.mark.parametrize("n", [3 * 64 * 64, 3 * 256 * 256, 1000])
def test_quantile_exact(ops, dev, n):
    """torch.quantile(|x0|, 0.95) (ip.py:2097-2101) reproduced bit-exactly (fp32 rank + lerp semantics)."""
    torch.manual_seed(10)
    B = 3
    x = torch.randn(B, n) * torch.tensor([0.5, 1.0, 3.0]).view(B, 1)
    x[2, : n // 2] = x[2, 0]  # heavy ties
    a = x.abs()
    ref = torch.quantile(a, 0.95, dim=-1)
    out = torch.empty(B, device=dev)
    scratch = torch.empty(B * (4 * 256 + 8), dtype=torch.int32, device=dev)
    plan = ops.Plan()
    ops.quantile(plan, a.to(dev), out, scratch, B=B, n=n, q=0.95)
    _run(plan)
    _run(plan)  # scratch is re-zeroed by the op itself: replay must give the same answer
    assert torch.equal(out.cpu(), ref), f"{out.cpu()} vs {ref}"


def test_ddpm_step_vs_formula(ops, dev):
    """CFG + x0 + dynamic threshold + posterior + noise (ip.py:1522, 314-318, 2094-2109, 252-270, 2160-2164)."""
    torch.manual_seed(11)
    B, n = 2, 3 * 32 * 32
    x = torch.randn(B, n)
    pred = torch.randn(2 * B, n)
    noise = torch.randn(B, n)
    T = 4
    times = torch.linspace(1.0, 0.0, T + 1)
    logsnr = lambda t: -torch.log(((torch.cos((t + 0.008) / 1.008 * math.pi * 0.5) ** -2) - 1).clamp(min=1e-5))
    coef = torch.zeros(T, 8)
    for i in range(T):
        l, ln = logsnr(times[i]), logsnr(times[i + 1])
        coef[i] = torch.tensor([torch.sqrt(torch.sigmoid(l)), torch.sqrt(torch.sigmoid(-l)), torch.sqrt(torch.sigmoid(ln)),
                                torch.sqrt(torch.sigmoid(-ln)), -torch.expm1(l - ln), 0.0 if times[i + 1] == 0 else 1.0, l, 0.0])
    for step_i in (1, T - 1):
        alpha, sigma, alpha_n, sigma_n, c, nz = coef[step_i, :6]
        eps = pred[B:] + (pred[:B] - pred[B:]) * 3.0
        x0 = (x - sigma * eps) / alpha.clamp(min=1e-8)
        s = torch.quantile(x0.abs(), 0.95, dim=-1).clamp(min=1.0).view(B, 1)
        x0c = x0.clamp(-s, s) / s
        mean = alpha_n * (x * (1 - c) / alpha + c * x0c)
        ref = mean + nz * torch.sqrt((sigma_n ** 2 * c).clamp(min=1e-20)) * noise
        xd = x.clone().to(dev)
        x0d, ab = torch.empty(B, n, device=dev), torch.empty(B, n, device=dev)
        q = torch.empty(B, device=dev)
        scratch = torch.empty(B * (4 * 256 + 8), dtype=torch.int32, device=dev)
        step = torch.tensor([step_i], dtype=torch.int32, device=dev)
        final = torch.zeros(B, n, device=dev)
        plan = ops.Plan()
        ops.cfg_x0(plan, xd, pred.to(dev), coef.to(dev), step, x0d, ab, B=B, n_per_sample=n, cfg=True, cond_scale=3.0)
        ops.quantile(plan, ab, q, scratch, B=B, n=n, q=0.95)
        ops.ddpm_update(plan, xd, x0d, q, coef.to(dev), noise.to(dev), final, step, B=B, n_per_sample=n, dynamic_threshold=True,
                        total_steps=T, seed=0, stream_id=0)
        _run(plan)
        assert nerr(xd, ref) < 1e-5
        assert int(step.item()) == step_i + 1
        if step_i == T - 1:
            assert nerr(final, (ref.clamp(-1, 1) + 1) * 0.5) < 1e-5


def test_ddpm_step_row_keys_draw_each_request_its_own_noise(ops, dev):
    """ABI 11, ImagenDdpmUpdateParams.row_keys: (Philox key, global sample index) per row.  A uniform table reproduces the seed_ptr / sample_offset
    path bit for bit; rows keyed as a second request (its own seed, indices restarting at 0) get exactly what that request draws alone."""
    torch.manual_seed(12)
    B, n, T = 5, 3 * 16 * 16, 4
    x, x0 = torch.randn(B, n), torch.randn(B, n)
    coef = torch.zeros(T, 8)
    coef[:, 0], coef[:, 2], coef[:, 3], coef[:, 4], coef[:, 5] = 0.8, 0.85, 0.5, 0.3, 1.0

    def run(rows, **kw):
        xd, step = x[rows].clone().to(dev), torch.tensor([1], dtype=torch.int32, device=dev)
        plan = ops.Plan()
        ops.ddpm_update(plan, xd, x0[rows].to(dev), None, coef.to(dev), None, None, step, B=len(rows), n_per_sample=n, dynamic_threshold=False,
                        total_steps=T, seed=0, stream_id=1, **kw)
        _run(plan)
        return xd.cpu()

    def key_rows(spans):
        k = torch.zeros(sum(c for _, c, _ in spans), 4, dtype=torch.int32)
        r = 0
        for sd, c, i0 in spans:
            k[r:r + c, 0], k[r:r + c, 1], k[r:r + c, 2] = sd & 0x7FFFFFFF, (sd >> 31) & 0x7FFFFFFF, torch.arange(i0, i0 + c, dtype=torch.int32)
            r += c
        return k.to(dev)

    sa, sb = 1234567, (7 << 31) | 99
    seed_dev = lambda sd: torch.tensor([sd & 0x7FFFFFFF, (sd >> 31) & 0x7FFFFFFF], dtype=torch.int32, device=dev)
    all_rows = list(range(B))
    alone = run(all_rows, seed_ptr=seed_dev(sa), sample_offset=3)
    assert torch.equal(run(all_rows, row_keys=key_rows([(sa, B, 3)])), alone)
    merged = run(all_rows, row_keys=key_rows([(sa, 3, 0), (sb, 2, 0)]))
    assert torch.equal(merged[:3], run([0, 1, 2], seed_ptr=seed_dev(sa)))
    assert torch.equal(merged[3:], run([3, 4], seed_ptr=seed_dev(sb)))
    assert not torch.equal(merged[3:], alone[3:])


def test_lincomb_masked_blend_and_counter(ops, dev):
    """LINCOMB: out = mask ? w0*t0 + w1*t1 + w4*z : mask_else, in place, row picked by the device counter; `advance` bumps it,
    a DDPM_UPDATE with advance=False leaves it alone (the inpainting step sequence of Imagen._stage)."""
    torch.manual_seed(21)
    B, n = 2, 3 * 16 * 16
    known, noise, x = torch.randn(B, n), torch.randn(B, n), torch.randn(B, n)
    mask = (torch.rand(B, n) > 0.4).float()
    coef = torch.zeros(3, 8)
    coef[:, 0] = torch.tensor([0.3, 0.6, 0.9])
    coef[:, 1] = torch.tensor([0.8, 0.5, 0.2])
    step = torch.tensor([1], dtype=torch.int32, device=dev)
    xd = x.to(dev)
    plan = ops.Plan()
    ops.lincomb(plan, known.to(dev), xd, coef.to(dev), step, B=B, n_per_sample=n, t1=noise.to(dev), mask=mask.to(dev), mask_else=xd)
    _run(plan)
    ref = torch.where(mask != 0, 0.6 * known + 0.5 * noise, x)
    assert torch.allclose(xd.cpu(), ref, atol=1e-6) and int(step.item()) == 1
    # Philox column: masked-out elements untouched, the others get fresh unit-variance noise on top of w0*t0
    coef2 = torch.zeros(3, 8)
    coef2[:, 0], coef2[:, 4] = 1.0, 2.0
    big = torch.zeros(1, 1 << 18, device=dev)
    m2 = (torch.rand(1, 1 << 18) > 0.5).float().to(dev)
    plan = ops.Plan()
    ops.lincomb(plan, big, big, coef2.to(dev), step, B=1, n_per_sample=1 << 18, mask=m2, mask_else=big, advance=True, seed=3, stream_id=0x100)
    _run(plan)
    assert int(step.item()) == 2
    z = big[m2 != 0]
    assert (big[m2 == 0] == 0).all() and abs(z.std().item() - 2.0) < 0.02 and abs(z.mean().item()) < 0.02
    # posterior step that keeps the counter
    x0 = torch.zeros(B, n, device=dev)
    c3 = torch.tensor([[1.0, 0.0, 1.0, 1.0, 1.0, 1.0, 0.0, 0.0]] * 3, device=dev)
    plan = ops.Plan()
    ops.ddpm_update(plan, xd, x0, None, c3, None, None, step, B=B, n_per_sample=n, dynamic_threshold=False, total_steps=3, seed=1,
                    stream_id=0, advance=False)
    _run(plan)
    assert int(step.item()) == 2


def test_philox_normal_statistics(ops, dev):
    B, n = 1, 1 << 20
    x = torch.zeros(B, n, device=dev)
    x0 = torch.zeros(B, n, device=dev)
    coef = torch.tensor([[1.0, 0.0, 1.0, 1.0, 1.0, 1.0, 0.0, 0.0]], device=dev)  # mean = x0 = 0, var = 1
    step = torch.zeros(1, dtype=torch.int32, device=dev)
    plan = ops.Plan()
    ops.ddpm_update(plan, x, x0, None, coef, None, None, step, B=B, n_per_sample=n, dynamic_threshold=False, total_steps=5,
                    seed=1234, stream_id=7)
    _run(plan)
    z = x.cpu().flatten()
    assert abs(z.mean().item()) < 5e-3 and abs(z.std().item() - 1) < 5e-3
    assert abs((z ** 4).mean().item() - 3) < 0.05
    assert abs(torch.corrcoef(torch.stack((z[:-1], z[1:])))[0, 1].item()) < 5e-3


def test_graph_capture_replay(ops, dev):
    """A plan captured into a hipGraph replays with the device step counter advancing (per-timestep graph)."""
    torch.manual_seed(12)
    B, n, T = 1, 4096, 3
    x = torch.zeros(B, n, device=dev)
    x0 = torch.zeros(B, n, device=dev)
    coef = torch.zeros(T, 8, device=dev)
    coef[:, 0] = 1
    coef[:, 2] = 1
    coef[:, 3] = 1
    coef[:, 4] = 1
    coef[:, 5] = 1
    step = torch.zeros(1, dtype=torch.int32, device=dev)
    plan = ops.Plan()
    ops.ddpm_update(plan, x, x0, None, coef, None, None, step, B=B, n_per_sample=n, dynamic_threshold=False, total_steps=T,
                    seed=5, stream_id=0)
    stream = torch.cuda.Stream()
    torch.cuda.synchronize()
    with torch.cuda.stream(stream):
        g = ops.Graph(plan, stream)
        outs = []
        for _ in range(T):
            g.launch()
            stream.synchronize()
            outs.append(x.clone())
    assert int(step.item()) == T
    assert not torch.equal(outs[0], outs[1]) and not torch.equal(outs[1], outs[2])  # fresh noise per step
