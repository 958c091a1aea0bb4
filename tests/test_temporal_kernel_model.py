"""CPU: a line-by-line numpy transliteration of the two kernels in csrc/temporal.hip (flat buffers, the kernels' own pointer
arithmetic, one 64-lane wave per pixel, online softmax) against the plan interpreter's high-level restatement of their contracts.
The HIP kernels have not run on a GPU yet; this checks their index arithmetic and loop bounds independently of the contract text.
(If csrc/temporal.hip changes, this model must change with it.)"""
import numpy as np
import pytest
import torch

from imagen_pytorch_amd import ops
from plan_interp import Interpreter


def peg_model(x, w, bias, B, F, P, C, causal):
    """temporal_peg_kernel: thread i = ((b*F + f)*P + px)*groups + g."""
    groups = C >> 3
    out = np.zeros_like(x)
    n = B * F * P * groups
    frame = P * C
    first = -2 if causal else -1
    for i in range(n):
        g = i % groups
        pos = i // groups
        f = (pos // P) % F
        xo = pos * C + g * 8
        acc = x[xo:xo + 8].astype(np.float32) + bias[g * 8:g * 8 + 8]
        for k in range(3):
            ff = f + first + k
            if ff < 0 or ff >= F:
                continue
            v = x[xo + (first + k) * frame: xo + (first + k) * frame + 8].astype(np.float32)
            acc = acc + w[(g * 8 + np.arange(8)) * 3 + k] * v
        out[xo:xo + 8] = acc.astype(np.float16)
    return out


def attention_model(qkv, null_kv, q_scale, k_scale, bias, B, F, P, heads, ld, ld_o, causal, scale):
    """temporal_attention_kernel: one wave per item = b*P + px, lane = head dimension."""
    lane = np.arange(64)
    o = np.zeros(B * F * P * ld_o, dtype=np.float16)
    J = F + 1
    inner = heads * 64
    for item in range(B * P):
        b, px = item // P, item % P
        base = (b * F * P + px) * ld
        fstride = P * ld
        ks, qs = k_scale[lane], q_scale[lane] * scale
        kh = np.zeros((F + 1, 64), np.float32)
        vv = np.zeros((F + 1, 64), np.float32)
        nk, nv = null_kv[lane], null_kv[64 + lane]
        kh[0] = nk * (1.0 / max(np.sqrt((nk * nk).sum()), 1e-12)) * ks
        vv[0] = nv
        for j in range(F):
            row = base + j * fstride + inner
            k = qkv[row + lane].astype(np.float32)
            v = qkv[row + 64 + lane].astype(np.float32)
            kh[1 + j] = k * (1.0 / max(np.sqrt((k * k).sum()), 1e-12)) * ks
            vv[1 + j] = v
        obase = (b * F * P + px) * ld_o
        ostride = P * ld_o
        for h in range(heads):
            for i in range(F):
                q = qkv[base + i * fstride + h * 64 + lane].astype(np.float32)
                qn = q * (1.0 / max(np.sqrt((q * q).sum()), 1e-12)) * qs
                last = i + 1 if causal else F
                mx, den, acc = np.float32(-3.0e38), np.float32(0), np.zeros(64, np.float32)
                for j in range(last + 1):
                    s = np.float32((qn * kh[j]).sum() + bias[(h * F + i) * J + j])
                    mn = max(mx, s)
                    c, e = np.exp(mx - mn), np.exp(s - mn)
                    den = den * c + e
                    acc = acc * c + e * vv[j]
                    mx = mn
                o[obase + i * ostride + h * 64 + lane] = (acc / den).astype(np.float16)
    return o


@pytest.mark.parametrize("causal", [True, False])
def test_temporal_peg_kernel_model(causal):
    torch.manual_seed(0)
    R, Fr, S, C = 2, 4, 3, 16
    x = ops.new_act(R * Fr, S, S, C, "cpu")
    x.t.copy_(torch.randn(x.t.shape).half())
    out = ops.new_act(R * Fr, S, S, C, "cpu")
    w, b = torch.randn(C, 3), torch.randn(C)
    plan = ops.Plan()
    ops.temporal_peg(plan, x, w, b, out, B=R, F=Fr, causal=causal)
    Interpreter().run(plan)
    model = peg_model(x.t.numpy().reshape(-1), w.numpy().reshape(-1), b.numpy(), R, Fr, S * S, C, causal)
    assert np.abs(model.astype(np.float32) - out.t.numpy().reshape(-1).astype(np.float32)).max() < 2e-2   # one fp16 ulp at |x| ~ 8


@pytest.mark.parametrize("Fr,causal", [(4, True), (5, False)])
def test_temporal_attention_kernel_model(Fr, causal):
    torch.manual_seed(1)
    R, P, heads = 2, 5, 2
    rows = R * Fr * P
    qkv = ops.new_act(1, 1, rows, heads * 64 + 128, "cpu")
    qkv.t.copy_(torch.randn(qkv.t.shape).half())
    o = ops.new_act(1, 1, rows, heads * 64, "cpu", zero=True)
    nkv, qs, ks, bias = torch.randn(2, 64), torch.rand(64) + 0.5, torch.rand(64) + 0.5, torch.randn(heads, Fr, Fr + 1)
    plan = ops.Plan()
    ops.temporal_attention(plan, qkv, nkv, qs, ks, bias, o, B=R, F=Fr, P=P, heads=heads, causal=causal, scale=8.0)
    Interpreter().run(plan)
    model = attention_model(qkv.t.numpy().reshape(-1), nkv.numpy().reshape(-1), qs.numpy(), ks.numpy(), bias.numpy().reshape(-1), R, Fr, P, heads,
                            qkv.ld, o.ld, causal, 8.0)
    assert np.abs(model.astype(np.float32) - o.t.numpy().reshape(-1).astype(np.float32)).max() < 4e-3
