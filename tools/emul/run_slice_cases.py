#!/usr/bin/env python
"""Channel-slice outputs on an emulated kernel library (IMAGEN_LIB_PATH): a fused Block conv (ChanRMSNorm statistics -> gain -> SiLU -> 3x3)
writing `Cout` channels at an offset into a wider NHWC tensor (output stride ldy != Cout) — the store pattern of the UpsampleCombiner plan
(engine.py: _combine_upsample_fmaps), which no GPU test had exercised when it was written.  Checks the values and that the neighbouring channels
stay untouched, for the wave-specialised and the streaming family."""
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, torch.nn.functional as F
torch.cuda.synchronize = lambda *a, **k: None
from imagen_pytorch_amd import ops
from imagen_pytorch_amd.ops import Act
ops.current_stream_handle = lambda: 0
dev = torch.device("cpu")
g = torch.Generator().manual_seed(0)
def case(B, H, W, Cin, Cout, Ctot, off, cfg=None):
    x = (torch.randn(B, Cin, H, W, generator=g)).half().float()
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(9 * Cin)
    b = torch.randn(Cout, generator=g) * 0.1
    gamma = 1 + 0.2 * torch.randn(Cin, generator=g)
    ref = F.conv2d(F.silu(F.normalize(x, dim=1) * math.sqrt(Cin) * gamma.view(1, -1, 1, 1)), w, b, padding=1)
    xa = ops.act_from_nchw(x)
    pw = ops.pack_weight(w, b, dev)
    cat = ops.new_act(B, H, W, Ctot, dev); cat.t.fill_(7.0)
    dst = Act(cat.t, B, H, W, Cout, Ctot, H * W * Ctot, off)
    ssq = (x * x).sum(1).reshape(-1).contiguous()
    pa = torch.zeros(pw.Cin_pad); pa[:Cin] = gamma * math.sqrt(Cin)
    plan = ops.Plan("slice")
    p = ops.igemm(plan, xa, pw, dst, ssq_a=ssq, pa=pa, pstride=0, act_in=ops.ACT_SILU, cfg=cfg)
    plan.run()
    out = cat.t.reshape(B, H, W, Ctot)
    got = out[..., off:off + Cout].permute(0, 3, 1, 2).float()
    err = float((got - ref).norm() / ref.norm())
    untouched = bool((out[..., :off] == 7.0).all() and (out[..., off + Cout:] == 7.0).all())
    fam = ops.cfg_table()[p.cfg][3]
    print(f"Cin {Cin:3d} -> {Cout} into [{off}:{off+Cout}) of {Ctot} @{H}x{W}  family {fam} cfg {(p.cfg, p.TH, p.TW)}  err {err:.2e}  neighbours untouched: {untouched}")
    assert err < 1e-3 and untouched
case(2, 16, 16, 256, 32, 160, 32)
case(2, 16, 16, 128, 32, 160, 64)
case(2, 16, 16, 64, 32, 160, 96)
case(2, 48, 48, 32, 32, 160, 128)                      # 2*3*3 = 18 tiles < STREAM_MIN_TILES: wave-specialised family
case(1, 16, 16, 16, 8, 24, 8)                          # the tiny test unets: 8-channel slices
sid = ops.stream_cfg()
case(2, 40, 36, 32, 32, 160, 128, cfg=(sid, 16, 16))   # the streaming family (what the planner picks for a 32-channel map at 256^2)
