#!/usr/bin/env python
"""Per-launch time of small ops INSIDE a hipGraph chain (the way the sampler runs them): N copies of one op captured as one graph,
replayed; time / N is the op's cost including the dependent-dispatch overhead that a stand-alone event-timed launch does not show.

    python tools/latency_probe.py [filter]
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from imagen_pytorch_amd import ops

dev = torch.device("cuda:0")
N = 40


def timed(build, name):
    plan = ops.Plan(name)
    build(plan)
    one = list(plan.ops)
    for _ in range(N - 1):
        plan.ops.extend(one)
    plan._arr = None
    stream = torch.cuda.Stream()
    with torch.cuda.stream(stream):
        plan.run()
        torch.cuda.synchronize()
        g = ops.Graph(plan, stream)
        for _ in range(3):
            g.launch()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        R = 10
        e0.record(stream)
        for _ in range(R):
            g.launch()
        e1.record(stream)
        torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / (R * N * len(one))
    print(f"{name:58s} {us:7.2f} us/launch  ({len(one)} launch(es) per op)", flush=True)
    return us


def conv(B, H, W, C1, C2, Cout, K, *, pro=False, addend=False, rows_gemm=False, cfg=None, gca=False, post=False, ssq=False):
    def build(plan):
        x1 = ops.new_act(B, H, W, C1, dev); x1.t.normal_()
        x2 = None
        if C2:
            x2 = ops.new_act(B, H, W, C2, dev); x2.t.normal_()
        C = C1 + C2
        pw = ops.pack_weight(torch.randn(Cout, C, K, K) / (C * K * K) ** 0.5, torch.zeros(Cout), dev)
        y = ops.new_act(B, H, W, Cout, dev)
        kw = {}
        if pro:
            kw.update(rs=torch.rand(B * H * W, device=dev) + 0.5, pa=torch.rand(B, pw.Cin_pad, device=dev) + 0.5,
                      ps=torch.rand(B, pw.Cin_pad, device=dev), pstride=pw.Cin_pad, act_in=ops.ACT_SILU)
        if addend:
            kw.update(addend=ops.new_act(B, H, W, Cout, dev, zero=True), gate=torch.rand(B, Cout, device=dev))
        if ssq:
            kw.update(ssq_out=torch.zeros(B * H * W, device=dev))
        if gca:
            kw.update(gca=dict(wk=torch.randn(Cout, device=dev), bk=0.1))
        if post:
            kw.update(post=dict(pa=torch.rand(B, Cout, device=dev), ps=torch.rand(B, Cout, device=dev), pstride=Cout), ssq_out=torch.zeros(B * H * W, device=dev))
        p = ops.igemm(plan, x1, pw, y, x2=x2, cfg=cfg, **kw)
        build.desc = f"cfg{p.cfg}{ops.cfg_table()[p.cfg]} t{p.TH}x{p.TW}"
    return build


def main():
    flt = sys.argv[1] if len(sys.argv) > 1 else ""
    cases = []

    def add(name, b):
        if flt in name:
            cases.append((name, b))

    add("memset32 (floor)", lambda plan: ops.memset32(plan, torch.zeros(64, dtype=torch.int32, device=dev), 0))

    def rowstat(B, H, W, C):
        def b(plan):
            x = ops.new_act(B, H, W, C, dev); x.t.normal_()
            ops.rowstat(plan, x, mode=2, rs=torch.zeros(B * H * W, device=dev))
        return b
    add("rowstat 16x8x8x256", rowstat(16, 8, 8, 256))
    add("rowstat 16x64x64x32", rowstat(16, 64, 64, 32))

    def gca_final(C, chunks):
        def b(plan):
            hid = max(3, C // 2)
            part = torch.rand(16, chunks, C + 2, device=dev)
            ops.gca_final(plan, part, torch.randn(C, hid, device=dev) * 0.1, torch.zeros(hid, device=dev), torch.randn(hid, C, device=dev) * 0.1,
                          torch.zeros(C, device=dev), torch.zeros(16, C, device=dev), B=16, C=C, chunks=chunks)
        return b
    add("gca_final C=256 chunks=8", gca_final(256, 8))
    add("gca_final C=128 chunks=32", gca_final(128, 32))
    add("gca_final C=32 chunks=64", gca_final(32, 64))

    def gca_full(H, C):
        def b(plan):
            h = ops.new_act(16, H, H, C, dev); h.t.normal_()
            hid = max(3, C // 2)
            chunks = ops.gca_chunks(H * H, 16, C)
            ops.gca(plan, h, torch.randn(C, device=dev), 0.1, torch.randn(C, hid, device=dev) * 0.1, torch.zeros(hid, device=dev),
                    torch.randn(hid, C, device=dev) * 0.1, torch.zeros(C, device=dev), torch.zeros(16, chunks, C + 2, device=dev),
                    torch.zeros(16, C, device=dev), chunks)
        return b
    add("gca (partial+final) 8x8x256", gca_full(8, 256))
    add("gca (partial+final) 32x32x64", gca_full(32, 64))
    add("gca (partial+final) 64x64x128", gca_full(64, 128))

    def gate_res(H, C):
        def b(plan):
            h = ops.new_act(16, H, H, C, dev); r = ops.new_act(16, H, H, C, dev); o = ops.new_act(16, H, H, C, dev)
            ops.gate_residual(plan, h, torch.rand(16, C, device=dev), r, o, rs_out=torch.zeros(16 * H * H, device=dev), raw_ssq=True)
        return b
    add("gate_residual 16x8x8x256", gate_res(8, 256))
    add("gate_residual 16x64x64x32", gate_res(64, 32))

    def attn(B, heads, rows, J):
        def b(plan):
            D = 64
            Jp = (J + 31) // 32 * 32
            q = torch.randn(B, rows, heads, D, device=dev).half()
            khat = torch.nn.functional.normalize(torch.randn(B, heads, Jp, D, device=dev), dim=-1).half()
            vt = torch.randn(B, heads, D, Jp, device=dev).half()
            o = torch.empty(B, rows, heads, D, dtype=torch.float16, device=dev)
            ops.attention(plan, q, khat, vt, o, B=B, heads=heads, rows=rows, J=J, q_strides=(rows * heads * D, D, heads * D),
                          k_strides=(heads * Jp * D, Jp * D, D), vt_strides=(heads * D * Jp, D * Jp, Jp), o_strides=(rows * heads * D, D, heads * D),
                          q_scale=torch.ones(D, device=dev), q_mult=8 * ops.LOG2E)
        return b
    add("attn self 1024 tok (B16, rows 8192, J 1065) 35.7 GF", attn(16, 1, 8192, 1065))
    add("attn self 256 tok (B16, rows 2048, J 297)", attn(16, 1, 2048, 297))
    add("attn self 64 tok (B16, rows 512, J 105)", attn(16, 1, 512, 105))
    add("attn cross 1024 tok (B16 x 8 heads, rows 1024, J 41)", attn(16, 8, 1024, 41))

    for nm, args, kws in [
        ("conv 128->128 k3 @8 raw", (16, 8, 8, 128, 0, 128, 3), {}),
        ("conv 128->128 k3 @8 pro", (16, 8, 8, 128, 0, 128, 3), dict(pro=True)),
        ("conv 128->128 k3 @8 pro+post", (16, 8, 8, 128, 0, 128, 3), dict(pro=True, post=True)),
        ("conv 128->128 k3 @8 raw+gca", (16, 8, 8, 128, 0, 128, 3), dict(gca=True)),
        ("conv 256->256 k3 @8 raw", (16, 8, 8, 256, 0, 256, 3), {}),
        ("conv 384->256 k3 @8 pro", (16, 8, 8, 256, 128, 256, 3), dict(pro=True)),
        ("conv 64->64 k3 @16 raw", (16, 16, 16, 64, 0, 64, 3), {}),
        ("conv 64->64 k3 @16 pro", (16, 16, 16, 64, 0, 64, 3), dict(pro=True)),
        ("conv 32->32 k3 @64 raw", (16, 64, 64, 32, 0, 32, 3), {}),
        ("conv 32->32 k3 @64 pro", (16, 64, 64, 32, 0, 32, 3), dict(pro=True)),
        ("conv 128->128 k3 @32 raw", (16, 32, 32, 128, 0, 128, 3), {}),
        ("conv 64->64 k3 @64 raw", (16, 64, 64, 64, 0, 64, 3), {}),
        ("res_conv 384->256 k1 @8 addend", (16, 8, 8, 256, 128, 256, 1), dict(addend=True)),
        ("res_conv 192->128 k1 @16 addend", (16, 16, 16, 128, 64, 128, 1), dict(addend=True)),
        ("res_conv 64->32 k1 @256 addend", (16, 256, 256, 32, 32, 32, 1), dict(addend=True)),
        ("res_conv 96->64 k1 @128 addend", (16, 128, 128, 64, 32, 64, 1), dict(addend=True)),
        ("res_conv 192->128 k1 @64 addend", (16, 64, 64, 128, 64, 128, 1), dict(addend=True)),
        ("res_conv 384->256 k1 @32 addend", (16, 32, 32, 256, 128, 256, 1), dict(addend=True)),
        ("lin 128->128 rows 16 (to_time_cond)", (1, 1, 16, 128, 0, 128, 1), {}),
        ("lin 256->128 rows 16x64 pro (ff.lin2)", (16, 1, 64, 256, 0, 128, 1), dict(pro=True)),
        ("lin 256->128 rows 16x64 raw", (16, 1, 64, 256, 0, 128, 1), {}),
        ("lin 32->64 rows 16x1024 pro (ff.lin1)", (16, 1, 1024, 32, 0, 64, 1), dict(pro=True)),
        ("lin 64->640 rows 16x1024 pro (qkv)", (16, 1, 1024, 64, 0, 640, 1), dict(pro=True)),
    ]:
        add(nm, conv(*args, **kws))
    single = {}
    for name, b in cases:
        try:
            single[name] = timed(b, name)
        except Exception as e:  # noqa: BLE001
            print(f"{name:58s} FAILED: {e}", flush=True)
    # the same ops interleaved (A, B, C, ..., A, B, C, ...): every launch follows a DIFFERENT kernel, as in the denoiser step — the
    # difference to the mean of the homogeneous chains is what switching kernels costs (instruction cache, argument / weight lines)
    mix = [(n, b) for n, b in cases if n.startswith(("conv", "lin", "res_conv")) and n in single]
    if len(mix) > 3:
        def build_mix(plan):
            for _, b in mix:
                b(plan)
        us = timed(build_mix, f"MIX of {len(mix)} igemm ops (per launch)")
        print(f"   mean of their homogeneous chains: {sum(single[n] for n, _ in mix) / len(mix):.2f} us/launch")


def separate():
    """What does a kernel SWITCH cost, and is it code or data?  One shape (64->64 3x3 @16x16, B16):
       P0 one launch repeated                      (code hot, data hot)
       P1 the same instantiation over 24 buffer sets, round-robin   (code hot, data not in L2)
       P2 different instantiations (tile cfgs x plain/generic epilogue), ONE small buffer set each, round-robin  (code switches)"""
    shape = (16, 16, 16, 64, 0, 64, 3)
    p0 = timed(conv(*shape, cfg=None), "P0 conv 64->64 k3 @16, one launch repeated")

    def many(plan):
        for _ in range(24):
            conv(*shape)(plan)
    timed(many, "P1 same instantiation, 24 buffer sets round-robin")
    tab = ops.cfg_table()
    variants = []
    for cid, (tp, bn, g, fam) in enumerate(tab):
        if fam != 0 or g != 4:
            continue
        for th, tw in ops._tile_shapes(tp, 16, 16):
            if ops.load_library().imagen_igemm_lds_bytes(cid, 3, 3, 1, th, tw) > 0:
                for add in (False, True):
                    variants.append(((cid, th, tw), add))
                break
    singles = []
    for cfg, add in variants:
        try:
            singles.append(timed(conv(*shape, cfg=cfg, addend=add), f"   homogeneous cfg{cfg} {'generic' if add else 'plain'} epilogue"))
        except Exception as e:  # noqa: BLE001
            print("   skipped", cfg, add, e)

    def mixed(plan):
        for cfg, add in variants:
            conv(*shape, cfg=cfg, addend=add)(plan)
    timed(mixed, f"P2 {len(variants)} instantiations round-robin (per launch)")
    print(f"   mean of their homogeneous chains: {sum(singles) / max(len(singles), 1):.2f} us/launch")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--separate":
        separate()
    else:
        main()
