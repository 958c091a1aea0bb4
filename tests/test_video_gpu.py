"""MI355X tests of the Imagen-Video path (SURVEY.md §8(f) NEXT-2).  Default-on since round 2 (first GPU run: gpurun_out of
round-2 call gpu_r2_a.sh — both temporal kernels, the Unet3D forward and the two video samplers passed as written; the one failure was
this file's own view-pattern test reading its outputs before the plan had run).  IMAGEN_VIDEO_GPU_TESTS=0 switches the file off.

The two temporal kernels are compared stand-alone with the ORACLE's functions (oracle/unet3d_oracle.py: temporal_peg, attention3d +
dynamic_position_bias) and, as a second opinion, with tests/plan_interp.py's restatement of their contract (include/imagen_hip.h); the model
test compares Unet3D.forward with the reference fixture (tests/golden/unet3d_tiny.pt), bar as for the image Unet.
"""
import os
import sys

import pytest
import torch
import torch.nn.functional as F

from conftest import gpu_device

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(os.environ.get("IMAGEN_VIDEO_GPU_TESTS") == "0",
                                                  reason="video path switched off with IMAGEN_VIDEO_GPU_TESTS=0")]
GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def nerr(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return float((a - b).norm() / b.norm())


def _both(build):
    """Run the same op list on the GPU (HIP) and on the CPU (plan interpreter); return the two output tensors."""
    from imagen_pytorch_amd import ops
    from plan_interp import Interpreter

    outs = []
    for leg, dev in (("kernels", gpu_device()), ("interpreter", torch.device("cpu"))):   # (IMAGEN_EMUL_TESTS=1: the kernels run on the CPU emulation)
        torch.manual_seed(0)
        plan = ops.Plan()
        out = build(plan, dev)
        if leg == "interpreter":
            it = Interpreter()
            it.run(plan)
        else:
            plan.run()
            if dev.type == "cuda":
                torch.cuda.synchronize()
        if isinstance(out, (list, tuple)):
            out = torch.cat([t.flatten() for t in out])
        outs.append(out.float().cpu())
    return outs


def test_igemm_view_patterns_of_the_video_planner():
    """The IGEMM argument patterns only the video planner uses: frame-shifted 1x1 GEMMs accumulating in place through `res == y`
    (temporal conv), a two-input GEMM over even / odd frames with batch stride 2 frames (temporal down-sampling), and one GEMM per phase
    writing every r-th frame (temporal pixel-shuffle)."""
    from imagen_pytorch_amd import ops
    from imagen_pytorch_amd.ops import Act

    def build(plan, dev):
        ops.KEEP_REFERENCE_WEIGHTS = dev.type == "cpu"
        try:
            R, f, P, C, Co = 2, 6, 64, 32, 32
            x = ops.new_act(R * f, 8, 8, C, dev)
            x.t.copy_(torch.randn(x.t.shape).half())
            w = torch.randn(Co, C, 3) / (3 * C) ** 0.5
            bias = torch.randn(Co)
            y = ops.new_act(R * f, 8, 8, Co, dev, zero=True)
            for shift in range(3):                                  # engine3d._temporal_conv
                pw = ops.pack_weight(w[:, :, 2 - shift], bias if shift == 0 else None, dev)
                n = f - shift
                xin = Act(x.t, R, n, P, C, C, f * P * C, 0)
                yout = Act(y.t, R, n, P, Co, Co, f * P * Co, shift * P * Co)
                ops.igemm(plan, xin, pw, yout, res=yout if shift else None)
            fr = P * Co                                              # engine3d._temporal_down
            wd = torch.randn(16, 2 * Co) / (2 * Co) ** 0.5
            pwd = ops.pack_weight(torch.cat((wd[:, 0::2], wd[:, 1::2]), dim=1), torch.randn(16), dev)
            even = Act(y.t, R * f // 2, 8, 8, Co, Co, 2 * fr, 0)
            odd = Act(y.t, R * f // 2, 8, 8, Co, Co, 2 * fr, fr)
            d = ops.new_act(R * f // 2, 8, 8, 16, dev)
            ops.igemm(plan, even, pwd, d, x2=odd)
            wu, bu = torch.randn(2 * 24, 16) / 4.0, torch.randn(2 * 24)   # engine3d._temporal_up
            up = ops.new_act(R * f, 8, 8, 24, dev, zero=True)
            for j in range(2):
                pwu = ops.pack_weight(wu[j::2], bu[j::2], dev)
                ops.igemm(plan, d, pwu, Act(up.t, R * f // 2, 8, 8, 24, 24, 2 * P * 24, j * P * 24), act_out=ops.ACT_SILU)
            return [y.t, d.t, up.t]   # (read AFTER the plan has run)
        finally:
            ops.KEEP_REFERENCE_WEIGHTS = False

    hip, ref = _both(build)
    assert ref.abs().sum() > 0 and nerr(hip, ref) < 2e-3


@pytest.mark.parametrize("f,P,C,Co", [(6, 64, 32, 32), (16, 40, 64, 96), (1, 64, 32, 32), (2, 24, 128, 64), (5, 72, 64, 64)])
def test_causal_temporal_conv_in_one_launch(f, P, C, Co):
    """Round 6 (ABI 10, ImagenIgemmParams.pad_x1): Imagen-Video's causal Conv1d(k = 3) over the frames of every pixel (iv.py:436-449) as ONE igemm
    launch — a 3 x 1 window over rows = frames in the (clip, frame, pixel) view, two zero rows in front of every clip, none behind, no x padding
    (ops.igemm(causal_rows=True), kernel family 0) — against torch's conv1d on identical fp16 inputs and against the plan interpreter: clips of 1, 2
    (fewer frames than taps), 5, 6 and 16 frames, pixel counts that do not fill the tiles, three channel widths; the rows of one clip must never see
    the previous clip's last frames."""
    from imagen_pytorch_amd import ops
    from imagen_pytorch_amd.ops import Act

    R = 3
    g = torch.Generator().manual_seed(f * 131 + P)
    x_ref = (torch.randn(R, f, P, C, generator=g) * 0.8).half()
    w = torch.randn(Co, C, 3, generator=g) / (3 * C) ** 0.5
    bias = torch.randn(Co, generator=g) * 0.1

    def build(plan, dev):
        ops.KEEP_REFERENCE_WEIGHTS = dev.type == "cpu"
        try:
            x = ops.new_act(R * f, 1, P, C, dev)
            x.t.copy_(x_ref.reshape(x.t.shape))
            y = ops.new_act(R * f, 1, P, Co, dev)
            y.t.fill_(float("nan"))
            pw = ops.pack_weight(w.unsqueeze(-1), bias, dev)
            p = ops.igemm(plan, Act(x.t, R, f, P, C, C, f * P * C, 0), pw, Act(y.t, R, f, P, Co, Co, f * P * Co, 0), causal_rows=True)
            assert ops.cfg_table()[p.cfg][3] == 0 and (p.KH, p.KW, p.pad, p.pad_x1, p.OH, p.OW) == (3, 1, 2, 1, f, P)
            return y.t
        finally:
            ops.KEEP_REFERENCE_WEIGHTS = False

    hip, interp = _both(build)
    xt = x_ref.float().permute(0, 2, 3, 1).reshape(R * P, C, f)                       # (clip x pixel, channel, frame)
    ref = F.conv1d(F.pad(xt, (2, 0)), w.half().float(), bias).reshape(R, P, Co, f).permute(0, 3, 1, 2)
    e_hip, e_int = nerr(hip.reshape(ref.shape), ref), nerr(interp.reshape(ref.shape), ref)
    assert e_hip < 1e-3 and e_int < 1e-3, (e_hip, e_int)


@pytest.mark.parametrize("causal", [True, False])
def test_temporal_peg_kernel(causal):
    from imagen_pytorch_amd import ops

    def build(plan, dev):
        R, Fr, S, C = 2, 5, 8, 24
        x = ops.new_act(R * Fr, S, S, C, dev)
        x.t.copy_(torch.randn(x.t.shape).half())
        out = ops.new_act(R * Fr, S, S, C, dev)
        ops.temporal_peg(plan, x, torch.randn(C, 3).to(dev), torch.randn(C).to(dev), out, B=R, F=Fr, causal=causal)
        return out.t

    hip, ref = _both(build)
    assert nerr(hip, ref) < 1e-3


@pytest.mark.parametrize("Fr,causal", [(4, True), (16, True), (7, False)])
def test_temporal_attention_kernel(Fr, causal):
    from imagen_pytorch_amd import ops

    def build(plan, dev):
        R, P, heads = 2, 37, 3
        rows = R * Fr * P
        qkv = ops.new_act(1, 1, rows, heads * 64 + 128, dev)
        qkv.t.copy_(torch.randn(qkv.t.shape).half())
        o = ops.new_act(1, 1, rows, heads * 64, dev, zero=True)
        ops.temporal_attention(plan, qkv, torch.randn(2, 64).to(dev), (torch.rand(64) + 0.5).to(dev), (torch.rand(64) + 0.5).to(dev),
                               torch.randn(heads, Fr, Fr + 1).to(dev), o, B=R, F=Fr, P=P, heads=heads, causal=causal, scale=8.0)
        return o.t

    hip, ref = _both(build)
    assert nerr(hip, ref) < 2e-3


def test_temporal_attention_null_value_is_not_rounded():
    """The null value of the temporal attention is an fp32 parameter shared by every pixel: rounding it to fp16 (as V^T column 0 of the MFMA
    kernel did in round 4) is not noise but ONE error vector in every row, weighted by the null key's softmax weight — about a half in the
    first frame of a causal clip.  Normwise it hides under the 2.8e-4 fp16 rounding of the output; averaged over the pixels it does not:
    the mean error of the first frame's output channels has to be the averaged-down rounding noise, not (null value - fp16(null value)) / 2.
    (On the C5 denoiser that bias was worth 1.00e-3 -> 1.05e-3 of distance to the oracle.)"""
    from imagen_pytorch_amd import ops

    R, Fr, P, heads = 2, 4, 256, 2
    keep = {}

    def build(plan, dev):
        rows = R * Fr * P
        qkv = ops.new_act(1, 1, rows, heads * 64 + 128, dev)
        qkv.t.copy_(torch.randn(qkv.t.shape).half())
        o = ops.new_act(1, 1, rows, heads * 64, dev, zero=True)
        null_kv = torch.randn(2, 64) * 4.0 + 0.37          # (fp16 spacing of 2^-9 .. 2^-8 over most of the vector)
        keep["nv"] = null_kv[1].clone()
        ops.temporal_attention(plan, qkv, null_kv.to(dev), torch.ones(64).to(dev), torch.ones(64).to(dev),
                               torch.zeros(heads, Fr, Fr + 1).to(dev), o, B=R, F=Fr, P=P, heads=heads, causal=True, scale=1.0)
        return o.t

    hip, ref = _both(build)
    assert nerr(hip, ref) < 1e-3
    d = (hip - ref).reshape(R, Fr, P, heads, 64)[:, 0].mean(dim=(0, 1, 2))      # first frame: keys = null + itself, weights ~ 1/2 each
    lost = keep["nv"] - keep["nv"].half().float()
    assert lost.norm() > 1e-3                                                  # the case does exercise the rounding
    # the kernel that rounds the null value shows d ~ lost / 2 (ratio ~ 0.5); the averaged-down output rounding alone is ~ 0.05 of |lost|
    assert float(d.norm() / lost.norm()) < 0.15, float(d.norm() / lost.norm())


@pytest.mark.parametrize("causal", [True, False])
def test_temporal_peg_kernel_vs_oracle(causal):
    """TEMPORAL_PEG alone against the ORACLE's temporal_peg (oracle/unet3d_oracle.py, iv.py:1413-1414: F.pad + depthwise Conv3d (3, 1, 1) +
    residual) — not against the contract interpreter, which is this repo's own restatement of the kernel."""
    from imagen_pytorch_amd import ops
    from oracle import unet3d_oracle as u3
    from oracle.unet_oracle import _SD

    dev = gpu_device()
    g = torch.Generator().manual_seed(3)
    R, Fr, S, C = 2, 6, 8, 40
    x = torch.randn(R, C, Fr, S, S, generator=g).half().float()
    sd = {"fn.1.weight": torch.randn(C, 1, 3, 1, 1, generator=g) * 0.5, "fn.1.bias": torch.randn(C, generator=g) * 0.2}
    ref = u3.temporal_peg(_SD(sd), x, causal=causal)                                   # (R, C, F, S, S)
    xa = ops.new_act(R * Fr, S, S, C, dev)
    xa.t.copy_(x.permute(0, 2, 3, 4, 1).reshape(R * Fr, S, S, C).half())             # frame-major clips: (b, f) consecutive NHWC frames
    out = ops.new_act(R * Fr, S, S, C, dev)
    plan = ops.Plan()
    ops.temporal_peg(plan, xa, sd["fn.1.weight"].reshape(C, 3).contiguous().to(dev), sd["fn.1.bias"].to(dev), out, B=R, F=Fr, causal=causal)
    plan.run()
    torch.cuda.synchronize()
    got = out.t.float().cpu().reshape(R, Fr, S, S, C).permute(0, 4, 1, 2, 3)
    assert nerr(got, ref) < 1e-3


@pytest.mark.parametrize("Fr,causal", [(4, True), (16, True), (7, False)])
def test_temporal_attention_kernel_vs_oracle(Fr, causal):
    """TEMPORAL_ATTENTION alone against the ORACLE's attention3d (iv.py:499-570 along the frame axis, iv.py:257-270): the q | k | v rows are
    the oracle-side projections of the normalised sequence (fp32 torch, rounded to the kernel's fp16 storage), the bias table is the oracle's
    dynamic_position_bias + null_attn_bias; `to_out` is the identity so that the oracle's output is LayerNorm(kernel output) * g — the
    kernel's own product is what is compared."""
    import torch.nn.functional as F
    from imagen_pytorch_amd import ops
    from oracle import unet3d_oracle as u3
    from oracle.unet_oracle import _SD, gain_layernorm

    dev = gpu_device()
    g = torch.Generator().manual_seed(4)
    R, P, heads, dh = 2, 37, 3, 64
    C = heads * dh
    rn = lambda *s: torch.randn(*s, generator=g)
    sd = {"norm.g": 1 + 0.1 * rn(C), "to_q.weight": rn(C, C) / C ** 0.5, "to_kv.weight": rn(2 * dh, C) / C ** 0.5, "null_kv": rn(2, dh),
          "q_scale": torch.rand(dh, generator=g) + 0.5, "k_scale": torch.rand(dh, generator=g) + 0.5, "null_attn_bias": rn(heads),
          "to_out.0.weight": torch.eye(C), "to_out.1.g": 1 + 0.1 * rn(C),
          "rel_pos_bias.mlp.0.0.weight": rn(16, 1), "rel_pos_bias.mlp.0.0.bias": rn(16) * 0.1, "rel_pos_bias.mlp.0.1.g": torch.ones(16),
          "rel_pos_bias.mlp.1.weight": rn(heads, 16) * 0.3, "rel_pos_bias.mlp.1.bias": rn(heads) * 0.1}
    p = _SD(sd)
    seq = rn(R * P, Fr, C)                                                             # one sequence of F frames per (clip, pixel)
    xn = gain_layernorm(seq, sd["norm.g"])
    q16, kv16 = F.linear(xn, sd["to_q.weight"]).half(), F.linear(xn, sd["to_kv.weight"]).half()
    # attention3d's arithmetic from its l2norm on, restated piece by piece on the ROUNDED projections (checked against the function itself below)
    q = q16.float().reshape(R * P, Fr, heads, dh).permute(0, 2, 1, 3)
    k = torch.cat((sd["null_kv"][0].expand(R * P, 1, dh), kv16[..., :dh].float()), dim=1)
    v = torch.cat((sd["null_kv"][1].expand(R * P, 1, dh), kv16[..., dh:].float()), dim=1)
    qh = F.normalize(q, dim=-1, eps=1e-12) * sd["q_scale"]
    kh = F.normalize(k, dim=-1, eps=1e-12) * sd["k_scale"]
    sim = torch.einsum("bhid,bjd->bhij", qh, kh) * 8.0
    bias = torch.cat((sd["null_attn_bias"].reshape(heads, 1, 1).expand(heads, Fr, 1), u3.dynamic_position_bias(p.sub("rel_pos_bias"), Fr)), dim=-1)
    sim = sim + bias
    if causal:
        sim = sim.masked_fill(torch.ones(Fr, Fr + 1, dtype=torch.bool).triu(2), -torch.finfo(sim.dtype).max)
    ref_o = torch.einsum("bhij,bjd->bhid", sim.softmax(dim=-1), v).permute(0, 2, 1, 3).reshape(R * P, Fr, C)
    # cross-check of the restated pieces against the oracle's own function on unrounded projections (same arithmetic, fp32 end to end)
    full = u3.attention3d(p, seq, None, causal)
    qf = F.linear(xn, sd["to_q.weight"]).reshape(R * P, Fr, heads, dh).permute(0, 2, 1, 3)
    kvf = F.linear(xn, sd["to_kv.weight"])
    kf = torch.cat((sd["null_kv"][0].expand(R * P, 1, dh), kvf[..., :dh]), dim=1)
    vf = torch.cat((sd["null_kv"][1].expand(R * P, 1, dh), kvf[..., dh:]), dim=1)
    simf = torch.einsum("bhid,bjd->bhij", F.normalize(qf, dim=-1, eps=1e-12) * sd["q_scale"], F.normalize(kf, dim=-1, eps=1e-12) * sd["k_scale"]) * 8.0 + bias
    if causal:
        simf = simf.masked_fill(torch.ones(Fr, Fr + 1, dtype=torch.bool).triu(2), -torch.finfo(simf.dtype).max)
    of = torch.einsum("bhij,bjd->bhid", simf.softmax(dim=-1), vf).permute(0, 2, 1, 3).reshape(R * P, Fr, C)
    assert nerr(gain_layernorm(of, sd["to_out.1.g"]), full) < 1e-5, "the pieces above must BE the oracle's attention3d"
    # ---- the kernel: rows (b, f, p) of q | k | v
    rows = R * Fr * P
    qkv = ops.new_act(1, 1, rows, C + 2 * dh, dev)
    packed = torch.cat((q16, kv16), dim=-1).reshape(R, P, Fr, C + 2 * dh).permute(0, 2, 1, 3).reshape(rows, C + 2 * dh)
    qkv.t.copy_(packed.reshape(qkv.t.shape))
    o = ops.new_act(1, 1, rows, C, dev, zero=True)
    plan = ops.Plan()
    ops.temporal_attention(plan, qkv, sd["null_kv"].contiguous().to(dev), sd["q_scale"].to(dev), sd["k_scale"].to(dev), bias.contiguous().to(dev), o,
                           B=R, F=Fr, P=P, heads=heads, causal=causal, scale=8.0)
    plan.run()
    torch.cuda.synchronize()
    got = o.t.float().cpu().reshape(R, Fr, P, C).permute(0, 2, 1, 3).reshape(R * P, Fr, C)
    assert nerr(got, ref_o) < 2e-3


@pytest.mark.parametrize("tag", ["base", "sr"])
def test_unet3d_forward_vs_reference_fixture(tag):
    from imagen_pytorch_amd import Unet3D

    dev = gpu_device()
    g = torch.load(os.path.join(GOLDEN, "unet3d_tiny.pt"), weights_only=False)["runs"][tag]
    u = Unet3D(**g["kwargs"]).eval()
    u.load_state_dict(g["state_dict"])
    u = u.to(dev)
    kw = dict(text_embeds=g["text_embeds"].to(dev), text_mask=g["text_mask"].to(dev), **{k: v.to(dev) for k, v in g["extra"].items()})
    x, t = g["x"].to(dev), g["time"].to(dev)
    e_c = nerr(u(x, t, **kw), g["out_cond"])
    e_n = nerr(u(x, t, cond_drop_prob=1.0, **kw), g["out_null"])
    e_i = nerr(u(x, t, ignore_time=True, **kw), g["out_notime"])
    cfg = u.forward_with_cond_scale(x, t, cond_scale=3.0, **kw)
    e_g = nerr(cfg, g["out_cfg"])
    print(f"unet3d[{tag}] vs reference: cond {e_c:.2e} null {e_n:.2e} ignore_time {e_i:.2e} cfg {e_g:.2e}")
    assert max(e_c, e_n, e_i) < 1e-2 and e_g < 2e-2


def test_video_cascade_sample_vs_reference_fixture():
    """Imagen.sample(video_frames=4) over two Unet3D stages vs the recorded reference run (same draws); graph == eager."""
    from imagen_pytorch_amd import Imagen, Unet3D

    dev = gpu_device()
    g = torch.load(os.path.join(GOLDEN, "sample_tiny_video.pt"), weights_only=False)
    unets = [Unet3D(**spec["kwargs"]).eval() for spec in g["unets"]]
    imagen = Imagen(unets, image_sizes=g["image_sizes"], timesteps=g["timesteps"], text_embed_dim=32, cond_drop_prob=0.1).to(dev)
    for u, spec in zip(imagen.unets, g["unets"]):
        u.load_state_dict(spec["state_dict"])
    noise_fn = lambda tag, shape: g["noise"][tag].to(dev)
    res = {}
    for use_graph in (False, True):
        outs = imagen.sample(text_embeds=g["text_embeds"].to(dev), video_frames=g["frames"], cond_scale=g["cond_scale"], use_tqdm=False,
                             return_all_unet_outputs=True, noise_fn=noise_fn, use_graph=use_graph)
        errs = [nerr(o, r) for o, r in zip(outs, g["outputs"])]
        print("video cascade", "graph" if use_graph else "eager", errs)
        assert all(o.shape == r.shape for o, r in zip(outs, g["outputs"])) and max(errs) < 2e-2
        res[use_graph] = outs
    assert all(torch.equal(a, b) for a, b in zip(res[False], res[True]))


def test_video_elucidated_sample_vs_reference_fixture():
    from imagen_pytorch_amd import ElucidatedImagen, Unet3D

    dev = gpu_device()
    g = torch.load(os.path.join(GOLDEN, "sample_tiny_video.pt"), weights_only=False)
    e = g["edm"]
    unets = [Unet3D(**spec["kwargs"]).eval() for spec in g["unets"]]
    model = ElucidatedImagen(unets, image_sizes=g["image_sizes"], text_embed_dim=32, cond_drop_prob=0.1, **e["hparams"]).to(dev).eval()
    for u, spec in zip(model.unets, g["unets"]):
        u.load_state_dict(spec["state_dict"])
    nf = lambda tag, shape: e["noise"][tag].to(dev)
    outs = model.sample(text_embeds=g["text_embeds"].to(dev), video_frames=g["frames"], cond_scale=g["cond_scale"], use_tqdm=False,
                        return_all_unet_outputs=True, noise_fn=nf)
    e0 = nerr(outs[0], e["outputs"][0])
    alone = model.sample(text_embeds=g["text_embeds"].to(dev), video_frames=g["frames"], cond_scale=g["cond_scale"], use_tqdm=False, noise_fn=nf,
                         start_at_unet_number=2, start_image_or_video=e["outputs"][0].to(dev))
    e1 = nerr(alone, e["outputs"][1])
    print(f"video EDM vs reference: stage1 {e0:.2e}, stage 2 alone {e1:.2e}")
    assert e0 < 5e-2 and e1 < 5e-2   # (Heun trajectories of dim-8 toy video unets, whose single forward is held to 1e-2: measured 1.5-3.2e-2 over the rounds;
                                     #  the per-step parity of the video path is test_unet3d_forward_vs_oracle_c5's)


@pytest.mark.parametrize("tag", ["cond_both", "cond_pre_tds", "init_skip", "inpaint"])
def test_video_sample_options_vs_reference_fixture(tag):
    """Video-stage options of Imagen.sample on the GPU — prompt frames before and after the clip (static slots of the packed input clip,
    per-step frame placement, final conv's own low-res frame order, output cut), prompt frames under a per-stage frame rate, init videos +
    skip_steps, video inpainting — vs recorded runs of the live reference (same draws); graph == eager."""
    from imagen_pytorch_amd import Imagen, Unet3D

    dev = gpu_device()
    o = torch.load(os.path.join(GOLDEN, "sample_tiny_video_options.pt"), weights_only=False)
    g = torch.load(os.path.join(GOLDEN, o["weights_from"]), weights_only=False)
    run = o["runs"][tag]
    unets = [Unet3D(**spec["kwargs"]).eval() for spec in g["unets"]]
    imagen = Imagen(unets, image_sizes=g["image_sizes"], timesteps=o["timesteps"], text_embed_dim=32, cond_drop_prob=0.1,
                    temporal_downsample_factor=run.get("temporal_downsample_factor", 1)).to(dev)
    for u, spec in zip(imagen.unets, g["unets"]):
        u.load_state_dict(spec["state_dict"])
    kw = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in run["kwargs"].items()}
    res = {}
    for use_graph in (False, True):
        outs = imagen.sample(text_embeds=g["text_embeds"].to(dev), video_frames=o["frames"], cond_scale=g["cond_scale"], use_tqdm=False,
                             return_all_unet_outputs=True, noise_fn=lambda t, shape: run["noise"][t].to(dev), use_graph=use_graph, **kw)
        errs = [nerr(a, r) for a, r in zip(outs, run["outputs"])]
        print("video options", tag, "graph" if use_graph else "eager", errs)
        assert all(a.shape == r.shape for a, r in zip(outs, run["outputs"])) and max(errs) < 2e-2
        res[use_graph] = outs
    assert all(torch.equal(a, b) for a, b in zip(res[False], res[True]))
    if tag == "inpaint":
        m = run["kwargs"]["inpaint_masks"][:, None].expand(-1, 3, -1, -1, -1)
        assert torch.allclose(res[True][-1].cpu()[m], run["kwargs"]["inpaint_videos"][m], atol=1e-6)


def test_unet3d_cond_images_vs_oracle():
    """Unet3D(cond_images_channels=5).forward_with_cond_scale on the GPU against the oracle (iv.py:1722-1731): the static second input of the
    init conv on every frame (green on MI355X since round 3's call A; CPU: tests/test_plan_interp.py)."""
    from imagen_pytorch_amd import Unet3D
    from oracle import unet3d_oracle as u3

    dev = gpu_device()
    g = torch.load(os.path.join(GOLDEN, "unet3d_tiny.pt"), weights_only=False)["runs"]["sr"]
    kw = {**g["kwargs"], "cond_images_channels": 5}
    torch.manual_seed(6)
    u = Unet3D(**kw).eval()
    sd = {k: v.clone() for k, v in u.state_dict().items()}
    for k, v in g["state_dict"].items():
        if sd[k].shape == v.shape:
            sd[k] = v.clone()
    u.load_state_dict(sd)
    u = u.to(dev)
    ci = torch.rand(g["x"].shape[0], 5, 8, 8)
    extra = {k: v for k, v in g["extra"].items()}
    got = u.forward_with_cond_scale(g["x"].to(dev), g["time"].to(dev), text_embeds=g["text_embeds"].to(dev), text_mask=g["text_mask"].to(dev),
                                    cond_images=ci.to(dev), cond_scale=3.0, **{k: v.to(dev) for k, v in extra.items()})
    with torch.no_grad():
        ref = u3.unet3d_forward_with_cond_scale(sd, kw, g["x"], g["time"], text_embeds=g["text_embeds"], text_mask=g["text_mask"], cond_images=ci,
                                                cond_scale=3.0, **extra)
    assert nerr(got, ref) < 1e-2, nerr(got, ref)


def _derandomise_unet3d(unet, seed=1234):
    """Unet3D starts as an image Unet applied per frame (zero final_conv, dirac temporal convs, zero out-norm gain of the temporal
    attentions: iv.py:1578, 415-417, 496-497): randomise the three so the temporal paths count (as oracle/make_golden.py does)."""
    g = torch.Generator().manual_seed(seed)
    for name, prm in unet.named_parameters():
        if name.startswith("final_conv."):
            prm.data.copy_(torch.randn(prm.shape, generator=g) * 0.05)
        elif ".temporal_conv." in name:
            prm.data.add_(torch.randn(prm.shape, generator=g) * (0.5 / (3 * prm.shape[1]) ** 0.5 if prm.ndim > 1 else 0.05))   # dirac + a dense perturbation
        elif name.endswith("fn.fn.to_out.1.g"):
            prm.data.copy_(1.0 + 0.2 * torch.randn(prm.shape, generator=g))


def _tap3d(act, rows):
    """Engine tap (frames as the batch: [rows * f, H, W, C] fp16 NHWC) -> (rows, C, f, H, W) fp32."""
    from imagen_pytorch_amd import ops

    t = ops.act_to_nchw(act)                       # (rows * f, C, H, W)
    f = t.shape[0] // rows
    return t.reshape(rows, f, *t.shape[1:]).permute(0, 2, 1, 3, 4)


C5_TOL = 1.0e-3    # north_star's bar.  Measured on MI355X, round 5 (calls Q, R): cond 9.64e-4, null 9.76e-4 (rounds 4 / 5 before: 1.05e-3 / 1.06e-3,
                   # asserted at 1.5e-3 / 1.15e-3).  Two causes, both found with the plan interpreter: the MFMA temporal attention of round 4
                   # rounded the fp32 null value to fp16 — one coherent error vector in every row, 1.05 -> 1.00e-3 (csrc/temporal.hip,
                   # test_temporal_attention_null_value_is_not_rounded above) — and the output stage's weights (final_conv, block1 of
                   # final_res_block) were fp16-rounded on the video path while the image path splits them: 1.00 -> 0.964e-3
                   # (engine3d.SPLIT_OUTPUT_STAGE).  The run is deterministic (same kernels, same draws): the 2.4 % headroom is not a noise margin.
C5_CFG_TOL = 3.1e-3   # the CFG-3 combination 3 e_cond - 2 e_null of two in-tolerance forwards (bound 5e-3; measured 2.77e-3, 2.7-2.8e-3 in every round)


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_unet3d_forward_vs_oracle_c5(seed):
    """(seed 0 is the case every round measured; seeds 1 / 2 are two more weight / input draws of the same configuration: VERDICT round 5.)
    BASELINE config C5's own denoiser — Unet3D(dim=64, dim_mults=(1, 2, 4, 8)) on one 16 x 64 x 64 clip, the CFG batch of 2 rows the
    sampler runs — against oracle/unet3d_oracle.py (iv.py:1650-1941), temporal layers de-identity-initialised, stage by stage: the taps
    'mid_peg' / 'mid_tattn' compare the two temporal kernels in place with the oracle's temporal_peg / temporal_attention."""
    from imagen_pytorch_amd import Unet3D
    from oracle import unet3d_oracle as u3

    dev = gpu_device()
    kw = dict(dim=64, dim_mults=(1, 2, 4, 8))
    torch.manual_seed(seed)
    u = Unet3D(**kw).eval()
    _derandomise_unet3d(u, seed=1234 + seed)
    sd = {k: v.clone() for k, v in u.state_dict().items()}
    x, t = torch.randn(1, 3, 16, 64, 64), torch.tensor([0.3])
    te = torch.randn(1, 24, 768)
    mask = torch.ones(1, 24, dtype=torch.bool)
    mask[0, 19:] = False
    taps = {}
    with torch.no_grad():
        ref = u3.unet3d_forward(sd, kw, x, t, text_embeds=te, text_mask=mask, taps=taps)
        ref_null = u3.unet3d_forward(sd, kw, x, t, text_embeds=te, text_mask=mask, cond_drop_prob=1.0)
    u = u.to(dev)
    args = dict(text_embeds=te.to(dev), text_mask=mask.to(dev))
    got = u(x.to(dev), t.to(dev), **args)
    eng = next(iter(u._engines.values()))
    rep = {k: nerr(_tap3d(a, 1)[:1], taps[k]) for k, a in eng.taps.items() if k in taps}
    print("c5 per-stage normwise error:", {k: f"{v:.1e}" for k, v in rep.items()})
    e = nerr(got, ref)
    e_null = nerr(u(x.to(dev), t.to(dev), cond_drop_prob=1.0, **args), ref_null)
    cfg = u.forward_with_cond_scale(x.to(dev), t.to(dev), cond_scale=3.0, **args)      # the sampler's 2-row plan
    e_cfg = nerr(cfg, ref_null + (ref - ref_null) * 3.0)
    print(f"c5 Unet3D(dim=64) 16x64x64 vs oracle: cond {e:.2e} null {e_null:.2e} cfg3 {e_cfg:.2e}")
    from conftest import record_parity
    record_parity("unet3d_forward_vs_oracle_c5" + (f"-seed{seed}" if seed else ""), cond=e, null=e_null, cfg3=e_cfg, taps=rep, tol=C5_TOL)
    assert {"mid_peg", "mid_tattn"} <= set(rep)
    assert e < C5_TOL and e_null < C5_TOL and e_cfg < C5_CFG_TOL, (e, e_null, e_cfg, rep)
