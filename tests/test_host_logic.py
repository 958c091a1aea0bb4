"""CPU (-m "not gpu"): host-side logic of the MI355X path — ABI mirror, symbol export, weight packing layout,
tile selection constraints, the planner (dry run, no launches), the drop-in surface, and the batch-sharded
multi-process path over gloo (world_size 2)."""
import ctypes
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def test_library_loads_and_exports_every_declared_symbol(lib):
    from imagen_pytorch_amd import _abi

    import re
    header = open(_abi.HEADER).read()
    declared = set(re.findall(r"\b(imagen_\w+)\s*\(", header))
    declared -= {"imagen_stream_t"}
    assert declared, "no entry points parsed from include/imagen_hip.h"
    for sym in declared:
        assert hasattr(lib, sym), f"libimagen_hip.so does not export {sym} declared in include/imagen_hip.h"
    assert declared == set(_abi.EXPORTED_SYMBOLS), "the binding's symbol list (and INTEGRATION.md's) must be exactly the header's entry points"
    assert lib.imagen_abi_version() == _abi.ENUMS["IMAGEN_ABI_VERSION"]


def test_struct_mirrors_match_c_sizes(lib):
    from imagen_pytorch_amd import _abi

    assert len(_abi.OP_STRUCT) == _abi.ENUMS["IMAGEN_OP_KIND_COUNT"] - 1
    for kind, st in _abi.OP_STRUCT.items():
        assert lib.imagen_sizeof(kind) == ctypes.sizeof(st) > 0


def test_launch_rejects_bad_arguments_without_a_gpu(lib):
    """Argument validation happens on the host before any launch: error code + message, no crash."""
    from imagen_pytorch_amd import _abi

    assert lib.imagen_launch(9999, ctypes.c_void_p(1), 0, None) != 0
    assert b"unknown op kind" in lib.imagen_last_error()
    p = _abi.STRUCTS["ImagenIgemmParams"]()
    p.cfg = 0
    assert lib.imagen_launch(_abi.ENUMS["IMAGEN_OP_IGEMM"], ctypes.addressof(p), ctypes.sizeof(p), None) != 0
    assert b"null" in lib.imagen_last_error()


def test_stale_struct_mirror_is_refused(lib):
    """ABI 7: every launch carries the caller's sizeof(params struct); a binding built against another version of the header (one field
    short here) is refused before anything is read, by imagen_launch and by imagen_plan_run alike."""
    from imagen_pytorch_amd import _abi

    K = _abi.ENUMS["IMAGEN_OP_IGEMM"]
    p = _abi.STRUCTS["ImagenIgemmParams"]()
    assert lib.imagen_launch(K, ctypes.addressof(p), ctypes.sizeof(p) - 8, None) != 0
    assert b"params struct" in lib.imagen_last_error()
    ref = _abi.OpRef(kind=K, params_bytes=ctypes.sizeof(p) - 8, params=ctypes.addressof(p))
    assert lib.imagen_plan_run(ctypes.addressof(ref), 1, None) != 0
    assert b"params struct" in lib.imagen_last_error()


def test_missing_library_fails_loudly(tmp_path):
    from imagen_pytorch_amd import _abi

    saved = _abi._lib
    _abi._lib = None
    try:
        with pytest.raises(_abi.ImagenHipError):
            _abi.load_library(str(tmp_path / "nope.so"))
    finally:
        _abi._lib = saved


def test_no_cpu_fallback():
    from imagen_pytorch_amd import Imagen, Unet

    u = Unet(dim=8, cond_dim=32, text_embed_dim=32, dim_mults=(1, 2), attn_heads=2, max_text_len=16, attn_pool_num_latents=8).eval()
    with pytest.raises(RuntimeError, match="no CPU path"):
        u(torch.randn(1, 3, 16, 16), torch.zeros(1), text_embeds=torch.randn(1, 4, 32))
    im = Imagen(u, image_sizes=16, timesteps=2, text_embed_dim=32)
    with pytest.raises(RuntimeError, match="no CPU path"):
        im.sample(text_embeds=torch.randn(1, 4, 32), use_tqdm=False)
    with pytest.raises(NotImplementedError):
        Unet(dim=8, use_linear_attn=True)


def test_weight_packing_layout(lib):
    """imagen_pack_igemm_weights: element (chunk, tap, group, cout, j) <- W[cout][chunk*KC + group*8 + j][tap] * in_scale."""
    from imagen_pytorch_amd import ops

    torch.manual_seed(0)
    for G, Cin, Cout, K in ((4, 64, 40, 3), (1, 24, 8, 1), (1, 8, 32, 15), (4, 96, 130, 2)):
        w = torch.randn(Cout, Cin, K, K)
        sc = torch.rand(Cin) + 0.5
        pw = ops.pack_weight(w, torch.randn(Cout), "cpu", in_scale=sc, G=G)
        KC, ntap = 8 * G, K * K
        KGP = (ntap * G + 1) // 2 * 2
        NC = pw.Cin_pad // KC
        body = NC * KGP * pw.Cout_pad * 8
        assert pw.w.numel() == body + 32 * pw.Cout_pad * 8 and pw.w[body:].abs().sum() == 0   # zero tail for the weight look-ahead
        packed = pw.w[:body].view(NC, KGP, pw.Cout_pad, 8).float()
        ref = torch.zeros_like(packed)
        ws = (w * sc.view(1, -1, 1, 1)).half().float().reshape(Cout, Cin, ntap)
        wp = torch.zeros(Cout, pw.Cin_pad, ntap)
        wp[:, :Cin] = ws
        for chunk in range(NC):
            for tap in range(ntap):
                for cg in range(G):
                    ref[chunk, tap * G + cg, :Cout, :] = wp[:, chunk * KC + cg * 8: chunk * KC + cg * 8 + 8, tap]
        assert torch.equal(packed, ref)
        assert pw.bias.shape[0] == pw.Cout_pad and pw.Cout_pad % 128 == 0


def test_tile_selection_respects_kernel_limits():
    from imagen_pytorch_amd import ops

    tab = ops.cfg_table()
    for G, Cout, OH, OW, B, K, stride in [(4, 32, 256, 256, 16, 3, 1), (4, 256, 32, 32, 16, 3, 1), (4, 256, 8, 8, 16, 3, 1),
                                          (1, 32, 256, 256, 16, 15, 1), (4, 64, 128, 128, 16, 2, 2), (4, 512, 1, 1024, 16, 1, 1),
                                          (4, 13000, 1, 16, 1, 1, 1), (1, 16, 24, 40, 1, 3, 1), (4, 3, 64, 64, 2, 3, 1)]:
        for fam in (None, 0):
            cfg, th, tw = ops.pick_cfg(G, Cout, OH, OW, B, K, K, stride, family=fam)
            tp, bn, g, family = tab[cfg]
            assert g == G and th * tw == tp and (fam is None or family == fam)
            it = ((th - 1) * stride + K) * ((tw - 1) * stride + K)
            assert it * G <= 256 * ops.load_library().imagen_igemm_stage_slots(cfg, K, K)
            lds = ops.load_library().imagen_igemm_lds_bytes(cfg, K, K, stride, th, tw)
            assert 0 < lds <= ops.MAX_LDS_BYTES


@pytest.mark.parametrize("name", ["unet_tiny_base.pt", "unet_tiny_sr.pt"])
def test_planner_dry_run_and_state_dict_surface(name):
    """The drop-in Unet loads the reference fixture's state_dict strictly; the planner builds the complete kernel plan
    (on CPU memory, never launched) for plain and CFG row layouts."""
    from imagen_pytorch_amd import Unet, _abi
    from imagen_pytorch_amd.engine import UnetEngine

    g = torch.load(os.path.join(GOLDEN, name), weights_only=False)
    u = Unet(**g["kwargs"]).eval()
    assert list(u.state_dict().keys()) == list(g["state_dict"].keys())
    u.load_state_dict(g["state_dict"])
    S = g["x"].shape[-1]
    for rows, src in ((2, 2), (4, 2)):
        eng = UnetEngine(u, rows, src, S, "cpu", dry=True)
        kinds = [k for k, _, _ in eng.step_plan.ops]
        assert kinds[0] == _abi.ENUMS["IMAGEN_OP_PACK_IMAGE"] and kinds[-1] == _abi.ENUMS["IMAGEN_OP_IGEMM"]
        assert kinds.count(_abi.ENUMS["IMAGEN_OP_ATTENTION"]) == len(eng.attn_sites) > 0
        keep = torch.ones(rows, dtype=torch.bool)
        keep[src:] = False
        lt = g["extra"].get("lowres_noise_times")
        eng.set_conditioning(text_embeds=g["text_embeds"], text_mask=g["text_mask"], keep=keep, lowres_noise_times=lt)
        static = eng._static_plans[g["text_embeds"].shape[1]][0]
        assert len(static) > 20
        # every attention site's K buffer has room for [context | null | self] rows
        for s in eng.attn_sites:
            assert s["Jp"] % 32 == 0
    # cast_model_parameters semantics (ip.py:1446-1470)
    same = u.cast_model_parameters(lowres_cond=u.lowres_cond, text_embed_dim=g["kwargs"]["text_embed_dim"], channels=3, channels_out=3,
                                   cond_on_text=True)
    assert same is u
    other = u.cast_model_parameters(lowres_cond=not u.lowres_cond, text_embed_dim=g["kwargs"]["text_embed_dim"], channels=3,
                                    channels_out=3, cond_on_text=True)
    assert other is not u and other.lowres_cond != u.lowres_cond


def test_imagen_constructor_surface():
    from imagen_pytorch_amd import Imagen, Unet

    k = dict(dim=8, cond_dim=32, dim_mults=(1, 2), attn_heads=2, max_text_len=16, attn_pool_num_latents=8)
    im = Imagen((Unet(**k), Unet(**k)), image_sizes=(16, 32), timesteps=(10, 5), cond_drop_prob=0.1)
    assert im.text_embed_dim == 768 and [u.lowres_cond for u in im.unets] == [False, True]
    assert [s.noise_schedule for s in im.noise_schedulers] == ["cosine", "cosine"]
    im3 = Imagen((Unet(**k), Unet(**k), Unet(**k)), image_sizes=(16, 32, 64), timesteps=3)
    assert [s.noise_schedule for s in im3.noise_schedulers] == ["cosine", "cosine", "linear"]   # ip.py:1853-1855
    tab = im.noise_schedulers[0].step_coefficients()
    assert tab.shape == (10, 8) and tab[-1, 5] == 0 and (tab[:-1, 5] == 1).all()
    with pytest.raises(NotImplementedError):
        im(torch.randn(1, 3, 16, 16))


def test_shard_bounds():
    from imagen_pytorch_amd.distributed import shard_bounds

    for total in (1, 7, 8, 32, 33):
        for world in (1, 2, 3, 8):
            spans = [shard_bounds(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


WORKER = r"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from imagen_pytorch_amd.distributed import sample_sharded
from oracle import sampler_oracle as so
g = torch.load(os.path.join(sys.argv[1], "tests", "golden", "sample_tiny_cascade.pt"), weights_only=False)
dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{sys.argv[2]}", rank=int(sys.argv[3]), world_size=int(sys.argv[4]))
unets = [(u["state_dict"], u["kwargs"]) for u in g["unets"]]
B = int(sys.argv[5])
torch.manual_seed(5)
te = torch.randn(B, 6, 32)
def noise_for(tag, shape, offset):
    # noise keyed by (tag, GLOBAL sample index): the property the Philox kernels provide on the GPU
    out = torch.empty(shape)
    for i in range(shape[0]):
        gen = torch.Generator().manual_seed(hash((tag, offset + i)) % (2**31))
        out[i] = torch.randn(shape[1:], generator=gen)
    return out
def sample_fn(text_embeds, text_masks, sample_offset, **kw):
    with torch.no_grad():
        return so.imagen_sample(unets, g["image_sizes"], text_embeds, timesteps=2, cond_scale=3.0, text_masks=text_masks,
                                noise_fn=lambda tag, shape: noise_for(tag, shape, sample_offset))
full = sample_sharded(sample_fn, te)
if dist.get_rank() == 0:
    torch.save(full, sys.argv[6])
dist.barrier()
dist.destroy_process_group()
"""


@pytest.mark.parametrize("B", [4, 3])
def test_batch_sharded_sampling_matches_single_process_gloo(tmp_path, B):
    """world_size-2 gloo run of the sharded sampling path (the oracle stands in for the GPU sampler): the all-gathered
    batch equals the single-process result when noise is keyed by the global sample index."""
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, PYTHONHASHSEED="0", OMP_NUM_THREADS="2")
    port = 29500 + (os.getpid() % 400) + B
    outs = []
    for world in (1, 2):
        out = tmp_path / f"out_{world}.pt"
        procs = [subprocess.Popen([sys.executable, str(script), ROOT, str(port + world), str(r), str(world), str(B), str(out)], env=env)
                 for r in range(world)]
        for p in procs:
            assert p.wait(timeout=600) == 0
        outs.append(torch.load(out))
    assert outs[0].shape == (B, 3, 32, 32)
    assert torch.allclose(outs[0], outs[1], atol=1e-6), (outs[0] - outs[1]).abs().max()


def test_elucidated_step_tables_match_oracle():
    """Host tables of the ElucidatedImagen sampler (imagen-pytorch_amd/elucidated.py::_tables) vs the oracle's per-step scalars
    (oracle/elucidated_oracle.py, el.py:323-336, 373-391, 428-436, 489-529)."""
    import math

    from imagen_pytorch_amd import ElucidatedImagen, Unet
    from oracle import elucidated_oracle as eo

    u = Unet(dim=8, cond_dim=32, text_embed_dim=32, dim_mults=(1, 2), num_resnet_blocks=1, layer_attns=(False, True),
             layer_cross_attns=(False, True), attn_heads=2, max_text_len=16, attn_pool_num_latents=8)
    hp = dict(eo.DEFAULT_HPARAMS, num_sample_steps=7)
    m = ElucidatedImagen((u,), image_sizes=(16,), text_embed_dim=32, **{k: v for k, v in hp.items()})
    init_sigma, (coef, w_hat, w_euler, w_heun, w_renoise) = m._tables(m.hparams[0])
    table, init_ref = eo.step_table(hp)
    assert init_sigma == init_ref and coef.shape == (14, 8)
    assert torch.equal(w_renoise[:, 0], torch.ones(14)) and not w_renoise[:, 1:].any()      # no resampling: the identity everywhere
    # inpainting with R = 3 resamples: every timestep's rows repeated (two per inner iteration, one at the last timestep), the
    # re-noising weight sigma - sigma_next on the second row of every iteration but the last resample (el.py:532-535)
    R, N = 3, 7
    _, (coef3, w_hat3, w_euler3, w_heun3, w_ren3) = m._tables(m.hparams[0], R)
    assert coef3.shape == (2 * (N - 1) * R + 2 * R, 8)        # R single rows for the last timestep + R spare (row count 2N at R = 1)
    for i, (sigma, sigma_next, gamma) in enumerate(table):
        for r in range(R):
            e = m._row(i, r, N, R)
            assert torch.equal(coef3[e], coef[2 * i]) and torch.equal(w_hat3[e], w_hat[2 * i]) and torch.equal(w_euler3[e], w_euler[2 * i])
            if i < N - 1:
                assert e == 2 * (i * R + (R - 1 - r))
                assert torch.equal(coef3[e + 1], coef[2 * i + 1]) and torch.equal(w_heun3[e + 1], w_heun[2 * i + 1])
                want = (sigma - sigma_next) if r > 0 else 0.0
                assert math.isclose(w_ren3[e + 1, 4].item(), want, rel_tol=1e-6) and w_ren3[e + 1, 0].item() == 1.0
            else:
                assert e == 2 * (N - 1) * R + (R - 1 - r)
    assert not w_ren3[2 * (N - 1) * R:, 1:].any()
    sd = hp["sigma_data"]
    for i, (sigma, sigma_next, gamma) in enumerate(table):
        sh = sigma + gamma * sigma
        c_skip, c_out = sd ** 2 / (sh ** 2 + sd ** 2), sh * sd * (sd ** 2 + sh ** 2) ** -0.5
        ref = torch.tensor([1 / c_skip, -c_out / c_skip, math.log(sh) * 0.25], dtype=torch.float32)
        assert torch.allclose(coef[2 * i, [0, 1, 6]], ref, rtol=1e-6)
        assert math.isclose(w_hat[2 * i, 4].item(), math.sqrt(sh ** 2 - sigma ** 2) * hp["S_noise"], rel_tol=1e-6, abs_tol=1e-12)
        assert math.isclose(w_hat[2 * i, 5].item(), (sh ** 2 + sd ** 2) ** -0.5, rel_tol=1e-6)
        # Euler / Heun weights reproduce the update formulas on scalars
        x_hat, x0, x0p = 0.7, -0.2, 0.4
        d = (x_hat - x0) / sh
        x_next = x_hat + (sigma_next - sh) * d
        assert math.isclose(w_euler[2 * i, 0].item() * x_hat + w_euler[2 * i, 1].item() * x0, x_next, rel_tol=1e-5, abs_tol=1e-6)
        if sigma_next != 0:
            dp = (x_next - x0p) / sigma_next
            ref_x = x_hat + 0.5 * (sigma_next - sh) * (d + dp)
            w = w_heun[2 * i + 1]
            got = w[0].item() * x_hat + w[1].item() * x0 + w[2].item() * x_next + w[3].item() * x0p
            assert math.isclose(got, ref_x, rel_tol=1e-4, abs_tol=1e-4)
        else:
            assert i == len(table) - 1 and w_euler[2 * i, 0].item() == 0.0 and w_euler[2 * i, 1].item() == 1.0


def test_text_encoder_hook_and_conditioning_handle():
    """sample(texts=...) goes through the `encode_text` hook (ip.py:2326-2332); prepare_conditioning() stages prompts once.
    Host logic only: no kernel runs here."""
    from imagen_pytorch_amd import Conditioning, Imagen, Unet

    kw = dict(dim=8, cond_dim=32, text_embed_dim=32, dim_mults=(1, 2), num_resnet_blocks=1, layer_attns=(False, True),
              layer_cross_attns=(False, True), attn_heads=2, max_text_len=16, attn_pool_num_latents=8)
    imagen = Imagen([Unet(**kw)], image_sizes=(16,), timesteps=2, text_embed_dim=32)
    calls = []

    def fake_encoder(texts, return_attn_mask=False):
        calls.append(list(texts))
        emb = torch.zeros(len(texts), 5, 32)
        for i, t in enumerate(texts):
            emb[i, : len(t.split())] = 1.0 + i
        mask = emb.abs().sum(-1) > 0
        return (emb, mask) if return_attn_mask else emb

    imagen.encode_text = fake_encoder
    emb, mask = imagen._resolve_text(["a b c", "d"], None, None, torch.device("cpu"))
    assert calls == [["a b c", "d"]] and emb.shape == (2, 5, 32) and mask.tolist() == [[True] * 3 + [False] * 2, [True] + [False] * 4]
    with pytest.raises(AssertionError, match="text cannot be empty"):
        imagen._resolve_text(["ok", ""], None, None, torch.device("cpu"))
    # precomputed embeddings bypass the hook; the default mask is "any non-zero feature" (ip.py:2337)
    emb2, mask2 = imagen._resolve_text(None, emb, None, torch.device("cpu"))
    assert len(calls) == 1 and torch.equal(mask2, mask)
    c = imagen.prepare_conditioning(["x y", "z"], device="cpu")
    assert isinstance(c, Conditioning) and c.batch_size == 2 and c.text_embeds.shape == (2, 5, 32) and len(calls) == 2
    assert imagen.prepare_conditioning(text_embeds=emb, device="cpu").token is not c.token
    with pytest.raises(AssertionError, match="invalid text embedding dimension"):
        imagen.prepare_conditioning(text_embeds=torch.zeros(1, 3, 7), device="cpu")
    # the default hook needs T5 weights on local disk: absent here -> a loud, actionable error (never the network)
    fresh = Imagen([Unet(**kw)], image_sizes=(16,), timesteps=2, text_embed_dim=32, text_encoder_name="google/t5-v1_1-small")
    with pytest.raises(RuntimeError, match="not available from local files"):
        fresh.encode_text(["hello"], return_attn_mask=True)


def test_round4_planner_predicates():
    """Host-side decisions added in round 4 (no GPU): the logit bound that selects the bounded-logit attention tiling, the width above which the
    GlobalContext finalisation runs as two many-workgroup launches, the step count STEP_SLICE clamps to, and the family-7 routing rule."""
    import math

    from imagen_pytorch_amd import ops

    # |q^ . k^| * q_mult <= q_mult * max |q_scale * k_scale| (+ 1 % and 0.05 for the fp16 rounding of the unit rows)
    ones = torch.ones(64)
    b = ops.attention_logit_bound(ones, ones, 8 * ops.LOG2E)
    assert math.isclose(b, 8 * ops.LOG2E * 1.01 + 0.05) and b <= ops.ATTN_BOUND_MAX           # the reference's init: bounded tiling
    assert ops.attention_logit_bound(1.5 * ones, ones, 8 * ops.LOG2E) > ops.ATTN_BOUND_MAX    # trained scales can leave it: online softmax
    qs = torch.linspace(0.5, 1.2, 64)
    assert math.isclose(ops.attention_logit_bound(qs, -qs, 2.0), 2.0 * 1.44 * 1.01 + 0.05, rel_tol=1e-6)   # the sign does not matter, the largest product does
    # wide GlobalContext blocks (C2's 512- / 1024-channel levels) finalise in two phases; the README widths and anything the kernels cannot hold do not
    assert ops.gca_final_is_wide(512, 256) and ops.gca_final_is_wide(1024, 512)
    assert not ops.gca_final_is_wide(256, 128) and not ops.gca_final_is_wide(128, 64) and not ops.gca_final_is_wide(2048, 1024)
    # family 7 exists in the library and is built for 128-pixel x 128-cout tiles of 32-channel chunks
    gid = ops.gemm_cfg()
    assert gid is not None and ops.cfg_table()[gid] == (128, 128, 4, 7)


def test_family7_respects_the_launcher_k_limit():
    """a 1x1 layer whose (split-precision-doubled) input width exceeds conv_gemm.hip's Cin_pad <= 2048 must not be routed to
    family 7 — before the fix a Cin = 4096 layer and a split Cin = 2048 layer got cfg 49 and failed at launch time with no fallback.  Dry plans on
    CPU memory; `imagen_launch` is only asked whether it accepts the parameters (the CPU library refuses to run, which is a different error)."""
    from imagen_pytorch_amd import ops

    gid = ops.gemm_cfg()
    rows, cout = 8192, 1024
    for cin, split in ((4096, False), (2048, True), (2048, False), (1024, True)):
        w = ops.pack_weight(torch.randn(cout, cin) * 0.01, None, "cpu", split=split)
        x = ops.new_act(1, 1, rows, cin, "cpu")
        y = ops.new_act(1, 1, rows, cout, "cpu")
        plan = ops.Plan("k-limit")
        p = ops.igemm(plan, x, w, y, label="wide")
        fam = ops.cfg_table()[p.cfg][3]
        if w.Cin_pad > ops.GEMM_MAX_K:
            assert fam != 7, f"Cin_pad {w.Cin_pad} routed to family 7"
        else:
            assert p.cfg == gid, f"Cin_pad {w.Cin_pad} should still take family 7 (got cfg {p.cfg})"
    # the doubled affine of a split launch belongs to the plan (ADVICE: no process-global cache)
    pa = torch.ones(64)
    w = ops.pack_weight(torch.randn(32, 64) * 0.1, None, "cpu", split=True)
    plan = ops.Plan("twice")
    ops.igemm(plan, ops.new_act(1, 1, 64, 64, "cpu"), w, ops.new_act(1, 1, 64, 32, "cpu"), pa=pa)
    assert len(plan.twice) == 1 and not hasattr(ops, "_TWICE")


def test_family8_routing_rules():
    """The small-map 3x3 family (conv_small.hip) takes a launch only where it measured faster (round 5, calls H - L): at most 4096 output pixels, at
    most 64 MB of weight data streamed by the launch's pixel tiles together; the 32-cout tile unless the epilogue needs every channel of a pixel
    (then the narrowest tile over all couts, Cout <= 128).  Dry plans on CPU memory."""
    from imagen_pytorch_amd import ops

    tab = ops.cfg_table()

    def pick(B, H, C1, Cout, C2=0, **kw):
        w = ops.pack_weight(torch.randn(Cout, C1 + C2, 3, 3) * 0.01, None, "cpu", G=4)
        x1, x2 = ops.new_act(B, H, H, C1, "cpu"), (ops.new_act(B, H, H, C2, "cpu") if C2 else None)
        p = ops.igemm(ops.Plan("fam8"), x1, w, ops.new_act(B, H, H, Cout, "cpu"), x2=x2, **kw)
        return tab[p.cfg], (p.TH, p.TW), p

    ssq = lambda B, H: dict(ssq_a=torch.ones(B * H * H), pa=torch.ones(1, 1), act_in=ops.ACT_SILU)
    assert pick(16, 8, 256, 256, C2=128)[:2] == ((32, 32, 4, 8), (4, 8))                       # README unet1's widest small-map layer: 54 MB of weight stream
    assert pick(16, 16, 128, 128)[:2] == ((32, 32, 4, 8), (2, 16))
    t, sh, p = pick(16, 16, 128, 128, gca=dict(wk=torch.ones(128), bk=0.0))                   # GlobalContext partials: one tile over all 128 couts
    assert t == (32, 128, 4, 8) and p.gca_chunks == 8 and p.gca_part_t is not None
    t, sh, p = pick(16, 8, 64, 64, ssq_out=torch.empty(16 * 64))
    assert t == (32, 64, 4, 8) and p.ssq_emitted
    t, sh, p = pick(16, 8, 256, 256, ssq_out=torch.empty(16 * 64))                             # wider than any tile: the 32-cout tile, statistics left to the caller
    assert t == (32, 32, 4, 8) and not p.ssq_emitted
    assert pick(16, 32, 128, 128)[0][3] != 8                                                   # 16384 pixels: the 32^2 maps stay where they were
    assert pick(16, 16, 512, 512)[0][3] != 8 and pick(16, 8, 1024, 1024)[0][3] != 8            # C2's layers: 604 MB of weight stream
    assert pick(4, 16, 512, 512)[0][3] != 8 and pick(2, 16, 512, 512)[0][3] != 8               # ... 151 MB at a quarter of the rows, 75 MB at an eighth: still not
    assert pick(1, 16, 512, 512)[0][3] == 8                                                    # ... 38 MB: taken (the limit is 64 MB since round 6, call D: C2's 100 - 134 MB 1x1 GEMMs lost 2x on it)


def test_time_chain_runs_in_fp32(monkeypatch):
    """The timestep-conditioning chain of the image planner (to_time_cond, the batched time MLPs) is two LINEAR_F32 launches with fp32 rows between
    them and SCALE_SHIFT reading fp32 rows — per step AND in the per-request time table — and the table estimate of BASELINE C2 stays under the
    default cap with the fp32 rows counted (engine.TIME_CHAIN_F32; 0 restores the fp16 GEMMs of rounds 1-5)."""
    from imagen_pytorch_amd import Unet, _abi, engine
    from imagen_pytorch_amd.engine import UnetEngine

    K = _abi.ENUMS
    kw = dict(dim=32, cond_dim=64, dim_mults=(1, 2), num_resnet_blocks=1, layer_attns=(False, True), layer_cross_attns=(False, True), attn_heads=2)
    u = Unet(**kw).eval()

    def chain(plan, prefix=""):
        by = {label: (kind, p) for kind, p, label in plan.ops}
        return by[prefix + "to_time_cond"], by[prefix + "time_mlps"], by[prefix + "scale_shift"]

    eng = UnetEngine(u, 4, 2, 16, "cpu", dry=True)
    (k1, p1), (k2, p2), (k3, p3) = chain(eng.step_plan)
    assert k1 == k2 == K["IMAGEN_OP_LINEAR_F32"] and k3 == K["IMAGEN_OP_SCALE_SHIFT"]
    assert p1.x_f32 == 0 and p1.res and p1.act_in == 0 and p2.x_f32 == 1 and p2.x == p1.y and p2.act_in == K["IMAGEN_ACT_SILU"] and p3.ss_f32 == 1 and p3.ss == p2.y
    eng.set_conditioning(text_embeds=torch.randn(2, 8, 768), text_mask=None, keep=torch.ones(4, dtype=torch.bool), lowres_noise_times=None)
    coef, step = torch.zeros(7, 8), torch.zeros(1, dtype=torch.int32)
    assert eng.enable_time_table(coef, step) is not None
    (k1, p1), (k2, p2), (k3, p3) = chain(eng._tt_plan, "tt.")
    assert k1 == k2 == K["IMAGEN_OP_LINEAR_F32"] and p1.rows == p2.rows == 7 * 4 and p3.ss_f32 == 1 and p3.B == 7 * 4
    assert not any(label in ("to_time_cond", "time_mlps", "scale_shift") for _, _, label in eng.step_plan_tt.ops)

    monkeypatch.setattr(engine, "TIME_CHAIN_F32", 0)
    old = UnetEngine(Unet(**kw).eval(), 4, 2, 16, "cpu", dry=True)
    (k1, _), (k2, _), (_, p3) = chain(old.step_plan)
    assert k1 == k2 == K["IMAGEN_OP_IGEMM"] and p3.ss_f32 == 0


def test_a_removed_switch_in_the_environment_is_an_error():
    """the A/B switches of rounds 2-4 (IMAGEN_CONV_DMA, IMAGEN_BIG_PREP, ...) are module constants now; a script that still
    exports one must not silently run the default configuration twice."""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, IMAGEN_CONV_DMA="0")
    r = subprocess.run([sys.executable, "-c", "import imagen_pytorch_amd"], cwd=root, env=env, capture_output=True, text=True)
    assert r.returncode != 0 and "IMAGEN_CONV_DMA" in r.stderr and "module constants" in r.stderr, r.stderr[-400:]
    env = dict(os.environ, IMAGEN_ROWCHAIN="1")
    r = subprocess.run([sys.executable, "-c", "import imagen_pytorch_amd"], cwd=root, env=env, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-400:]


def test_merged_request_seed_spans_and_guards():
    """Imagen.sample_requests, host side (the sampling itself is a -m gpu test): one Philox span per request with sample indices restarting at 0, a
    single seed is one span at the shard's offset; options the merged path does not cover raise before anything is launched, and so does the
    ElucidatedImagen sampler (its noise launches take one key per batch)."""
    from imagen_pytorch_amd import ElucidatedImagen, Imagen, Unet
    from imagen_pytorch_amd.imagen import _seed_spans

    assert _seed_spans(7, 8, sample_offset=16) == [(7, 0, 8, 16)]
    assert _seed_spans([(41, 2), ((5 << 31) | 42, 3), (43, 1)], 6) == [(41, 0, 2, 0), ((5 << 31) | 42, 2, 3, 0), (43, 5, 1, 0)]
    with pytest.raises(AssertionError):
        _seed_spans([(1, 2), (2, 2)], 5)
    kw = dict(dim=8, cond_dim=16, dim_mults=(1, 2), num_resnet_blocks=1, layer_attns=(False, True), layer_cross_attns=(False, True), attn_heads=2)
    imagen = Imagen([Unet(**kw)], image_sizes=(16,), timesteps=4, text_embed_dim=32)
    reqs = [dict(text_embeds=torch.randn(2, 5, 32), seed=1), dict(text_embeds=torch.randn(1, 7, 32), seed=2)]
    for bad in (dict(init_images=torch.zeros(3, 3, 16, 16)), dict(inpaint_images=torch.zeros(3, 3, 16, 16)), dict(noise_fn=lambda *a: None)):
        with pytest.raises(NotImplementedError):
            imagen.sample_requests(reqs, **bad)
    with pytest.raises(AssertionError):
        imagen.sample_requests([dict(reqs[0], cond_scale=2.0), reqs[1]])
    assert imagen.sample_requests([]) == []
    edm = ElucidatedImagen([Unet(**kw)], image_sizes=(16,), num_sample_steps=4, text_embed_dim=32)
    with pytest.raises(NotImplementedError):
        edm.sample_requests(reqs)


def test_first_small_unet_of_a_process_can_be_recast():
    """Regression (round 6): the first dim < 128 Unet of a process printed the reference's hint through a function-local `import sys as _sys`, which
    `dict(locals())` then kept among the constructor kwargs — Imagen's cast_model_parameters re-constructed it with an unknown keyword."""
    import subprocess
    code = ("import torch\nfrom imagen_pytorch_amd import Imagen, Unet\n"
            "u = Unet(dim=8, cond_dim=16, dim_mults=(1, 2), num_resnet_blocks=1, layer_attns=(False, True), layer_cross_attns=(False, True), attn_heads=2)\n"
            "assert '_sys' not in u._locals\nImagen([u], image_sizes=(16,), timesteps=4, text_embed_dim=32)\nprint('ok')\n")
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
    assert r.returncode == 0 and r.stdout.decode().strip().endswith("ok"), r.stdout.decode()[-1500:]
