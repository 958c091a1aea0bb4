"""CPU: the oracle restatement (oracle/) against the golden fixtures generated from the live reference by
oracle/make_golden.py.  These fixtures travel to the GPU box, so the oracle is pinned there too."""
import os

import pytest
import torch

from oracle import sampler_oracle as so
from oracle import unet3d_oracle as u3
from oracle import unet_oracle as uo

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def _load(name):
    return torch.load(os.path.join(GOLDEN, name), weights_only=False)


@pytest.mark.parametrize("name", ["unet_tiny_base.pt", "unet_tiny_sr.pt"])
def test_unet_oracle_matches_reference_fixture(name):
    g = _load(name)
    kw = dict(text_embeds=g["text_embeds"], text_mask=g["text_mask"], **g["extra"])
    with torch.no_grad():
        cond = uo.unet_forward(g["state_dict"], g["kwargs"], g["x"], g["time"], **kw)
        null = uo.unet_forward(g["state_dict"], g["kwargs"], g["x"], g["time"], cond_drop_prob=1.0, **kw)
        cfg = uo.unet_forward_with_cond_scale(g["state_dict"], g["kwargs"], g["x"], g["time"], cond_scale=3.0, **kw)
    for got, ref in ((cond, g["out_cond"]), (null, g["out_null"]), (cfg, g["out_cfg"])):
        assert ref.abs().mean() > 0.05, "vacuous fixture (zero-initialised final_conv?)"
        assert torch.allclose(got, ref, atol=2e-5, rtol=1e-5), (got - ref).abs().max()


def test_sampler_oracle_matches_reference_fixture():
    g = _load("sample_tiny_cascade.pt")
    unets = [(u["state_dict"], u["kwargs"]) for u in g["unets"]]
    noise_fn = lambda tag, shape: g["noise"][tag]
    with torch.no_grad():
        outs = so.imagen_sample(unets, g["image_sizes"], g["text_embeds"], timesteps=g["timesteps"], cond_scale=g["cond_scale"],
                                noise_fn=noise_fn, return_all=True)
    for got, ref in zip(outs, g["outputs"]):
        assert got.shape == ref.shape
        assert torch.allclose(got, ref, atol=2e-4), (got - ref).abs().max()


@pytest.mark.parametrize("run", ["init_skip", "inpaint"])
def test_sampler_oracle_options_match_reference_fixture(run):
    """init_images + skip_steps, and the inpainting resample loop (ip.py:2167-2289), vs recorded runs of the live reference."""
    g = _load("sample_tiny_options.pt")
    r = g["runs"][run]
    unets = [(u["state_dict"], u["kwargs"]) for u in g["unets"]]
    kw = {k: r[k] for k in ("init_images", "skip_steps", "inpaint_images", "inpaint_masks", "inpaint_resample_times") if k in r}
    with torch.no_grad():
        outs = so.imagen_sample(unets, g["image_sizes"], g["text_embeds"], timesteps=g["timesteps"], cond_scale=g["cond_scale"],
                                noise_fn=lambda tag, shape: r["noise"][tag], return_all=True, **kw)
    for got, ref in zip(outs, r["outputs"]):
        assert got.shape == ref.shape
        assert torch.allclose(got, ref, atol=2e-4), (got - ref).abs().max()


def test_inpaint_tables_match_oracle():
    """schedules.inpaint_coefficients vs the oracle's q_sample / q_sample_from_to on a probe image."""
    from imagen_pytorch_amd.schedules import GaussianDiffusionContinuousTimes

    T, R = 5, 3
    sch = GaussianDiffusionContinuousTimes(noise_schedule="cosine", timesteps=T)
    step, blend, renoise = sch.inpaint_coefficients(R, philox=True)
    assert step.shape == blend.shape == renoise.shape == (T * R, 8)
    assert torch.equal(step, sch.step_coefficients().repeat_interleave(R, dim=0))
    x, z = torch.randn(1, 3, 4, 4), torch.randn(1, 3, 4, 4)
    row = 0
    for i, (t, tn) in enumerate(so.sampling_time_pairs(T)):
        for r in reversed(range(R)):
            a, s = so.alpha_sigma(so.SCHEDULES["cosine"](t))
            assert torch.allclose(blend[row, 0] * x + blend[row, 4] * z, a * x + s * z, atol=1e-6)
            if r == 0 or i == T - 1:
                assert renoise[row].tolist() == [1.0, 0, 0, 0, 0, 0, 0, 0]
            else:
                ref = so.q_sample_from_to(x, tn.reshape(1), t.reshape(1), z, "cosine")
                assert torch.allclose(renoise[row, 0] * x + renoise[row, 4] * z, ref, atol=1e-6)
            row += 1
    _, b1, q1 = sch.inpaint_coefficients(R, philox=False)     # injected noise: the weight moves to column 1
    assert torch.equal(b1[:, 1], blend[:, 4]) and torch.equal(q1[:, 1], renoise[:, 4]) and not b1[:, 4].any() and not q1[:, 4].any()


def test_schedule_tables_match_oracle():
    """Host coefficient table (imagen-pytorch_amd/schedules.py) vs the oracle's per-step formulas."""
    from imagen_pytorch_amd.schedules import GaussianDiffusionContinuousTimes

    for sched in ("cosine", "linear"):
        T = 17
        tab = GaussianDiffusionContinuousTimes(noise_schedule=sched, timesteps=T).step_coefficients()
        pairs = so.sampling_time_pairs(T)
        for i, (t, tn) in enumerate(pairs):
            l, ln = so.SCHEDULES[sched](t), so.SCHEDULES[sched](tn)
            a, s = so.alpha_sigma(l)
            an, sn = so.alpha_sigma(ln)
            ref = torch.stack([a, s, an, sn, -torch.special.expm1(l - ln), torch.tensor(0.0 if tn == 0 else 1.0), l])
            assert torch.equal(tab[i, :7], ref)


def test_elucidated_oracle_matches_reference_fixture():
    """oracle/elucidated_oracle.py vs the recorded ElucidatedImagen.sample run of the live reference (5 Karras steps, Heun
    correction, dynamic thresholding, CFG 3, 2-stage cascade with low-res augmentation), identical Gaussian draws.
    Stage 1 in isolation to 2e-4; stage 2 in isolation (started from the reference's own stage-1 image) to 2e-4; the chained
    cascade to 2e-3 (the 5-step sampler from sigma = 80 amplifies stage 1's 4e-5 by ~20x)."""
    import torch.nn.functional as F

    from oracle import elucidated_oracle as eo
    from oracle.unet_oracle import unet_forward_with_cond_scale

    g = _load("sample_tiny_elucidated.pt")
    unets = [(u["state_dict"], u["kwargs"]) for u in g["unets"]]
    noise_fn = lambda tag, shape: g["noise"][tag]
    with torch.no_grad():
        outs = eo.elucidated_sample(unets, g["image_sizes"], g["text_embeds"], hparams=g["hparams"], cond_scale=g["cond_scale"],
                                    noise_fn=noise_fn, return_all=True)
    assert torch.allclose(outs[0], g["outputs"][0], atol=2e-4), (outs[0] - g["outputs"][0]).abs().max()
    assert torch.allclose(outs[1], g["outputs"][1], atol=2e-3), (outs[1] - g["outputs"][1]).abs().max()
    # stage 2 alone
    hp = dict(eo.DEFAULT_HPARAMS, **g["hparams"])
    sd, kw = unets[1]
    te = g["text_embeds"]
    tm = torch.any(te != 0.0, dim=-1)
    lt = torch.full((te.shape[0],), 0.2)
    up = F.interpolate(g["outputs"][0], g["image_sizes"][1], mode="nearest") * 2 - 1
    a, sg = so.alpha_sigma(so.SCHEDULES["linear"](lt).reshape(-1, 1, 1, 1))
    li = a * up + sg * g["noise"][("lowres", 1)]
    net = lambda x, c: unet_forward_with_cond_scale(sd, kw, x, c, cond_scale=g["cond_scale"], text_embeds=te, text_mask=tm,
                                                    lowres_cond_img=li, lowres_noise_times=lt)
    with torch.no_grad():
        out = eo.one_unet_sample(net, tuple(g["outputs"][1].shape), hp, noise_fn=noise_fn, stage=1)
    assert torch.allclose(out, g["outputs"][1], atol=2e-4), (out - g["outputs"][1]).abs().max()


@pytest.mark.parametrize("tag", ["base", "sr"])
def test_unet3d_oracle_matches_reference_fixture(tag):
    """SURVEY §8(f) NEXT-2 groundwork: oracle/unet3d_oracle.py vs Unet3D.forward of the live reference (tiny clip, temporal convs /
    temporal attention / time token shift / temporal down+up-sampling de-initialised so they all contribute)."""
    g = _load("unet3d_tiny.pt")["runs"][tag]
    kw = dict(text_embeds=g["text_embeds"], text_mask=g["text_mask"], **g["extra"])
    with torch.no_grad():
        cond = u3.unet3d_forward(g["state_dict"], g["kwargs"], g["x"], g["time"], **kw)
        null = u3.unet3d_forward(g["state_dict"], g["kwargs"], g["x"], g["time"], cond_drop_prob=1.0, **kw)
        cfg = u3.unet3d_forward_with_cond_scale(g["state_dict"], g["kwargs"], g["x"], g["time"], cond_scale=3.0, **kw)
        notime = u3.unet3d_forward(g["state_dict"], g["kwargs"], g["x"], g["time"], ignore_time=True, **kw)
    assert (g["out_cond"] - g["out_notime"]).abs().mean() > 0.5, "vacuous fixture: the temporal layers are still at their identity init"
    for got, ref in ((cond, g["out_cond"]), (null, g["out_null"]), (cfg, g["out_cfg"]), (notime, g["out_notime"])):
        assert got.shape == ref.shape and ref.abs().mean() > 0.05
        assert torch.allclose(got, ref, atol=5e-5, rtol=1e-5), (got - ref).abs().max()


def test_video_sampler_oracle_matches_reference_fixture():
    """Imagen.sample over two Unet3D stages (`video_frames` = 4): sampler oracle + Unet3D oracle vs the recorded reference run."""
    g = _load("sample_tiny_video.pt")
    unets = [(u["state_dict"], u["kwargs"]) for u in g["unets"]]
    with torch.no_grad():
        outs = so.imagen_sample(unets, g["image_sizes"], g["text_embeds"], timesteps=g["timesteps"], cond_scale=g["cond_scale"],
                                noise_fn=lambda tag, shape: g["noise"][tag], return_all=True, video_frames=g["frames"])
    for got, ref in zip(outs, g["outputs"]):
        assert got.shape == ref.shape and got.ndim == 5
        assert torch.allclose(got, ref, atol=2e-4), (got - ref).abs().max()
    # first stage at half the frame rate (temporal_downsample_factor = (2, 1)): nearest resize over the frame axis in between
    t = g["tds"]
    with torch.no_grad():
        outs = so.imagen_sample(unets, g["image_sizes"], g["text_embeds"], timesteps=g["timesteps"], cond_scale=g["cond_scale"],
                                noise_fn=lambda tag, shape: t["noise"][tag], return_all=True, video_frames=g["frames"],
                                temporal_downsample_factor=t["temporal_downsample_factor"])
    assert outs[0].shape[2] == g["frames"] // 2 and outs[1].shape[2] == g["frames"]
    for got, ref in zip(outs, t["outputs"]):
        assert torch.allclose(got, ref, atol=2e-4), (got - ref).abs().max()


def test_video_elucidated_oracle_matches_reference_fixture():
    """ElucidatedImagen.sample over the two Unet3D stages of the tiny video cascade (3 Karras steps, Heun) vs the recorded reference run."""
    from oracle import elucidated_oracle as eo

    g = _load("sample_tiny_video.pt")
    e = g["edm"]
    unets = [(u["state_dict"], u["kwargs"]) for u in g["unets"]]
    with torch.no_grad():
        outs = eo.elucidated_sample(unets, g["image_sizes"], g["text_embeds"], hparams=e["hparams"], cond_scale=g["cond_scale"],
                                    noise_fn=lambda tag, shape: e["noise"][tag], return_all=True, video_frames=g["frames"])
    # fp32 round-off through the sigma_max = 80 loop: max-abs 5e-4 (mean 2e-5) on stage 1, 2.7e-3 (mean 7e-6) on the chained stage 2
    assert torch.allclose(outs[0], e["outputs"][0], atol=1e-3), (outs[0] - e["outputs"][0]).abs().max()
    assert torch.allclose(outs[1], e["outputs"][1], atol=5e-3), (outs[1] - e["outputs"][1]).abs().max()
    assert (outs[1] - e["outputs"][1]).abs().mean() < 5e-5


@pytest.mark.parametrize("tag", ["init_skip", "inpaint", "sigma"])
def test_elucidated_oracle_options_match_reference_fixture(tag):
    """one_unet_sample options of the Karras et al. sampler (init_images + skip_steps, inpainting with resampling, per-call sigma
    overrides; el.py:393-545) vs recorded runs of the live reference on the weights of sample_tiny_elucidated.pt."""
    from oracle import elucidated_oracle as eo

    o = _load("sample_tiny_elucidated_options.pt")
    g = _load(o["weights_from"])
    run = o["runs"][tag]
    unets = [(u["state_dict"], u["kwargs"]) for u in g["unets"]]
    with torch.no_grad():
        outs = eo.elucidated_sample(unets, g["image_sizes"], g["text_embeds"], hparams=g["hparams"], cond_scale=g["cond_scale"],
                                    noise_fn=lambda t, shape: run["noise"][t], return_all=True, **run["kwargs"])
    # tolerances of the plain EDM test above (fp32 round-off through the sigma_max = 80 loop)
    assert torch.allclose(outs[0], run["outputs"][0], atol=1e-3), (outs[0] - run["outputs"][0]).abs().max()
    assert torch.allclose(outs[1], run["outputs"][1], atol=5e-3), (outs[1] - run["outputs"][1]).abs().max()
    assert (outs[1] - run["outputs"][1]).abs().mean() < 1e-4


def test_video_elucidated_oracle_prompt_frames_match_reference_fixture():
    """ElucidatedImagen.sample over the Unet3D stages with cond_video_frames vs the recorded reference run (tolerances of the plain
    EDM video test above: fp32 round-off through the sigma_max = 80 loop)."""
    from oracle import elucidated_oracle as eo

    o = _load("sample_tiny_video_options.pt")
    g = _load(o["weights_from"])
    run = o["runs"]["edm_cond_pre"]
    unets = [(u["state_dict"], u["kwargs"]) for u in g["unets"]]
    with torch.no_grad():
        outs = eo.elucidated_sample(unets, g["image_sizes"], g["text_embeds"], hparams=run["hparams"], cond_scale=g["cond_scale"],
                                    noise_fn=lambda tag, shape: run["noise"][tag], return_all=True, video_frames=o["frames"], **run["kwargs"])
    assert torch.allclose(outs[0], run["outputs"][0], atol=1e-3), (outs[0] - run["outputs"][0]).abs().max()
    assert torch.allclose(outs[1], run["outputs"][1], atol=5e-3), (outs[1] - run["outputs"][1]).abs().max()
    assert (outs[1] - run["outputs"][1]).abs().mean() < 5e-5


VIDEO_OPTION_RUNS = ["cond_pre", "cond_post", "cond_both", "init_skip", "inpaint", "cond_pre_tds"]


def video_option_kwargs(run):
    """Oracle / product keyword arguments of one recorded run of sample_tiny_video_options.pt."""
    kw = dict(run["kwargs"])
    if "inpaint_videos" in kw:
        kw["inpaint_images"] = kw.pop("inpaint_videos")
    return kw


@pytest.mark.parametrize("tag", VIDEO_OPTION_RUNS)
def test_video_sampler_oracle_options_match_reference_fixture(tag):
    """Video-stage options of Imagen.sample — prompt frames before / after the clip (with the reference's own frame order, iv.py:1703,
    1716), init videos + skip_steps, video inpainting with resampling, prompt frames under a per-stage frame rate — vs recorded runs
    of the live reference on the weights of sample_tiny_video.pt."""
    o = _load("sample_tiny_video_options.pt")
    g = _load(o["weights_from"])
    run = o["runs"][tag]
    unets = [(u["state_dict"], u["kwargs"]) for u in g["unets"]]
    with torch.no_grad():
        outs = so.imagen_sample(unets, g["image_sizes"], g["text_embeds"], timesteps=o["timesteps"], cond_scale=g["cond_scale"],
                                noise_fn=lambda t, shape: run["noise"][t], return_all=True, video_frames=o["frames"],
                                temporal_downsample_factor=run.get("temporal_downsample_factor", 1), **video_option_kwargs(run))
    for got, ref in zip(outs, run["outputs"]):
        assert got.shape == ref.shape
        assert torch.allclose(got, ref, atol=2e-4), (got - ref).abs().max()
