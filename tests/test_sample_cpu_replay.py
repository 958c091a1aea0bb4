"""CPU: the REAL sampling driver — Imagen.sample / ElucidatedImagen.sample, p_sample_loop, stage caching, graph objects, low-res
conditioning, layout conversion — executed end to end with the kernel launcher replaced by the plan interpreter
(tests/plan_interp.py), against the recorded runs of the live reference.

Only test-side patches: `ops.Plan.run` and `ops.Graph` go to the interpreter, the few `torch.cuda` calls of the driver become
no-ops, and imagen._SAMPLING_DEVICE_TYPES admits 'cpu'.  Nothing of this exists in the product, which has no CPU path; what it buys
is that every line of the host-side sampling code is exercised without a GPU (the HIP kernels are covered by the -m gpu tests)."""
import contextlib
import os

import pytest
import torch

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def nerr(a, b):
    return float((a.float() - b.float()).norm() / b.float().norm())


class _Stream:
    device = torch.device("cpu")
    cuda_stream = 0

    def __init__(self, *a, **k):
        pass

    def synchronize(self):
        pass

    def wait_stream(self, other):
        pass

    def wait_event(self, ev):
        pass


class _Event:
    def __init__(self, *a, **k):
        pass

    def record(self, stream=None):
        pass


@pytest.fixture()
def cpu_backend(monkeypatch):
    from imagen_pytorch_amd import imagen as imagen_mod
    from imagen_pytorch_amd import ops
    from imagen_pytorch_amd import unet as unet_mod
    from plan_interp import Interpreter

    it = Interpreter()
    ops.KEEP_REFERENCE_WEIGHTS = True
    monkeypatch.setattr(ops.Plan, "run", lambda self, stream=None: it.run(self))

    class Graph:
        def __init__(self, plan, stream):
            self.plan = plan

        def launch(self):
            self.plan.run()

    monkeypatch.setattr(ops, "Graph", Graph)
    monkeypatch.setattr(imagen_mod, "_SAMPLING_DEVICE_TYPES", ("cuda", "cpu"))
    monkeypatch.setattr(unet_mod, "_ENGINE_DEVICE_TYPES", ("cuda", "cpu"))
    monkeypatch.setattr(torch.cuda, "Stream", _Stream)
    monkeypatch.setattr(torch.cuda, "Event", _Event)
    monkeypatch.setattr(torch.cuda, "current_stream", lambda *a, **k: _Stream())
    monkeypatch.setattr(torch.cuda, "synchronize", lambda *a, **k: None)
    monkeypatch.setattr(torch.cuda, "device", lambda *a, **k: contextlib.nullcontext())
    monkeypatch.setattr(torch.cuda, "stream", lambda *a, **k: contextlib.nullcontext())
    try:
        yield it
    finally:
        ops.KEEP_REFERENCE_WEIGHTS = False
        ops.REFERENCE_WEIGHTS.clear()


def _cascade(g, klass=None, **kw):
    from imagen_pytorch_amd import Imagen, Unet, Unet3D

    video = "frames" in g
    make = Unet3D if video else Unet
    unets = [make(**spec["kwargs"]).eval() for spec in g["unets"]]
    imagen = (klass or Imagen)(unets, image_sizes=g["image_sizes"], text_embed_dim=32, cond_drop_prob=0.1, **kw)
    for u, spec in zip(imagen.unets, g["unets"]):
        u.load_state_dict(spec["state_dict"])
    return imagen.eval()


@pytest.mark.parametrize("use_graph", [True, False])
def test_cascade_sample_driver(cpu_backend, use_graph):
    g = torch.load(os.path.join(GOLDEN, "sample_tiny_cascade.pt"), weights_only=False)
    imagen = _cascade(g, timesteps=g["timesteps"])
    outs = imagen.sample(text_embeds=g["text_embeds"], cond_scale=g["cond_scale"], use_tqdm=False, return_all_unet_outputs=True,
                         noise_fn=lambda tag, shape: g["noise"][tag], use_graph=use_graph, device="cpu")
    errs = [nerr(o, r) for o, r in zip(outs, g["outputs"])]
    assert max(errs) < 2e-2, errs
    # PIL output (ip.py:2488-2498): torchvision's ToPILImage semantics — mul(255).byte(), HWC, RGB
    pil = imagen.sample(text_embeds=g["text_embeds"], cond_scale=g["cond_scale"], use_tqdm=False, noise_fn=lambda tag, shape: g["noise"][tag],
                        use_graph=use_graph, device="cpu", return_pil_images=True)
    assert len(pil) == outs[-1].shape[0] and pil[0].mode == "RGB" and pil[0].size == (outs[-1].shape[-1], outs[-1].shape[-2])
    import numpy as np
    assert np.array_equal(np.asarray(pil[1]), outs[-1][1].mul(255).byte().permute(1, 2, 0).numpy())
    # a second call reuses the cached stages / graphs and gives the same images
    again = imagen.sample(text_embeds=g["text_embeds"], cond_scale=g["cond_scale"], use_tqdm=False, noise_fn=lambda tag, shape: g["noise"][tag],
                          use_graph=use_graph, device="cpu")
    assert torch.equal(again, outs[-1])
    # stage 2 alone from the reference's stage-1 image
    alone = imagen.sample(text_embeds=g["text_embeds"], cond_scale=g["cond_scale"], use_tqdm=False, noise_fn=lambda tag, shape: g["noise"][tag],
                          start_at_unet_number=2, start_image_or_video=g["outputs"][0], device="cpu")
    assert nerr(alone, g["outputs"][1]) < 2e-2


def test_pipelined_batches_match_sequential(cpu_backend):
    """sample_pipelined (stage s of batch k overlapped with stage s+1 of batch k-1, one worker thread + lane per stage) returns,
    per batch, exactly what sample() returns for that batch."""
    g = torch.load(os.path.join(GOLDEN, "sample_tiny_cascade.pt"), weights_only=False)
    imagen = _cascade(g, timesteps=g["timesteps"])
    fns = [lambda tag, shape, s=s: g["noise"][tag] * s for s in (1.0, 0.5, -1.0)]
    common = dict(cond_scale=g["cond_scale"], device="cpu")
    batches = [dict(text_embeds=g["text_embeds"] * (1.0 + 0.1 * i), noise_fn=fn) for i, fn in enumerate(fns)]
    seq = [imagen.sample(use_tqdm=False, **common, **b) for b in batches]
    pipe = imagen.sample_pipelined(batches, **common)
    assert len(pipe) == len(seq)
    for a, b in zip(pipe, seq):
        assert torch.equal(a, b)
    assert nerr(pipe[0], g["outputs"][1]) < 2e-2
    assert not torch.equal(pipe[0], pipe[1])
    # errors inside a stage worker surface in the caller
    with pytest.raises(AssertionError):
        imagen.sample_pipelined([dict(text_embeds=g["text_embeds"][..., :-1])], **common)


@pytest.mark.parametrize("run", ["init_skip", "inpaint"])
def test_sample_options_driver(cpu_backend, run):
    g = torch.load(os.path.join(GOLDEN, "sample_tiny_options.pt"), weights_only=False)
    r = g["runs"][run]
    imagen = _cascade(g, timesteps=g["timesteps"])
    kw = {k: r[k] for k in ("init_images", "skip_steps", "inpaint_images", "inpaint_masks", "inpaint_resample_times") if k in r}
    outs = imagen.sample(text_embeds=g["text_embeds"], cond_scale=g["cond_scale"], use_tqdm=False, return_all_unet_outputs=True,
                         noise_fn=lambda tag, shape: r["noise"][tag], device="cpu", **kw)
    errs = [nerr(o, ref) for o, ref in zip(outs, r["outputs"])]
    assert max(errs) < 2e-2, errs
    if run == "inpaint":
        m = r["inpaint_masks"][:, None].expand(-1, 3, -1, -1)
        assert torch.allclose(outs[-1][m], r["inpaint_images"][m], atol=1e-6)


@pytest.mark.parametrize("tds", [(1, 1), (2, 1)])
def test_video_sample_driver(cpu_backend, tds):
    g = torch.load(os.path.join(GOLDEN, "sample_tiny_video.pt"), weights_only=False)
    run = g if tds == (1, 1) else g["tds"]
    imagen = _cascade(g, timesteps=g["timesteps"], temporal_downsample_factor=tds)
    outs = imagen.sample(text_embeds=g["text_embeds"], video_frames=g["frames"], cond_scale=g["cond_scale"], use_tqdm=False,
                         return_all_unet_outputs=True, noise_fn=lambda tag, shape: run["noise"][tag], device="cpu")
    assert [tuple(o.shape) for o in outs] == [tuple(o.shape) for o in run["outputs"]]
    errs = [nerr(o, ref) for o, ref in zip(outs, run["outputs"])]
    assert max(errs) < 2e-2, errs


@pytest.mark.parametrize("tag", ["cond_pre", "cond_post", "cond_both", "init_skip", "inpaint", "cond_pre_tds"])
def test_video_sample_options_driver(cpu_backend, tag):
    """Video-stage options of sample() — prompt frames (cond_video_frames / post_cond_video_frames), init videos + skip_steps, video
    inpainting — through the real driver and the planner's strided frame copies, against recorded runs of the live reference."""
    o = torch.load(os.path.join(GOLDEN, "sample_tiny_video_options.pt"), weights_only=False)
    g = torch.load(os.path.join(GOLDEN, o["weights_from"]), weights_only=False)
    run = o["runs"][tag]
    imagen = _cascade(g, timesteps=o["timesteps"], temporal_downsample_factor=run.get("temporal_downsample_factor", 1))
    outs = imagen.sample(text_embeds=g["text_embeds"], video_frames=o["frames"], cond_scale=g["cond_scale"], use_tqdm=False,
                         return_all_unet_outputs=True, noise_fn=lambda t, shape: run["noise"][t], device="cpu", **run["kwargs"])
    assert [tuple(x.shape) for x in outs] == [tuple(x.shape) for x in run["outputs"]]
    errs = [nerr(a, ref) for a, ref in zip(outs, run["outputs"])]
    assert max(errs) < 2e-2, errs
    if tag == "inpaint":
        m = run["kwargs"]["inpaint_masks"][:, None].expand(-1, 3, -1, -1, -1)
        assert torch.allclose(outs[-1][m], run["kwargs"]["inpaint_videos"][m], atol=1e-6)


def test_elucidated_sample_driver(cpu_backend):
    from imagen_pytorch_amd import ElucidatedImagen

    g = torch.load(os.path.join(GOLDEN, "sample_tiny_elucidated.pt"), weights_only=False)
    model = _cascade(g, klass=ElucidatedImagen, **g["hparams"])
    nf = lambda tag, shape: g["noise"][tag]
    outs = model.sample(text_embeds=g["text_embeds"], cond_scale=g["cond_scale"], use_tqdm=False, return_all_unet_outputs=True, noise_fn=nf,
                        device="cpu")
    e0 = nerr(outs[0], g["outputs"][0])
    alone = model.sample(text_embeds=g["text_embeds"], cond_scale=g["cond_scale"], use_tqdm=False, noise_fn=nf, start_at_unet_number=2,
                         start_image_or_video=g["outputs"][0], device="cpu")
    e1 = nerr(alone, g["outputs"][1])
    assert e0 < 3e-2 and e1 < 3e-2, (e0, e1)    # cf. tests/test_plan_interp.py::test_elucidated_stage_plan_on_cpu


@pytest.mark.parametrize("which", ["model", "ema"])
def test_checkpoint_sample_driver(cpu_backend, which, tmp_path):
    from imagen_pytorch_amd import load_imagen_from_checkpoint

    g = torch.load(os.path.join(GOLDEN, "checkpoint_tiny.pt"), weights_only=False)
    path = tmp_path / "ckpt.pt"
    torch.save(g["checkpoint"], str(path))
    imagen = load_imagen_from_checkpoint(path, load_ema_if_available=which == "ema").eval()
    exp = g["expected"][which]
    out = imagen.sample(text_embeds=g["text_embeds"], cond_scale=g["cond_scale"], use_tqdm=False, noise_fn=lambda tag, shape: exp["noise"][tag],
                        device="cpu")
    assert nerr(out, exp["output"]) < 2e-2


def test_conditioning_handle_driver(cpu_backend):
    g = torch.load(os.path.join(GOLDEN, "sample_tiny_cascade.pt"), weights_only=False)
    imagen = _cascade(g, timesteps=g["timesteps"])
    nf = lambda tag, shape: g["noise"][tag]
    ref = imagen.sample(text_embeds=g["text_embeds"], cond_scale=3.0, use_tqdm=False, noise_fn=nf, device="cpu")
    cond = imagen.prepare_conditioning(text_embeds=g["text_embeds"], device="cpu")
    a = imagen.sample(conditioning=cond, cond_scale=3.0, use_tqdm=False, noise_fn=nf, device="cpu")
    engines = [st["eng"] for st in imagen._stages.values()]
    runs = [e.static_runs for e in engines]
    b = imagen.sample(conditioning=cond, cond_scale=3.0, use_tqdm=False, noise_fn=nf, device="cpu")
    assert torch.equal(a, ref) and torch.equal(b, ref)
    assert [e.static_runs for e in engines] == runs, "the second call with the same handle must not re-run the static plans"


def test_video_elucidated_sample_driver(cpu_backend):
    from imagen_pytorch_amd import ElucidatedImagen

    g = torch.load(os.path.join(GOLDEN, "sample_tiny_video.pt"), weights_only=False)
    e = g["edm"]
    model = _cascade(g, klass=ElucidatedImagen, **e["hparams"])
    nf = lambda tag, shape: e["noise"][tag]
    outs = model.sample(text_embeds=g["text_embeds"], video_frames=g["frames"], cond_scale=g["cond_scale"], use_tqdm=False,
                        return_all_unet_outputs=True, noise_fn=nf, device="cpu")
    assert tuple(outs[0].shape) == tuple(e["outputs"][0].shape)
    e0 = nerr(outs[0], e["outputs"][0])
    alone = model.sample(text_embeds=g["text_embeds"], video_frames=g["frames"], cond_scale=g["cond_scale"], use_tqdm=False, noise_fn=nf,
                         start_at_unet_number=2, start_image_or_video=e["outputs"][0], device="cpu")
    e1 = nerr(alone, e["outputs"][1])
    # a Heun trajectory of dim-8 toy video unets is a chaotic quantity: 2.9e-2 / 0.98e-2 with the output stage's weights rounded to fp16,
    # 4.2e-2 / 0.79e-2 with the split-precision output stage of round 5 (a MORE accurate single forward) — the bar of the GPU twin
    # (tests/test_video_gpu.py: 5e-2, measured 1.5-3.2e-2 over the rounds); the per-step forward is held to its own bar elsewhere
    assert e0 < 5e-2 and e1 < 5e-2, (e0, e1)


@pytest.mark.parametrize("tag", ["init_skip", "inpaint", "sigma"])
def test_elucidated_sample_options_driver(cpu_backend, tag):
    """ElucidatedImagen.sample(init_images= + skip_steps= | inpaint_images= ... | sigma_min= / sigma_max=) through the real driver:
    table rows per inner iteration, the blend / re-noising launches, the step-counter start — vs recorded runs of the live reference."""
    from imagen_pytorch_amd import ElucidatedImagen

    o = torch.load(os.path.join(GOLDEN, "sample_tiny_elucidated_options.pt"), weights_only=False)
    g = torch.load(os.path.join(GOLDEN, o["weights_from"]), weights_only=False)
    run = o["runs"][tag]
    model = _cascade(g, klass=ElucidatedImagen, **g["hparams"])
    nf = lambda t, shape: run["noise"][t]
    for use_graph in (True, False):
        outs = model.sample(text_embeds=g["text_embeds"], cond_scale=g["cond_scale"], use_tqdm=False, return_all_unet_outputs=True,
                            noise_fn=nf, device="cpu", use_graph=use_graph, **run["kwargs"])
        e0 = nerr(outs[0], run["outputs"][0])
        assert e0 < 3e-2, (tag, use_graph, e0)
    alone = model.sample(text_embeds=g["text_embeds"], cond_scale=g["cond_scale"], use_tqdm=False, noise_fn=nf, device="cpu",
                         start_at_unet_number=2, start_image_or_video=run["outputs"][0], **run["kwargs"])
    e1 = nerr(alone, run["outputs"][1])
    assert e1 < 3e-2, (tag, e1)
    if tag == "inpaint":
        m = run["kwargs"]["inpaint_masks"][:, None].expand(-1, 3, -1, -1)
        assert torch.allclose(alone[m], run["kwargs"]["inpaint_images"][m], atol=1e-6)


def test_video_elucidated_prompt_frames_driver(cpu_backend):
    """ElucidatedImagen over Unet3D stages with cond_video_frames (el.py:679-695) vs a recorded run of the live reference."""
    from imagen_pytorch_amd import ElucidatedImagen

    o = torch.load(os.path.join(GOLDEN, "sample_tiny_video_options.pt"), weights_only=False)
    g = torch.load(os.path.join(GOLDEN, o["weights_from"]), weights_only=False)
    run = o["runs"]["edm_cond_pre"]
    model = _cascade(g, klass=ElucidatedImagen, **run["hparams"])
    nf = lambda tag, shape: run["noise"][tag]
    common = dict(text_embeds=g["text_embeds"], video_frames=o["frames"], cond_scale=g["cond_scale"], use_tqdm=False, noise_fn=nf,
                  device="cpu", **run["kwargs"])
    outs = model.sample(return_all_unet_outputs=True, **common)
    e0 = nerr(outs[0], run["outputs"][0])
    alone = model.sample(start_at_unet_number=2, start_image_or_video=run["outputs"][0], **common)
    e1 = nerr(alone, run["outputs"][1])
    assert e0 < 3e-2 and e1 < 3e-2, (e0, e1)


def test_reference_step_level_api(cpu_backend, monkeypatch):
    """Imagen.p_sample_loop / p_sample / p_mean_variance with the reference's signatures (ip.py:2042-2289): the cascade strung
    together from single steps reproduces the reference's recorded run, and agrees with the graph path of sample()."""
    from step_api_case import run_cascade_by_steps

    g = torch.load(os.path.join(GOLDEN, "sample_tiny_cascade.pt"), weights_only=False)
    imagen = _cascade(g, timesteps=g["timesteps"])
    outs = run_cascade_by_steps(imagen, g, monkeypatch, torch.device("cpu"))
    errs = [nerr(o, r) for o, r in zip(outs, g["outputs"])]
    assert max(errs) < 2e-2, errs
    fused = imagen.sample(text_embeds=g["text_embeds"], cond_scale=g["cond_scale"], use_tqdm=False, return_all_unet_outputs=True,
                          noise_fn=lambda tag, shape: g["noise"][tag], device="cpu")
    assert max(nerr(a, b) for a, b in zip(outs, fused)) < 5e-3
    # prompt frames handed to an image cascade are ignored, as in the reference (ip.py:2057-2070: video_kwargs only if is_video)
    sched = imagen.noise_schedulers[0]
    x = torch.zeros(1, 3, 16, 16)
    te1 = g["text_embeds"][:1]
    kw = dict(t=torch.ones(1), t_next=torch.full((1,), 0.5), noise_scheduler=sched, text_embeds=te1, text_mask=torch.any(te1 != 0., dim=-1))
    (m0, _, _), x0 = imagen.p_mean_variance(imagen.unets[0], x, **kw)
    (m1, _, _), x1 = imagen.p_mean_variance(imagen.unets[0], x, cond_video_frames=x[:, :, None], **kw)
    assert torch.equal(m0, m1) and torch.equal(x0, x1)
    with pytest.raises(AssertionError):      # a conditioning image for a unet built without cond_images_channels (ip.py:1555)
        imagen.p_sample(imagen.unets[0], x, torch.ones(1), t_next=torch.zeros(1), noise_scheduler=sched, cond_images=x)


@pytest.mark.parametrize("cond_ch,self_cond", [(0, True), (4, True)], ids=["self_cond", "cond_images+self_cond"])
def test_self_conditioned_sampling(cpu_backend, monkeypatch, cond_ch, self_cond):
    """Unet(self_cond=True) stages (ip.py:1541-1543, 2249): every step's denoiser reads the previous step's thresholded x0 — in the graph
    path from the buffer DDPM_UPDATE leaves it in, in the step-level API from p_sample's second return value."""
    from step_api_case import cond_images_cascade, recorded_noise

    imagen, te, cond, noise_fn, want, _ = cond_images_cascade(torch.device("cpu"), cond_ch=cond_ch, self_cond=self_cond)
    extra = {} if cond is None else dict(cond_images=cond)
    outs = imagen.sample(text_embeds=te, cond_scale=3., use_tqdm=False, return_all_unet_outputs=True, noise_fn=noise_fn, device="cpu", **extra)
    errs = [nerr(o, w) for o, w in zip(outs, want)]
    assert max(errs) < 2e-2, errs
    again = imagen.sample(text_embeds=te, cond_scale=3., use_tqdm=False, noise_fn=noise_fn, device="cpu", **extra)
    assert torch.equal(again, outs[-1]), "the self-conditioning buffer is reset at the start of every loop"
    # stage 1 step by step through the reference's API
    T = imagen.noise_schedulers[0].num_timesteps
    draws = [noise_fn(("init", 0), (2, 3, 16, 16))] + [noise_fn(("step", 0, i), (2, 3, 16, 16)) for i in range(T)]
    with recorded_noise(monkeypatch, draws, torch.device("cpu")):
        img = imagen.p_sample_loop(imagen.unets[0], (2, 3, 16, 16), noise_scheduler=imagen.noise_schedulers[0], text_embeds=te,
                                   text_mask=torch.any(te != 0., dim=-1), cond_scale=3., use_tqdm=False, **extra)
    assert nerr(img, want[0]) < 2e-2


def test_bilinear_resize_mode_sampling(cpu_backend):
    """Imagen(resize_mode='bilinear') / Unet(resize_mode='bilinear') (ip.py:1559, 1924): the low-res conditioning image and the
    conditioning image are resized on the host with that mode (the in-kernel nearest resize becomes the identity)."""
    from step_api_case import cond_images_cascade

    imagen, te, cond, noise_fn, want, _ = cond_images_cascade(torch.device("cpu"), resize_mode="bilinear")
    outs = imagen.sample(text_embeds=te, cond_images=cond, cond_scale=3., use_tqdm=False, return_all_unet_outputs=True, noise_fn=noise_fn,
                         device="cpu")
    errs = [nerr(o, w) for o, w in zip(outs, want)]
    assert max(errs) < 2e-2, errs
    _, _, _, _, want_nearest, _ = cond_images_cascade(torch.device("cpu"))
    assert nerr(want[-1], want_nearest[-1]) > 1e-2, "the mode must matter for this test to say anything"


def test_cond_images_sampling(cpu_backend):
    """sample(cond_images=...) (ip.py:2324, 2465 -> Unet.forward ip.py:1555-1560): the image is packed once per stage, resized to the
    stage's resolution, and read by the init conv as a second input."""
    from step_api_case import cond_images_cascade

    imagen, te, cond, noise_fn, want, _ = cond_images_cascade(torch.device("cpu"))
    outs = imagen.sample(text_embeds=te, cond_images=cond, cond_scale=3., use_tqdm=False, return_all_unet_outputs=True, noise_fn=noise_fn,
                         device="cpu")
    errs = [nerr(o, w) for o, w in zip(outs, want)]
    assert max(errs) < 2e-2, errs
    other = imagen.sample(text_embeds=te, cond_images=cond.flip(0), cond_scale=3., use_tqdm=False, noise_fn=noise_fn, device="cpu")
    assert nerr(other, want[-1]) > 5e-2, "the conditioning image must matter"
    with pytest.raises(AssertionError):
        imagen.sample(text_embeds=te, cond_scale=3., use_tqdm=False, noise_fn=noise_fn, device="cpu")     # unet expects one (ip.py:1555)


@pytest.mark.parametrize("tag", ["plain", "cond_pre", "inpaint"])
def test_video_step_level_api(cpu_backend, monkeypatch, tag):
    """The reference's step-level methods on a video stage (ip.py:2042-2289 with 5-D shapes): the base Unet3D stage of the tiny video
    cascade run by p_sample_loop with the recorded draws of the reference's sample() — plain, with prompt frames, with video inpainting."""
    from step_api_case import recorded_noise

    o = torch.load(os.path.join(GOLDEN, "sample_tiny_video_options.pt"), weights_only=False)
    g = torch.load(os.path.join(GOLDEN, o["weights_from"]), weights_only=False)
    run = g if tag == "plain" else o["runs"][tag]
    kw = {} if tag == "plain" else dict(run["kwargs"])
    imagen = _cascade(g, timesteps=g["timesteps"])
    from imagen_pytorch_amd import imagen as imagen_mod
    monkeypatch.setattr(imagen_mod.Imagen, "device", property(lambda self: torch.device("cpu")))
    T, B, Fr, S = g["timesteps"], g["text_embeds"].shape[0], g["frames"], g["image_sizes"][0]
    R = kw.get("inpaint_resample_times", 1)
    draws = [run["noise"][("init", 0)]]
    for i in range(T):
        for r in reversed(range(R)):
            if tag == "inpaint":
                draws.append(run["noise"][("inpaint", 0, i, r)])
                draws.append(run["noise"][("step", 0, i, r)])
                if r > 0 and i < T - 1:
                    draws.append(run["noise"][("renoise", 0, i, r)])
            else:
                draws.append(run["noise"][("step", 0, i)])
    te = g["text_embeds"]
    with recorded_noise(monkeypatch, draws, torch.device("cpu")) as left:
        img = imagen.p_sample_loop(imagen.unets[0], (B, 3, Fr, S, S), noise_scheduler=imagen.noise_schedulers[0], text_embeds=te,
                                   text_mask=torch.any(te != 0., dim=-1), cond_scale=g["cond_scale"], use_tqdm=False, **kw)
        assert not left
    assert nerr(img, run["outputs"][0]) < 2e-2


def test_video_cond_images_sampling(cpu_backend):
    """sample(video_frames=..., cond_images=...) over Unet3D(cond_images_channels=4) stages (iv.py:1722-1731, ip.py:2465): the image is packed
    once per stage onto every frame; the real driver replayed against the oracle (itself pinned to the live reference)."""
    from imagen_pytorch_amd import Imagen, Unet3D
    from oracle import sampler_oracle as so
    from oracle.make_golden import TINY_3D, derandomise_unet3d

    k1 = {**TINY_3D, "cond_images_channels": 4}
    k2 = {**TINY_3D, "cond_images_channels": 4, "temporal_strides": (2, 1), "num_resnet_blocks": (1, 2)}
    torch.manual_seed(8)
    unets = [Unet3D(**k1), Unet3D(**k2)]
    imagen = Imagen(unets, image_sizes=(8, 16), timesteps=2, text_embed_dim=32, cond_drop_prob=0.1).eval()
    for u in imagen.unets:
        derandomise_unet3d(u)
    g = torch.Generator().manual_seed(12)
    te = torch.randn(2, 9, 32, generator=g)
    cond = torch.rand(2, 4, 12, 12, generator=g)              # neither stage's size: both resize it
    draws = {}

    def noise(tag, shape):
        if tag not in draws:
            draws[tag] = torch.randn(tuple(shape), generator=g)
        return draws[tag]

    sds = [({k: v.detach().clone() for k, v in u.state_dict().items()}, {**kw, "lowres_cond": i > 0})
           for i, (u, kw) in enumerate(zip(imagen.unets, (k1, k2)))]
    with torch.no_grad():
        want = so.imagen_sample(sds, (8, 16), te, timesteps=2, cond_scale=3., return_all=True, noise_fn=noise, cond_images=cond, video_frames=4)
    outs = imagen.sample(text_embeds=te, video_frames=4, cond_images=cond, cond_scale=3., use_tqdm=False, return_all_unet_outputs=True,
                         noise_fn=noise, device="cpu")
    errs = [nerr(o, w) for o, w in zip(outs, want)]
    assert max(errs) < 2e-2, errs
    other = imagen.sample(text_embeds=te, video_frames=4, cond_images=cond.flip(0), cond_scale=3., use_tqdm=False, noise_fn=noise, device="cpu")
    assert nerr(other, want[-1]) > 2e-2, "the conditioning image must matter"


def test_video_elucidated_inpainting_driver(cpu_backend):
    """ElucidatedImagen over Unet3D stages with inpaint_videos + resampling and an init video + skip_steps: the driver against the oracle
    (whose image-stage EDM options and video stages are each pinned to recorded runs of the live reference; this is their combination)."""
    from imagen_pytorch_amd import ElucidatedImagen
    from oracle import elucidated_oracle as eo

    g = torch.load(os.path.join(GOLDEN, "sample_tiny_video.pt"), weights_only=False)
    hp = g["edm"]["hparams"]
    model = _cascade(g, klass=ElucidatedImagen, **hp)
    gen = torch.Generator().manual_seed(21)
    B, Fr, S0, S1 = g["text_embeds"].shape[0], g["frames"], *g["image_sizes"]
    known = torch.rand(B, 3, Fr, S1, S1, generator=gen)
    keep = torch.rand(B, Fr, S1, S1, generator=gen) > 0.5
    init = torch.rand(B, 3, Fr, S0, S0, generator=gen)
    draws = {}

    def noise(tag, shape):
        if tag not in draws:
            draws[tag] = torch.randn(tuple(shape), generator=gen)
        return draws[tag]

    unets = [(u["state_dict"], u["kwargs"]) for u in g["unets"]]
    for kw in (dict(inpaint_images=known, inpaint_masks=keep, inpaint_resample_times=2), dict(init_images=init, skip_steps=1)):
        with torch.no_grad():
            want = eo.elucidated_sample(unets, g["image_sizes"], g["text_embeds"], hparams=hp, cond_scale=g["cond_scale"], noise_fn=noise,
                                        return_all=True, video_frames=Fr, **kw)
        pkw = {("inpaint_videos" if k == "inpaint_images" else k): v for k, v in kw.items()}
        outs = model.sample(text_embeds=g["text_embeds"], video_frames=Fr, cond_scale=g["cond_scale"], use_tqdm=False, return_all_unet_outputs=True,
                            noise_fn=noise, device="cpu", **pkw)
        e0 = nerr(outs[0], want[0])
        assert e0 < 3e-2, (list(kw), e0)
        assert outs[1].shape == want[1].shape
        if "inpaint_images" in kw:
            m = keep[:, None].expand(-1, 3, -1, -1, -1)
            assert torch.allclose(outs[1][m], known[m], atol=1e-6)
