"""Generate the golden fixtures under tests/golden/ from the LIVE REFERENCE (build container only).

    python -m oracle.make_golden

The reference (lucidrains/imagen-pytorch at /root/reference) ships no golden vectors for the sampling path
(SURVEY.md §4), and it cannot travel to the GPU box, so this script imports it through oracle/ref_shim.py,
runs its own `Unet.forward`, `Unet.forward_with_cond_scale` and `Imagen.sample` on small seeded problems and
records inputs, weights (state_dict), every Gaussian draw and the outputs.  The fixtures pin
  * oracle/unet_oracle.py + oracle/sampler_oracle.py  (tests/test_oracle_golden.py, CPU), and
  * the HIP path                                       (tests/test_model_gpu.py, MI355X)
to the reference's actual numbers.  Nothing from the reference's sources is copied; only tensors are stored.
"""
from __future__ import annotations

import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle.ref_shim import load_reference  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")

TINY_BASE = dict(dim=8, cond_dim=32, text_embed_dim=32, dim_mults=(1, 2), num_resnet_blocks=1, layer_attns=(False, True),
                 layer_cross_attns=(False, True), attn_heads=2, attn_dim_head=64, max_text_len=16, attn_pool_num_latents=8)
TINY_SR = dict(dim=8, cond_dim=32, text_embed_dim=32, dim_mults=(1, 2), num_resnet_blocks=(1, 2), layer_attns=(False, True),
               layer_cross_attns=(False, True), attn_heads=2, attn_dim_head=64, max_text_len=16, attn_pool_num_latents=8, lowres_cond=True)


def _derandomise(unet):
    """final_conv is zero-initialised in the reference (ip.py:1438): a fresh Unet outputs exact zeros. Give it weights."""
    torch.nn.init.normal_(unet.final_conv.weight, std=0.3)
    torch.nn.init.normal_(unet.final_conv.bias, std=0.3)


def make_unet_fixture(ip, kwargs, size, seed, path):
    torch.manual_seed(seed)
    unet = ip.Unet(**kwargs).eval()
    _derandomise(unet)
    B = 2
    x = torch.randn(B, 3, size, size)
    time = torch.tensor([0.7, -2.3])                      # log-SNR conditions
    text_embeds = torch.randn(B, 11, kwargs["text_embed_dim"])
    text_mask = torch.ones(B, 11, dtype=torch.bool)
    text_mask[1, 7:] = False
    extra = {}
    if kwargs.get("lowres_cond"):
        extra = dict(lowres_cond_img=torch.randn(B, 3, size, size), lowres_noise_times=torch.tensor([1.1, 1.1]))
    with torch.no_grad():
        out_cond = unet(x, time, text_embeds=text_embeds, text_mask=text_mask, **extra)
        out_null = unet(x, time, text_embeds=text_embeds, text_mask=text_mask, cond_drop_prob=1., **extra)
        out_cfg = unet.forward_with_cond_scale(x, time, text_embeds=text_embeds, text_mask=text_mask, cond_scale=3., **extra)
    torch.save(dict(kwargs=kwargs, state_dict={k: v.clone() for k, v in unet.state_dict().items()}, x=x, time=time,
                    text_embeds=text_embeds, text_mask=text_mask, extra=extra, out_cond=out_cond, out_null=out_null, out_cfg=out_cfg,
                    generator="oracle/make_golden.py", reference="lucidrains/imagen-pytorch v2.0.0 Unet.forward (ip.py:1524-1725)"), path)
    print(f"wrote {path}: |out_cond| = {out_cond.abs().mean():.4f}")


def make_sample_fixture(ip, path, seed=7, T=3):
    torch.manual_seed(seed)
    u1, u2 = ip.Unet(**TINY_BASE), ip.Unet(**{k: v for k, v in TINY_SR.items() if k != "lowres_cond"})
    imagen = ip.Imagen((u1, u2), image_sizes=(16, 32), timesteps=T, text_embed_dim=32, cond_drop_prob=0.1).eval()
    for u in imagen.unets:
        _derandomise(u)
    text_embeds = torch.randn(2, 9, 32)
    # record every Gaussian draw of the reference, in order
    draws = []
    real_randn, real_randn_like = torch.randn, torch.randn_like

    def rec_randn(*a, **k):
        t = real_randn(*a, **k)
        draws.append(t.clone())
        return t

    def rec_randn_like(x, **k):
        t = real_randn_like(x, **k)
        draws.append(t.clone())
        return t

    torch.randn, torch.randn_like = rec_randn, rec_randn_like
    try:
        outs = imagen.sample(text_embeds=text_embeds, cond_scale=3., use_tqdm=False, return_all_unet_outputs=True)
    finally:
        torch.randn, torch.randn_like = real_randn, real_randn_like
    # order of draws (ip.py:2449, 2195, 2160): stage 0: init, T steps; stage 1: lowres aug, init, T steps
    noise = {}
    it = iter(draws)
    for stage in range(2):
        if stage > 0:
            noise[("lowres", stage)] = next(it)
        noise[("init", stage)] = next(it)
        for i in range(T):
            noise[("step", stage, i)] = next(it)
    assert next(it, None) is None
    unets = []
    for i, (u, kw) in enumerate(zip(imagen.unets, (TINY_BASE, TINY_SR))):
        unets.append(dict(kwargs={**{k: v for k, v in kw.items() if k != "lowres_cond"}, "lowres_cond": i > 0},
                          state_dict={k: v.clone() for k, v in u.state_dict().items()}))
    torch.save(dict(unets=unets, image_sizes=(16, 32), timesteps=T, cond_scale=3., text_embeds=text_embeds, noise=noise,
                    outputs=[o.clone() for o in outs], generator="oracle/make_golden.py",
                    reference="lucidrains/imagen-pytorch v2.0.0 Imagen.sample (ip.py:2291-2498)"), path)
    print(f"wrote {path}: out std {outs[-1].std():.4f}")


ELUCIDATED_HP = dict(num_sample_steps=5, sigma_min=0.002, sigma_max=80, sigma_data=0.5, rho=7, S_churn=80, S_tmin=0.05, S_tmax=50,
                     S_noise=1.003)


def make_elucidated_fixture(ip, el, path, seed=9):
    """ElucidatedImagen.sample (el.py:547-745) on the tiny 2-stage cascade, 5 Karras steps (4 of them with the Heun correction),
    every Gaussian draw recorded."""
    torch.manual_seed(seed)
    u1, u2 = ip.Unet(**TINY_BASE), ip.Unet(**{k: v for k, v in TINY_SR.items() if k != "lowres_cond"})
    model = el.ElucidatedImagen((u1, u2), image_sizes=(16, 32), text_embed_dim=32, cond_drop_prob=0.1, **ELUCIDATED_HP).eval()
    for u in model.unets:
        _derandomise(u)
    text_embeds = torch.randn(2, 9, 32)
    draws = []
    real_randn, real_randn_like = torch.randn, torch.randn_like

    def rec_randn(*a, **k):
        t = real_randn(*a, **k)
        draws.append(t.clone())
        return t

    def rec_randn_like(x, **k):
        t = real_randn_like(x, **k)
        draws.append(t.clone())
        return t

    torch.randn, torch.randn_like = rec_randn, rec_randn_like
    try:
        outs = model.sample(text_embeds=text_embeds, cond_scale=3., use_tqdm=False, return_all_unet_outputs=True)
    finally:
        torch.randn, torch.randn_like = real_randn, real_randn_like
    # order of draws (el.py:705, 442, 489): per stage: [lowres aug], init, one per step
    T = ELUCIDATED_HP["num_sample_steps"]
    noise = {}
    it = iter(draws)
    for stage in range(2):
        if stage > 0:
            noise[("lowres", stage)] = next(it)
        noise[("init", stage)] = next(it)
        for i in range(T):
            noise[("step", stage, i)] = next(it)
    assert next(it, None) is None
    unets = []
    for i, (u, kw) in enumerate(zip(model.unets, (TINY_BASE, TINY_SR))):
        unets.append(dict(kwargs={**{k: v for k, v in kw.items() if k != "lowres_cond"}, "lowres_cond": i > 0},
                          state_dict={k: v.clone() for k, v in u.state_dict().items()}))
    torch.save(dict(unets=unets, image_sizes=(16, 32), hparams=dict(ELUCIDATED_HP), cond_scale=3., text_embeds=text_embeds, noise=noise,
                    outputs=[o.clone() for o in outs], generator="oracle/make_golden.py",
                    reference="lucidrains/imagen-pytorch v2.0.0 ElucidatedImagen.sample (elucidated_imagen.py:547-745)"), path)
    print(f"wrote {path}: out std {outs[-1].std():.4f}")


TINY_3D = dict(dim=8, cond_dim=32, text_embed_dim=32, dim_mults=(1, 2), num_resnet_blocks=1, layer_attns=(False, True),
               layer_cross_attns=(False, True), attn_heads=2, attn_dim_head=64, max_text_len=16, attn_pool_num_latents=8,
               temporal_strides=(1, 2))


def derandomise_unet3d(unet):
    """Unet3D starts as an image Unet applied per frame: zero final_conv (iv.py:1578), dirac temporal convs (iv.py:415-417),
    zero out-norm gain of every temporal attention (`init_zero`, iv.py:496-497).  Randomise all three so the temporal paths count."""
    g = torch.Generator().manual_seed(1234)
    for name, prm in unet.named_parameters():
        if name.startswith("final_conv.") or ".temporal_conv." in name:
            prm.data.copy_(torch.randn(prm.shape, generator=g) * 0.3)
        elif name.endswith("fn.fn.to_out.1.g"):
            prm.data.copy_(1.0 + 0.2 * torch.randn(prm.shape, generator=g))


def make_unet3d_fixture(iv, path, seed=21):
    """Unet3D.forward of the live reference on a tiny clip (2 x 3 x 4 frames x 16 x 16), base and low-res-conditioned variants,
    plus `ignore_time` (the per-frame image path, iv.py:1736-1738)."""
    runs = {}
    for tag, kw in (("base", TINY_3D), ("sr", {**TINY_3D, "lowres_cond": True, "num_resnet_blocks": (1, 2)})):
        torch.manual_seed(seed)
        unet = iv.Unet3D(**kw).eval()
        derandomise_unet3d(unet)
        B, Fr, S = 2, 4, 16
        x = torch.randn(B, 3, Fr, S, S)
        time = torch.tensor([0.7, -2.3])
        text_embeds = torch.randn(B, 11, 32)
        text_mask = torch.ones(B, 11, dtype=torch.bool)
        text_mask[1, 7:] = False
        extra = dict(lowres_cond_img=torch.randn(B, 3, Fr, S, S), lowres_noise_times=torch.tensor([1.1, 1.1])) if kw.get("lowres_cond") else {}
        with torch.no_grad():
            out_cond = unet(x, time, text_embeds=text_embeds, text_mask=text_mask, **extra)
            out_null = unet(x, time, text_embeds=text_embeds, text_mask=text_mask, cond_drop_prob=1., **extra)
            out_cfg = unet.forward_with_cond_scale(x, time, text_embeds=text_embeds, text_mask=text_mask, cond_scale=3., **extra)
            out_notime = unet(x, time, text_embeds=text_embeds, text_mask=text_mask, ignore_time=True, **extra)
        runs[tag] = dict(kwargs=kw, state_dict={k: v.clone() for k, v in unet.state_dict().items()}, x=x, time=time, text_embeds=text_embeds,
                         text_mask=text_mask, extra=extra, out_cond=out_cond, out_null=out_null, out_cfg=out_cfg, out_notime=out_notime)
        print(f"unet3d[{tag}]: |out_cond| = {out_cond.abs().mean():.4f}, |cond - ignore_time| = {(out_cond - out_notime).abs().mean():.4f}")
    torch.save(dict(runs=runs, generator="oracle/make_golden.py --unet3d",
                    reference="lucidrains/imagen-pytorch v2.0.0 Unet3D.forward (imagen_video.py:1650-1941)"), path)
    print(f"wrote {path}")


def make_video_sample_fixture(ip, iv, path, seed=23, T=2, frames=4):
    """Imagen.sample of the live reference over two Unet3D stages (8 -> 16 pixels, `video_frames` = 4, CFG 3), every Gaussian draw
    recorded.  Draws per stage (ip.py:2449, 2195, 2160): [lowres augmentation], init, T steps — all of shape (b, c, f, h, w)."""
    torch.manual_seed(seed)
    kw1 = {**TINY_3D, "temporal_strides": (1, 2)}
    kw2 = {**TINY_3D, "temporal_strides": (2, 1), "num_resnet_blocks": (1, 2)}
    u1, u2 = iv.Unet3D(**kw1), iv.Unet3D(**kw2)
    imagen = ip.Imagen((u1, u2), image_sizes=(8, 16), timesteps=T, text_embed_dim=32, cond_drop_prob=0.1).eval()
    for u in imagen.unets:
        derandomise_unet3d(u)
    text_embeds = torch.randn(2, 9, 32)
    outs, draws = _record_draws(lambda: imagen.sample(text_embeds=text_embeds, video_frames=frames, cond_scale=3., use_tqdm=False,
                                                      return_all_unet_outputs=True))
    noise, it = {}, iter(draws)
    for stage in range(2):
        if stage > 0:
            noise[("lowres", stage)] = next(it)
        noise[("init", stage)] = next(it)
        for i in range(T):
            noise[("step", stage, i)] = next(it)
    assert next(it, None) is None
    unets = [dict(kwargs={**kw, "lowres_cond": i > 0}, state_dict={k: v.clone() for k, v in u.state_dict().items()})
             for i, (u, kw) in enumerate(zip(imagen.unets, (kw1, kw2)))]
    # same weights, first stage sampled at half the frame rate (temporal_downsample_factor = (2, 1), ip.py:1928-1935, 2383, 2441-2447)
    imagen2 = ip.Imagen(tuple(imagen.unets), image_sizes=(8, 16), timesteps=T, text_embed_dim=32, cond_drop_prob=0.1,
                        temporal_downsample_factor=(2, 1)).eval()
    outs2, draws2 = _record_draws(lambda: imagen2.sample(text_embeds=text_embeds, video_frames=frames, cond_scale=3., use_tqdm=False,
                                                        return_all_unet_outputs=True))
    noise2, it2 = {}, iter(draws2)
    for stage in range(2):
        if stage > 0:
            noise2[("lowres", stage)] = next(it2)
        noise2[("init", stage)] = next(it2)
        for i in range(T):
            noise2[("step", stage, i)] = next(it2)
    assert next(it2, None) is None and outs2[0].shape[2] == frames // 2
    tds = dict(temporal_downsample_factor=(2, 1), noise=noise2, outputs=[o.clone() for o in outs2])
    # same weights under the Karras et al. sampler (ElucidatedImagen over Unet3D stages, 3 steps)
    el = load_reference("elucidated_imagen")
    hp = dict(ELUCIDATED_HP, num_sample_steps=3)
    edm_model = el.ElucidatedImagen(tuple(imagen.unets), image_sizes=(8, 16), text_embed_dim=32, cond_drop_prob=0.1, **hp).eval()
    outs3, draws3 = _record_draws(lambda: edm_model.sample(text_embeds=text_embeds, video_frames=frames, cond_scale=3., use_tqdm=False,
                                                          return_all_unet_outputs=True))
    noise3, it3 = {}, iter(draws3)
    for stage in range(2):
        if stage > 0:
            noise3[("lowres", stage)] = next(it3)
        noise3[("init", stage)] = next(it3)
        for i in range(hp["num_sample_steps"]):
            noise3[("step", stage, i)] = next(it3)
    assert next(it3, None) is None
    edm = dict(hparams=hp, noise=noise3, outputs=[o.clone() for o in outs3])
    torch.save(dict(unets=unets, image_sizes=(8, 16), timesteps=T, frames=frames, cond_scale=3., text_embeds=text_embeds, noise=noise,
                    outputs=[o.clone() for o in outs], tds=tds, edm=edm, generator="oracle/make_golden.py --video",
                    reference="lucidrains/imagen-pytorch v2.0.0 Imagen.sample over Unet3D stages (ip.py:2291-2498, imagen_video.py)"), path)
    print(f"wrote {path}: outputs {[tuple(o.shape) for o in outs]}, std {outs[-1].std():.4f}, {len(draws)} draws")


def make_elucidated_options_fixture(ip, el, path, R=2, skip=1):
    """one_unet_sample options of the Karras et al. sampler (el.py:393-545) on the weights of sample_tiny_elucidated.pt (only inputs,
    draws and outputs are stored): (A) init_images + skip_steps, (B) inpainting with `inpaint_resample_times = R`, (C) per-call
    sigma_min / sigma_max overrides."""
    base = torch.load(os.path.join(GOLDEN, "sample_tiny_elucidated.pt"), weights_only=False)
    hp = base["hparams"]
    T = hp["num_sample_steps"]
    unets = []
    for spec in base["unets"]:
        u = ip.Unet(**spec["kwargs"])
        u.load_state_dict(spec["state_dict"])
        unets.append(u)
    model = el.ElucidatedImagen(tuple(unets), image_sizes=base["image_sizes"], text_embed_dim=32, cond_drop_prob=0.1, **hp).eval()
    for u, spec in zip(model.unets, base["unets"]):
        u.load_state_dict(spec["state_dict"])
    g = torch.Generator().manual_seed(37)
    S0, S1 = base["image_sizes"]
    init_images = torch.rand(2, 3, S0, S0, generator=g)
    inpaint_images = torch.rand(2, 3, S1, S1, generator=g)
    inpaint_masks = torch.rand(2, S1, S1, generator=g) > 0.5
    inpaint_masks[:, 8:20, 4:24] = True
    common = dict(text_embeds=base["text_embeds"], cond_scale=base["cond_scale"], use_tqdm=False, return_all_unet_outputs=True)
    torch.manual_seed(41)
    runs = {}

    def plain_tags(draws, first=0):
        noise, it = {}, iter(draws)
        for stage in range(2):
            if stage > 0:
                noise[("lowres", stage)] = next(it)
            noise[("init", stage)] = next(it)
            for i in range(first, T):
                noise[("step", stage, i)] = next(it)
        assert next(it, None) is None
        return noise

    kw = dict(init_images=init_images, skip_steps=skip)
    outs, draws = _record_draws(lambda: model.sample(**common, **kw))
    runs["init_skip"] = dict(kwargs=kw, noise=plain_tags(draws, first=skip), outputs=[o.clone() for o in outs])
    kw = dict(inpaint_images=inpaint_images, inpaint_masks=inpaint_masks, inpaint_resample_times=R)
    outs, draws = _record_draws(lambda: model.sample(**common, **kw))
    noise, it = {}, iter(draws)
    for stage in range(2):
        if stage > 0:
            noise[("lowres", stage)] = next(it)
        noise[("init", stage)] = next(it)
        for i in range(T):
            for r in reversed(range(R)):
                noise[("step", stage, i, r)] = next(it)
                if r > 0 and i < T - 1:
                    noise[("renoise", stage, i, r)] = next(it)
    assert next(it, None) is None
    runs["inpaint"] = dict(kwargs=kw, noise=noise, outputs=[o.clone() for o in outs])
    kw = dict(sigma_min=(0.01, 0.004), sigma_max=(40., 60.))
    outs, draws = _record_draws(lambda: model.sample(**common, **kw))
    runs["sigma"] = dict(kwargs=kw, noise=plain_tags(draws), outputs=[o.clone() for o in outs])
    torch.save(dict(weights_from="sample_tiny_elucidated.pt", runs=runs, generator="oracle/make_golden.py --elucidated-options",
                    reference="lucidrains/imagen-pytorch v2.0.0 ElucidatedImagen.sample with init_images / skip_steps / inpainting / "
                              "sigma overrides (elucidated_imagen.py:393-545, 547-745)"), path)
    for k, r in runs.items():
        print(f"wrote {path} [{k}]: out std {r['outputs'][-1].std():.4f}, {len(r['noise'])} draws")


def make_video_options_fixture(ip, iv, path, T=2, R=2, frames=4):
    """Video-stage options of Imagen.sample on the two Unet3D stages of sample_tiny_video.pt (same weights — only inputs, draws and
    outputs are stored here): prompt frames (`cond_video_frames`, `post_cond_video_frames`, both; iv.py:1682-1718, 1933-1939,
    ip.py:2417-2434), init videos + skip_steps, and video inpainting with resampling (ip.py:2196-2289)."""
    base = torch.load(os.path.join(GOLDEN, "sample_tiny_video.pt"), weights_only=False)
    assert base["frames"] == frames and base["timesteps"] == T
    unets = []
    for spec in base["unets"]:
        u = iv.Unet3D(**{k: v for k, v in spec["kwargs"].items() if k != "lowres_cond"}, lowres_cond=spec["kwargs"]["lowres_cond"])
        u.load_state_dict(spec["state_dict"])
        unets.append(u)
    imagen = ip.Imagen(tuple(unets), image_sizes=base["image_sizes"], timesteps=T, text_embed_dim=32, cond_drop_prob=0.1).eval()
    for u, spec in zip(imagen.unets, base["unets"]):       # cast_model_parameters may have re-instantiated (ip.py:1897-1903)
        u.load_state_dict(spec["state_dict"])
    text_embeds = base["text_embeds"]
    g = torch.Generator().manual_seed(29)
    S = base["image_sizes"][-1]
    pre = torch.rand(2, 3, 2, S, S, generator=g)           # at the last stage's size: that stage concatenates them with its low-res clip
    post = torch.rand(2, 3, 2, S, S, generator=g)
    init_video = torch.rand(2, 3, frames, base["image_sizes"][0], base["image_sizes"][0], generator=g)
    inpaint_videos = torch.rand(2, 3, frames, S, S, generator=g)
    inpaint_masks = torch.rand(2, frames, S, S, generator=g) > 0.5
    inpaint_masks[:, :, 4:10, 2:12] = True

    def plain_tags(draws, first=0):
        noise, it = {}, iter(draws)
        for stage in range(2):
            if stage > 0:
                noise[("lowres", stage)] = next(it)
            noise[("init", stage)] = next(it)
            for i in range(first, T):
                noise[("step", stage, i)] = next(it)
        assert next(it, None) is None
        return noise

    common = dict(text_embeds=text_embeds, video_frames=frames, cond_scale=base["cond_scale"], use_tqdm=False, return_all_unet_outputs=True)
    runs = {}
    torch.manual_seed(31)
    for tag, kw in (("cond_pre", dict(cond_video_frames=pre)), ("cond_post", dict(post_cond_video_frames=post)),
                    ("cond_both", dict(cond_video_frames=pre, post_cond_video_frames=post))):
        outs, draws = _record_draws(lambda: imagen.sample(**common, **kw))
        runs[tag] = dict(kwargs=kw, noise=plain_tags(draws), outputs=[o.clone() for o in outs])
    outs, draws = _record_draws(lambda: imagen.sample(**common, init_images=init_video, skip_steps=1))
    runs["init_skip"] = dict(kwargs=dict(init_images=init_video, skip_steps=1), noise=plain_tags(draws, first=1), outputs=[o.clone() for o in outs])
    kw = dict(inpaint_videos=inpaint_videos, inpaint_masks=inpaint_masks, inpaint_resample_times=R)
    outs, draws = _record_draws(lambda: imagen.sample(**common, **kw))
    noise, it = {}, iter(draws)
    for stage in range(2):
        if stage > 0:
            noise[("lowres", stage)] = next(it)
        noise[("init", stage)] = next(it)
        for i in range(T):
            for r in reversed(range(R)):
                noise[("inpaint", stage, i, r)] = next(it)
                noise[("step", stage, i, r)] = next(it)
                if r > 0 and i < T - 1:
                    noise[("renoise", stage, i, r)] = next(it)
    assert next(it, None) is None
    runs["inpaint"] = dict(kwargs=kw, noise=noise, outputs=[o.clone() for o in outs])
    # same prompt frames with the first stage at half the frame rate: the prompt is resized over time per stage (scale_video_time)
    imagen2 = ip.Imagen(tuple(imagen.unets), image_sizes=base["image_sizes"], timesteps=T, text_embed_dim=32, cond_drop_prob=0.1,
                        temporal_downsample_factor=(2, 1)).eval()
    pre4 = torch.rand(2, 3, 4, S, S, generator=g)
    outs, draws = _record_draws(lambda: imagen2.sample(**common, cond_video_frames=pre4))
    runs["cond_pre_tds"] = dict(kwargs=dict(cond_video_frames=pre4), temporal_downsample_factor=(2, 1), noise=plain_tags(draws),
                                outputs=[o.clone() for o in outs])
    # prompt frames under the Karras et al. sampler (el.py:679-695), hyper-parameters of sample_tiny_video.pt's EDM run
    el = load_reference("elucidated_imagen")
    hp = base["edm"]["hparams"]
    edm_model = el.ElucidatedImagen(tuple(imagen.unets), image_sizes=base["image_sizes"], text_embed_dim=32, cond_drop_prob=0.1, **hp).eval()
    outs, draws = _record_draws(lambda: edm_model.sample(**common, cond_video_frames=pre))
    noise, it = {}, iter(draws)
    for stage in range(2):
        if stage > 0:
            noise[("lowres", stage)] = next(it)
        noise[("init", stage)] = next(it)
        for i in range(hp["num_sample_steps"]):
            noise[("step", stage, i)] = next(it)
    assert next(it, None) is None
    runs["edm_cond_pre"] = dict(kwargs=dict(cond_video_frames=pre), hparams=hp, noise=noise, outputs=[o.clone() for o in outs])
    torch.save(dict(weights_from="sample_tiny_video.pt", timesteps=T, frames=frames, runs=runs, generator="oracle/make_golden.py --video-options",
                    reference="lucidrains/imagen-pytorch v2.0.0 Imagen.sample over Unet3D stages with cond_video_frames / "
                              "post_cond_video_frames / init_images + skip_steps / inpaint_videos (ip.py:2167-2498, imagen_video.py:1682-1718)"), path)
    for k, r in runs.items():
        print(f"wrote {path} [{k}]: outputs {[tuple(o.shape) for o in r['outputs']]}, std {r['outputs'][-1].std():.4f}, {len(r['noise'])} draws")


def _record_draws(fn):
    """Run fn() with torch.randn / randn_like recording every Gaussian draw, in call order."""
    draws = []
    real_randn, real_randn_like = torch.randn, torch.randn_like

    def rec_randn(*a, **k):
        t = real_randn(*a, **k)
        draws.append(t.clone())
        return t

    def rec_randn_like(x, **k):
        t = real_randn_like(x, **k)
        draws.append(t.clone())
        return t

    torch.randn, torch.randn_like = rec_randn, rec_randn_like
    try:
        out = fn()
    finally:
        torch.randn, torch.randn_like = real_randn, real_randn_like
    return out, draws


def make_sample_options_fixture(ip, path, seed=13, T=3, R=2, skip=1):
    """The p_sample_loop options outside the BASELINE configs (ip.py:2167-2289): (A) init_images + skip_steps, (B) inpainting
    with `inpaint_resample_times = R`, on the same tiny cascade."""
    torch.manual_seed(seed)
    u1, u2 = ip.Unet(**TINY_BASE), ip.Unet(**{k: v for k, v in TINY_SR.items() if k != "lowres_cond"})
    imagen = ip.Imagen((u1, u2), image_sizes=(16, 32), timesteps=T, text_embed_dim=32, cond_drop_prob=0.1).eval()
    for u in imagen.unets:
        _derandomise(u)
    text_embeds = torch.randn(2, 9, 32)
    init_images = torch.rand(2, 3, 16, 16)
    inpaint_images = torch.rand(2, 3, 32, 32)
    inpaint_masks = torch.rand(2, 32, 32) > 0.5
    inpaint_masks[:, 8:20, 4:24] = True
    runs = {}
    # (A) draws per stage (ip.py:2449, 2195, 2160): [lowres], init, one per remaining timestep
    outs, draws = _record_draws(lambda: imagen.sample(text_embeds=text_embeds, cond_scale=3., use_tqdm=False, return_all_unet_outputs=True,
                                                      init_images=init_images, skip_steps=skip))
    noise, it = {}, iter(draws)
    for stage in range(2):
        if stage > 0:
            noise[("lowres", stage)] = next(it)
        noise[("init", stage)] = next(it)
        for i in range(skip, T):
            noise[("step", stage, i)] = next(it)
    assert next(it, None) is None
    runs["init_skip"] = dict(noise=noise, outputs=[o.clone() for o in outs], init_images=init_images, skip_steps=skip)
    # (B) draws per inner iteration (ip.py:2244, 2160, 2269): known-image noise, step noise, re-noise unless r == 0 / last timestep
    outs, draws = _record_draws(lambda: imagen.sample(text_embeds=text_embeds, cond_scale=3., use_tqdm=False, return_all_unet_outputs=True,
                                                      inpaint_images=inpaint_images, inpaint_masks=inpaint_masks,
                                                      inpaint_resample_times=R))
    noise, it = {}, iter(draws)
    for stage in range(2):
        if stage > 0:
            noise[("lowres", stage)] = next(it)
        noise[("init", stage)] = next(it)
        for i in range(T):
            for r in reversed(range(R)):
                noise[("inpaint", stage, i, r)] = next(it)
                noise[("step", stage, i, r)] = next(it)
                if r > 0 and i < T - 1:
                    noise[("renoise", stage, i, r)] = next(it)
    assert next(it, None) is None
    runs["inpaint"] = dict(noise=noise, outputs=[o.clone() for o in outs], inpaint_images=inpaint_images, inpaint_masks=inpaint_masks,
                           inpaint_resample_times=R)
    unets = []
    for i, (u, kw) in enumerate(zip(imagen.unets, (TINY_BASE, TINY_SR))):
        unets.append(dict(kwargs={**{k: v for k, v in kw.items() if k != "lowres_cond"}, "lowres_cond": i > 0},
                          state_dict={k: v.clone() for k, v in u.state_dict().items()}))
    torch.save(dict(unets=unets, image_sizes=(16, 32), timesteps=T, cond_scale=3., text_embeds=text_embeds, runs=runs,
                    generator="oracle/make_golden.py --options",
                    reference="lucidrains/imagen-pytorch v2.0.0 Imagen.sample (ip.py:2291-2498) with init_images/skip_steps/inpainting"), path)
    for k, r in runs.items():
        print(f"wrote {path} [{k}]: out std {r['outputs'][-1].std():.4f}, {len(r['noise'])} draws")


def make_checkpoint_fixture(ip, path, seed=17, T=2):
    """A trainer-format checkpoint (tr.py:696-741) of a one-unet Imagen built by the reference's own `ImagenConfig(...).create()`,
    plus what the reference samples from its plain and from its EMA weights.  ema_pytorch is not installed, so the `ema` entry is
    laid out by hand as `nn.ModuleList([EMA(unet)]).state_dict()` would be ({i}.online_model.*, {i}.ema_model.*, {i}.initted,
    {i}.step); the EMA weights are a perturbed copy so that loading the wrong set is visible."""
    cfg = load_reference("configs")
    torch.manual_seed(seed)
    unet_kw = {k: (list(v) if k == "dim_mults" else v) for k, v in TINY_BASE.items()}
    params = dict(unets=[unet_kw], image_sizes=[16], timesteps=T, text_embed_dim=32, cond_drop_prob=0.1)
    imagen = cfg.ImagenConfig(**params).create().eval()
    _derandomise(imagen.unets[0])
    model_sd = {k: v.clone() for k, v in imagen.state_dict().items()}
    unet_sd = {k: v.clone() for k, v in imagen.unets[0].state_dict().items()}
    ema_sd = {k: (v * 0.9 + 0.02 * torch.randn_like(v) if v.is_floating_point() else v.clone()) for k, v in unet_sd.items()}
    ema = {f"0.online_model.{k}": v for k, v in unet_sd.items()}
    ema.update({f"0.ema_model.{k}": v for k, v in ema_sd.items()})
    ema["0.initted"], ema["0.step"] = torch.tensor([True]), torch.tensor([12])
    version = load_reference("version").__version__
    ckpt = dict(model=model_sd, version=version, steps=torch.tensor([12.]), ema=ema, imagen_type="original", imagen_params=imagen._config)
    text_embeds = torch.randn(2, 9, 32)
    expected = {}
    for which, sd in (("model", unet_sd), ("ema", ema_sd)):
        imagen.unets[0].load_state_dict(sd)
        out, draws = _record_draws(lambda: imagen.sample(text_embeds=text_embeds, cond_scale=3., use_tqdm=False))
        noise = {("init", 0): draws[0], **{("step", 0, i): d for i, d in enumerate(draws[1:])}}
        assert len(draws) == T + 1
        expected[which] = dict(output=out.clone(), noise=noise)
    el = cfg.ElucidatedImagenConfig(unets=[unet_kw], image_sizes=[16], text_embed_dim=32).create()
    torch.save(dict(checkpoint=ckpt, text_embeds=text_embeds, expected=expected, cond_scale=3., elucidated_config=el._config,
                    unet_config_defaults=cfg.UnetConfig(dim=8, dim_mults=[1, 2]).dict(),
                    generator="oracle/make_golden.py --checkpoint",
                    reference=f"lucidrains/imagen-pytorch v{version}: configs.ImagenConfig.create + ImagenTrainer.save layout (tr.py:696-741)"), path)
    print(f"wrote {path}: {len(model_sd)} model tensors, {len(ema)} ema entries, out std {expected['ema']['output'].std():.4f}")


def main():
    ip = load_reference()
    if "--video" in sys.argv:        # only the video sampling fixture
        make_video_sample_fixture(ip, load_reference("imagen_video"), os.path.join(GOLDEN, "sample_tiny_video.pt"))
        return
    if "--video-options" in sys.argv:   # only the video-stage options fixture (weights are those of sample_tiny_video.pt)
        make_video_options_fixture(ip, load_reference("imagen_video"), os.path.join(GOLDEN, "sample_tiny_video_options.pt"))
        return
    if "--elucidated-options" in sys.argv:   # only the EDM sampler-options fixture (weights are those of sample_tiny_elucidated.pt)
        make_elucidated_options_fixture(ip, load_reference("elucidated_imagen"), os.path.join(GOLDEN, "sample_tiny_elucidated_options.pt"))
        return
    if "--unet3d" in sys.argv:       # only the Imagen-Video denoiser fixture (SURVEY §8(f) NEXT-2 groundwork)
        make_unet3d_fixture(load_reference("imagen_video"), os.path.join(GOLDEN, "unet3d_tiny.pt"))
        return
    if "--checkpoint" in sys.argv:   # only the checkpoint-interchange fixture (SURVEY §8(f) NEXT-4)
        make_checkpoint_fixture(ip, os.path.join(GOLDEN, "checkpoint_tiny.pt"))
        return
    if "--options" in sys.argv:      # only the p_sample_loop-options fixture
        make_sample_options_fixture(ip, os.path.join(GOLDEN, "sample_tiny_options.pt"))
        return
    if "--elucidated" in sys.argv:   # only the NEXT-1 fixture (the others are committed and stay byte-identical)
        make_elucidated_fixture(ip, load_reference("elucidated_imagen"), os.path.join(GOLDEN, "sample_tiny_elucidated.pt"))
        return
    os.makedirs(GOLDEN, exist_ok=True)
    make_unet_fixture(ip, TINY_BASE, 16, 11, os.path.join(GOLDEN, "unet_tiny_base.pt"))
    make_unet_fixture(ip, TINY_SR, 32, 12, os.path.join(GOLDEN, "unet_tiny_sr.pt"))
    make_sample_fixture(ip, os.path.join(GOLDEN, "sample_tiny_cascade.pt"))


if __name__ == "__main__":
    main()
