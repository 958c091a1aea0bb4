"""Shared by the CPU replay and the MI355X tests: the tiny two-stage cascade of tests/golden/sample_tiny_cascade.pt sampled through the
reference's STEP-LEVEL methods (Imagen.p_sample_loop -> p_sample -> p_mean_variance, ip.py:2042-2289) the way the reference's own
sample() strings them together (ip.py:2436-2472), with torch's Gaussian draws replaced by the recorded ones of the reference run."""
import contextlib

import torch
import torch.nn.functional as F


@contextlib.contextmanager
def recorded_noise(monkeypatch, draws, device):
    """torch.randn / torch.randn_like hand out `draws` in order (shapes checked)."""
    queue = [d.to(device) for d in draws]

    def take(shape):
        d = queue.pop(0)
        assert tuple(d.shape) == tuple(shape), (d.shape, shape)
        return d.clone()

    with monkeypatch.context() as m:
        m.setattr(torch, "randn", lambda *shape, **kw: take(shape[0] if len(shape) == 1 and not isinstance(shape[0], int) else shape))
        m.setattr(torch, "randn_like", lambda t, **kw: take(t.shape))
        yield queue


def run_cascade_by_steps(imagen, g, monkeypatch, device):
    """Returns the [0, 1] images of both stages."""
    T = g["timesteps"]
    te = g["text_embeds"].to(device)
    mask = torch.any(te != 0., dim=-1)
    B = te.shape[0]
    outs = []
    img = None
    for idx, (unet, S) in enumerate(zip(imagen.unets, imagen.image_sizes)):
        sched = imagen.noise_schedulers[idx]
        kw = {}
        draws = []
        if unet.lowres_cond:     # ip.py:2443-2449
            level = imagen.lowres_sample_noise_level
            times = imagen.lowres_noise_schedule.get_times(B, level, device=device)
            low = imagen.normalize_img(F.interpolate(img, S, mode="nearest"))
            low, *_ = imagen.lowres_noise_schedule.q_sample(x_start=low, t=times, noise=g["noise"][("lowres", idx)].to(device))
            kw = dict(lowres_cond_img=low, lowres_noise_times=times)
        draws = [g["noise"][("init", idx)]] + [g["noise"][("step", idx, i)] for i in range(T)]
        with recorded_noise(monkeypatch, draws, device) as left:
            img = imagen.p_sample_loop(unet, (B, imagen.channels, S, S), noise_scheduler=sched, text_embeds=te, text_mask=mask,
                                       cond_scale=g["cond_scale"], pred_objective=imagen.pred_objectives[idx],
                                       dynamic_threshold=imagen.dynamic_thresholding[idx], use_tqdm=False, **kw)
            assert not left, "every recorded draw is consumed, in the reference's order"
        outs.append(img)
    return outs


def cond_images_cascade(device, timesteps=3, cond_ch=4, self_cond=False, resize_mode="nearest"):
    """A tiny two-stage cascade whose unets take a `cond_ch`-channel conditioning image (Unet(cond_images_channels=...), ip.py:1191-1194,
    1555-1560; None is returned for it when cond_ch = 0) and / or self-condition (Unet(self_cond=True), ip.py:1541-1543, 2249), its
    inputs, a recorded-noise function, the oracle's images for them and the oracle's (state_dict, kwargs) per stage."""
    from imagen_pytorch_amd import Imagen, Unet
    from oracle import sampler_oracle as so

    base = dict(dim=8, cond_dim=32, text_embed_dim=32, dim_mults=(1, 2), layer_attns=(False, True), layer_cross_attns=(False, True),
                attn_heads=2, max_text_len=16, attn_pool_num_latents=8, cond_images_channels=cond_ch, self_cond=self_cond, resize_mode=resize_mode)
    k1, k2 = dict(base, num_resnet_blocks=1), dict(base, num_resnet_blocks=(1, 2), memory_efficient=True)
    torch.manual_seed(3)
    unets = [Unet(**k1), Unet(**k2)]
    imagen = Imagen(unets, image_sizes=(16, 32), timesteps=timesteps, text_embed_dim=32, cond_drop_prob=0.1, resize_mode=resize_mode)
    for u in imagen.unets:                  # zero-initialised final conv: give it weights, or the test says nothing
        torch.nn.init.normal_(u.final_conv.weight, std=0.05)
        torch.nn.init.normal_(u.final_conv.bias, std=0.05)
    imagen = imagen.to(device).eval()
    g = torch.Generator().manual_seed(11)
    te = torch.randn(2, 7, 32, generator=g)
    cond = torch.rand(2, cond_ch, 24, 24, generator=g) if cond_ch else None      # neither stage's size: both resize it
    draws = {}

    def noise(tag, shape):
        if tag not in draws:
            draws[tag] = torch.randn(tuple(shape), generator=g)
        return draws[tag]

    sds = [({k: v.detach().cpu() for k, v in u.state_dict().items()}, {**kw, "lowres_cond": i > 0})
           for i, (u, kw) in enumerate(zip(imagen.unets, (k1, k2)))]
    with torch.no_grad():
        want = so.imagen_sample(sds, (16, 32), te, timesteps=timesteps, cond_scale=3., return_all=True, noise_fn=noise, cond_images=cond,
                                resize_mode=resize_mode)
    return imagen, te, cond, (lambda tag, shape: noise(tag, shape).to(device)), want, sds
