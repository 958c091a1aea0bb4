"""Build-container only: the oracle restatement against the LIVE reference (imported through oracle/ref_shim.py)
on README-sized unets.  Skipped wherever /root/reference is absent (e.g. the GPU box)."""
import pytest
import torch

from oracle import ref_shim
from oracle import sampler_oracle as so
from oracle import unet_oracle as uo

pytestmark = [pytest.mark.reference, pytest.mark.skipif(not ref_shim.reference_available(), reason="reference tree not present")]

U1 = dict(dim=32, cond_dim=512, dim_mults=(1, 2, 4, 8), num_resnet_blocks=3, layer_attns=(False, True, True, True),
          layer_cross_attns=(False, True, True, True))
U2 = dict(dim=32, cond_dim=512, dim_mults=(1, 2, 4, 8), num_resnet_blocks=(2, 4, 8, 8), layer_attns=(False, False, False, True),
          layer_cross_attns=(False, False, False, True), lowres_cond=True)


def _dezero(u):
    torch.nn.init.normal_(u.final_conv.weight, std=0.05)
    torch.nn.init.normal_(u.final_conv.bias, std=0.05)


@pytest.mark.parametrize("kw", [U1, U2], ids=["readme-unet1", "readme-unet2"])
def test_unet_forward(kw):
    ip = ref_shim.load_reference()
    torch.manual_seed(0)
    u = ip.Unet(**kw).eval()
    _dezero(u)
    B, S = 1, 32
    x, t = torch.randn(B, 3, S, S), torch.tensor([0.3])
    te = torch.randn(B, 20, 768)
    extra = dict(lowres_cond_img=torch.randn(B, 3, S, S), lowres_noise_times=torch.tensor([0.5])) if kw.get("lowres_cond") else {}
    with torch.no_grad():
        for cdp in (0.0, 1.0):
            r = u(x, t, text_embeds=te, cond_drop_prob=cdp, **extra)
            o = uo.unet_forward(u.state_dict(), kw, x, t, text_embeds=te, cond_drop_prob=cdp, **extra)
            assert torch.allclose(r, o, atol=1e-5), (r - o).abs().max()


def test_state_dict_layout_is_interchangeable():
    """The drop-in Unet must load a reference state_dict strictly (same keys, shapes, order)."""
    from imagen_pytorch_amd.unet import Unet

    ip = ref_shim.load_reference()
    for kw in (U1, U2, dict(dim=16, dim_mults=(1, 2), memory_efficient=True, lowres_cond=True, attn_heads=2)):
        ref = ip.Unet(**kw)
        ours = Unet(**kw)
        assert list(ref.state_dict().keys()) == list(ours.state_dict().keys())
        ours.load_state_dict(ref.state_dict())


@pytest.mark.parametrize("cond_ch,self_cond,mode", [(0, False, "nearest"), (4, False, "nearest"), (0, True, "nearest"), (4, True, "nearest"),
                                                    (4, False, "bilinear")],
                         ids=["plain", "cond_images", "self_cond", "cond_images+self_cond", "cond_images-bilinear"])
def test_sampler_small_cascade(cond_ch, self_cond, mode):
    ip = ref_shim.load_reference()
    torch.manual_seed(0)
    k1 = dict(dim=8, cond_dim=32, dim_mults=(1, 2), num_resnet_blocks=1, layer_attns=(False, True), layer_cross_attns=(False, True), attn_heads=2,
              cond_images_channels=cond_ch, self_cond=self_cond, resize_mode=mode)
    k2 = dict(dim=8, cond_dim=32, dim_mults=(1, 2), num_resnet_blocks=(1, 2), layer_attns=(False, True), layer_cross_attns=(False, True),
              attn_heads=2, memory_efficient=True, cond_images_channels=cond_ch, self_cond=self_cond, resize_mode=mode)
    extra = dict(cond_images=torch.rand(2, cond_ch, 24, 24)) if cond_ch else {}
    im = ip.Imagen((ip.Unet(**k1), ip.Unet(**k2)), image_sizes=(16, 32), timesteps=4, text_embed_dim=768, cond_drop_prob=0.1, resize_mode=mode)
    for u in im.unets:
        _dezero(u)
    te = torch.randn(2, 7, 768)
    torch.manual_seed(123)
    ref = im.sample(text_embeds=te, cond_scale=3., use_tqdm=False, return_all_unet_outputs=True, **extra)
    torch.manual_seed(123)  # identical CPU RNG stream: the oracle draws in the reference's order
    unets = [(u.state_dict(), {**kw, "lowres_cond": i > 0}) for i, (u, kw) in enumerate(zip(im.unets, (k1, k2)))]
    got = so.imagen_sample(unets, (16, 32), te, timesteps=4, cond_scale=3., return_all=True, resize_mode=mode, **extra)
    for a, b in zip(ref, got):
        assert torch.allclose(a, b, atol=5e-4), (a - b).abs().max()


def test_unet3d_forward_readme_config():
    """SURVEY §8(f) NEXT-2 groundwork: oracle/unet3d_oracle.py vs the live `Unet3D` at the README video config's structure
    (`Unet3D(dim = 64, dim_mults = (1, 2, 4, 8))`, README.md:587; here dim 32 and an 8-frame 16x16 clip to keep the CPU test
    short), with temporal strides and the identity-initialised temporal layers randomised."""
    from oracle import unet3d_oracle as u3
    from oracle.make_golden import derandomise_unet3d

    iv = ref_shim.load_reference("imagen_video")
    kw = dict(dim=32, dim_mults=(1, 2, 4, 8), temporal_strides=(1, 1, 2, 2), layer_attns=(False, False, False, True))
    torch.manual_seed(0)
    u = iv.Unet3D(**kw).eval()
    derandomise_unet3d(u)
    x, t = torch.randn(1, 3, 8, 16, 16), torch.tensor([0.3])
    te = torch.randn(1, 20, 768)
    with torch.no_grad():
        for cdp in (0.0, 1.0):
            r = u(x, t, text_embeds=te, cond_drop_prob=cdp)
            o = u3.unet3d_forward(u.state_dict(), kw, x, t, text_embeds=te, cond_drop_prob=cdp)
            assert r.abs().mean() > 0.05
            assert torch.allclose(r, o, atol=1e-4, rtol=1e-4), (r - o).abs().max()


@pytest.mark.parametrize("lowres", [False, True], ids=["base", "lowres"])
@pytest.mark.parametrize("prompts", ["pre", "post", "both"])
def test_unet3d_forward_with_prompt_frames(lowres, prompts):
    """Unet3D.forward(cond_video_frames=, post_cond_video_frames=) of the live reference vs the oracle's restatement (iv.py:1682-1718,
    1933-1939), including the reference's frame order for succeeding frames and the low-res clip it extends for final_conv."""
    from oracle import unet3d_oracle as u3
    from oracle.make_golden import TINY_3D, derandomise_unet3d

    iv = ref_shim.load_reference("imagen_video")
    kw = {**TINY_3D, "lowres_cond": lowres}
    torch.manual_seed(3)
    u = iv.Unet3D(**kw).eval()
    derandomise_unet3d(u)
    B, Fr, S = 2, 4, 16
    x, t, te = torch.randn(B, 3, Fr, S, S), torch.tensor([0.3, -1.1]), torch.randn(B, 9, 32)
    size = S if lowres else S // 2
    pre = torch.rand(B, 3, 2, size, size) if prompts in ("pre", "both") else None
    post = torch.rand(B, 3, 4, size, size) if prompts in ("post", "both") else None
    extra = dict(lowres_cond_img=torch.randn(B, 3, Fr, S, S), lowres_noise_times=torch.tensor([0.9, 0.9])) if lowres else {}
    with torch.no_grad():
        for cdp in (0.0, 1.0):
            r = u(x, t, text_embeds=te, cond_drop_prob=cdp, cond_video_frames=pre, post_cond_video_frames=post, **extra)
            o = u3.unet3d_forward(u.state_dict(), kw, x, t, text_embeds=te, cond_drop_prob=cdp, cond_video_frames=pre,
                                  post_cond_video_frames=post, **extra)
            assert r.shape == x.shape and r.abs().mean() > 0.05
            assert torch.allclose(r, o, atol=1e-4, rtol=1e-4), (r - o).abs().max()


@pytest.mark.parametrize("lowres,prompt", [(False, False), (True, False), (True, True)], ids=["base", "lowres", "lowres+prompt"])
def test_unet3d_forward_with_cond_images(lowres, prompt):
    """Unet3D(cond_images_channels=...) (iv.py:1307-1310, 1722-1731) of the live reference vs the oracle: one conditioning image per sample,
    repeated over the frames (also the prompt frames'), resized, concatenated in front of the input channels."""
    from oracle import unet3d_oracle as u3
    from oracle.make_golden import TINY_3D, derandomise_unet3d

    iv = ref_shim.load_reference("imagen_video")
    kw = {**TINY_3D, "lowres_cond": lowres, "cond_images_channels": 5}
    torch.manual_seed(4)
    u = iv.Unet3D(**kw).eval()
    derandomise_unet3d(u)
    B, Fr, S = 2, 4, 16
    x, t, te = torch.randn(B, 3, Fr, S, S), torch.tensor([0.3, -1.1]), torch.randn(B, 9, 32)
    ci = torch.rand(B, 5, 8, 8)
    extra = dict(lowres_cond_img=torch.randn(B, 3, Fr, S, S), lowres_noise_times=torch.tensor([0.9, 0.9])) if lowres else {}
    if prompt:
        extra["cond_video_frames"] = torch.rand(B, 3, 2, S, S)
    with torch.no_grad():
        for cdp in (0.0, 1.0):
            r = u(x, t, text_embeds=te, cond_drop_prob=cdp, cond_images=ci, **extra)
            o = u3.unet3d_forward(u.state_dict(), kw, x, t, text_embeds=te, cond_drop_prob=cdp, cond_images=ci, **extra)
            assert r.shape == x.shape and torch.allclose(r, o, atol=1e-4, rtol=1e-4), (r - o).abs().max()
    from imagen_pytorch_amd import Unet3D
    ours = Unet3D(**kw).state_dict()
    assert list(ours.keys()) == list(u.state_dict().keys()) and all(ours[k].shape == v.shape for k, v in u.state_dict().items())


def _sweep_inputs(kw, seed=5):
    torch.manual_seed(seed)
    B, S = 2, 16
    x, t = torch.randn(B, kw.get("channels", 3), S, S), torch.tensor([0.4, -1.7])
    te = torch.randn(B, 7, kw["text_embed_dim"]) if kw.get("cond_on_text", True) else None
    extra = dict(lowres_cond_img=torch.randn(B, 3, S, S), lowres_noise_times=torch.tensor([0.9, 0.9])) if kw.get("lowres_cond") else {}
    from unet_config_sweep import cond_images_for, self_cond_for
    if kw.get("cond_images_channels", 0):
        extra["cond_images"] = cond_images_for(kw, B)
    if kw.get("self_cond", False):
        extra["self_cond"] = self_cond_for(kw, B)
    return x, t, te, extra


@pytest.mark.parametrize("name", list(__import__("unet_config_sweep").SWEEP))
def test_unet_oracle_config_sweep(name):
    """The oracle against the live reference over constructor-flag combinations beyond the README unets (the planner is then checked
    against the oracle for the same configurations in tests/test_plan_interp.py)."""
    from unet_config_sweep import SWEEP

    kw = SWEEP[name]
    ip = ref_shim.load_reference()
    torch.manual_seed(1)
    u = ip.Unet(**kw).eval()
    _dezero(u)
    from imagen_pytorch_amd import Unet
    ours = Unet(**kw).state_dict()
    ref_sd = u.state_dict()
    assert list(ours.keys()) == list(ref_sd.keys()) and all(ours[k].shape == ref_sd[k].shape for k in ref_sd), "state_dict layout"
    x, t, te, extra = _sweep_inputs(kw)
    with torch.no_grad():
        for cdp in (0.0, 1.0):
            r = u(x, t, text_embeds=te, cond_drop_prob=cdp, **extra)
            o = uo.unet_forward(u.state_dict(), kw, x, t, text_embeds=te, cond_drop_prob=cdp, **extra)
            assert r.abs().mean() > 1e-3
            assert torch.allclose(r, o, atol=2e-5, rtol=1e-4), (name, cdp, (r - o).abs().max())


def test_scheduler_tensor_api_is_bit_identical():
    """GaussianDiffusionContinuousTimes' tensor methods (ip.py:223-318) against the live class: same fp32 expressions, same bits."""
    from imagen_pytorch_amd.schedules import GaussianDiffusionContinuousTimes as Ours

    ip = ref_shim.load_reference()
    torch.manual_seed(0)
    x0, xt, n = torch.randn(3, 3, 8, 8), torch.randn(3, 3, 8, 8), torch.randn(3, 3, 8, 8)
    t, tn = torch.tensor([0.9, 0.5, 0.02]), torch.tensor([0.85, 0.4, 0.0])
    calls = [("q_posterior", (x0, xt, t), {}), ("q_posterior", (x0, xt, t), dict(t_next=tn)), ("q_sample", (x0, t, n), {}),
             ("q_sample", (x0, 0.3, n), {}), ("q_sample_from_to", (x0, tn, t, n), {}), ("q_sample_from_to", (x0, 0.2, 0.6, n), {}),
             ("predict_start_from_v", (xt, t, n), {}), ("predict_start_from_noise", (xt, t, n), {})]
    for ns in ("linear", "cosine"):
        a, b = Ours(noise_schedule=ns, timesteps=37), ip.GaussianDiffusionContinuousTimes(noise_schedule=ns, timesteps=37)
        for name, args, kw in calls:
            ra, rb = getattr(a, name)(*args, **kw), getattr(b, name)(*args, **kw)
            ra, rb = (ra if isinstance(ra, tuple) else (ra,)), (rb if isinstance(rb, tuple) else (rb,))
            assert len(ra) == len(rb)
            for u, v in zip(ra, rb):
                assert u.shape == v.shape and torch.equal(u, v), (ns, name)
        ta, tb = a.get_sampling_timesteps(4, device="cpu"), b.get_sampling_timesteps(4, device="cpu")
        assert len(ta) == len(tb)
        for (x, y), (p, q) in zip(ta, tb):
            assert torch.equal(x, p) and torch.equal(y, q)
        assert a.sample_random_times(5, device="cpu").shape == b.sample_random_times(5, device="cpu").shape


@pytest.mark.parametrize("objective", ["noise", "x_start", "v"])
@pytest.mark.parametrize("dyn", [True, False])
def test_step_level_posterior_math(objective, dyn):
    """Imagen.p_mean_variance / p_sample around a given model output (no denoiser call): identical to the live reference."""
    from imagen_pytorch_amd import Imagen, Unet

    ip = ref_shim.load_reference()
    kw = dict(dim=8, cond_dim=32, dim_mults=(1, 2), num_resnet_blocks=1, layer_attns=(False, True), layer_cross_attns=(False, True), attn_heads=2)
    ref = ip.Imagen((ip.Unet(**kw),), image_sizes=(16,), timesteps=10, text_embed_dim=768, cond_drop_prob=0.1)
    ours = Imagen((Unet(**kw),), image_sizes=(16,), timesteps=10, text_embed_dim=768, cond_drop_prob=0.1)
    torch.manual_seed(1)
    x, out = torch.randn(2, 3, 16, 16), 2.0 * torch.randn(2, 3, 16, 16)
    t, tn = torch.tensor([0.6, 0.1]), torch.tensor([0.5, 0.0])
    common = dict(t_next=tn, model_output=out, pred_objective=objective, dynamic_threshold=dyn)
    (m1, v1, l1), s1 = ref.p_mean_variance(ref.unets[0], x, t, noise_scheduler=ref.noise_schedulers[0], **common)
    (m2, v2, l2), s2 = ours.p_mean_variance(ours.unets[0], x, t, noise_scheduler=ours.noise_schedulers[0], **common)
    for u, v in ((m1, m2), (v1, v2), (l1, l2), (s1, s2)):
        assert u.shape == v.shape and torch.allclose(u, v, rtol=0, atol=1e-6)


def test_cross_embed_downsample_is_unbuildable_in_the_reference():
    """Why `Unet(cross_embed_downsample=True)` stays a NotImplementedError here: the reference's own constructor fails on it
    (`partial(CrossEmbedLayer, kernel_sizes=...)(dim_in, dim_out)`, ip.py:1315 / 1357 / 1366), so no model with that flag exists."""
    from imagen_pytorch_amd import Unet

    ip = ref_shim.load_reference()
    kw = dict(dim=8, dim_mults=(1, 2), cross_embed_downsample=True)
    with pytest.raises(TypeError):
        ip.Unet(**kw)
    with pytest.raises(NotImplementedError):
        Unet(**kw)
