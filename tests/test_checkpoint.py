"""Checkpoint interchange with the reference trainer (SURVEY.md §8(f) NEXT-4) — host logic, CPU only.

tests/golden/checkpoint_tiny.pt holds a trainer-format checkpoint whose model was built by the live reference's own
`ImagenConfig(...).create()` (oracle/make_golden.py --checkpoint)."""
import os

import pytest
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def fixture():
    return torch.load(os.path.join(GOLDEN, "checkpoint_tiny.pt"), map_location="cpu", weights_only=False)


@pytest.fixture()
def ckpt_path(fixture, tmp_path):
    p = tmp_path / "ckpt.pt"
    torch.save(fixture["checkpoint"], str(p))
    return p


def _unet_tensors(fixture, which):
    if which == "model":
        return {k[len("unets.0."):]: v for k, v in fixture["checkpoint"]["model"].items()}
    return {k[len("0.ema_model."):]: v for k, v in fixture["checkpoint"]["ema"].items() if k.startswith("0.ema_model.")}


@pytest.mark.parametrize("use_ema", [False, True])
def test_load_imagen_from_checkpoint(fixture, ckpt_path, use_ema):
    from imagen_pytorch_amd import Imagen, load_imagen_from_checkpoint

    imagen = load_imagen_from_checkpoint(ckpt_path, load_ema_if_available=use_ema)
    assert type(imagen) is Imagen and len(imagen.unets) == 1 and imagen.image_sizes == (16,)
    assert imagen.noise_schedulers[0].num_timesteps == fixture["checkpoint"]["imagen_params"]["timesteps"]
    want = _unet_tensors(fixture, "ema" if use_ema else "model")
    got = imagen.unets[0].state_dict()
    assert list(got.keys()) == list(want.keys())
    for k in want:
        assert torch.equal(got[k], want[k]), k
    # the two weight sets really differ, so the assertion above distinguishes them
    other = _unet_tensors(fixture, "model" if use_ema else "ema")
    assert any(not torch.equal(got[k], other[k]) for k in want)
    # the reference can read back what we write: same top-level layout, same keys
    assert imagen._config["unets"][0]["attn_dim_head"] == 64
    assert list(imagen.state_dict().keys()) == list(fixture["checkpoint"]["model"].keys())


def test_load_without_weights_and_errors(fixture, ckpt_path, tmp_path):
    from imagen_pytorch_amd import load_imagen_from_checkpoint

    fresh = load_imagen_from_checkpoint(ckpt_path, load_weights=False)
    w = _unet_tensors(fixture, "model")
    assert any(not torch.equal(fresh.unets[0].state_dict()[k], w[k]) for k in w)   # random init, not the checkpoint
    bad = dict(fixture["checkpoint"])
    bad.pop("imagen_type")
    p = tmp_path / "noconfig.pt"
    torch.save(bad, str(p))
    with pytest.raises(ValueError, match="unknown imagen type"):                   # utils.py:33-34
        load_imagen_from_checkpoint(p)
    with pytest.raises(AssertionError, match="checkpoint not found"):
        load_imagen_from_checkpoint(tmp_path / "missing.pt")


def test_config_defaults_match_reference(fixture):
    """Defaults filled in by ImagenConfig / UnetConfig / ElucidatedImagenConfig (configs.py:42-49, 65-75, 109-127)."""
    from imagen_pytorch_amd import checkpoint as ck

    ref_unet = fixture["unet_config_defaults"]        # UnetConfig(dim=8, dim_mults=[1, 2]).dict() of the live reference
    mine = {"text_embed_dim": ck._default_text_embed_dim(), **ck._UNET_DEFAULTS, "dim": 8, "dim_mults": [1, 2]}
    assert mine == ref_unet
    ref_el = {k: v for k, v in fixture["elucidated_config"].items() if k not in ("unets", "image_sizes", "video", "text_embed_dim")}
    assert ck._ELUCIDATED_DEFAULTS == ref_el
    ref_im = {k: v for k, v in fixture["checkpoint"]["imagen_params"].items()
              if k not in ("unets", "image_sizes", "video", "text_embed_dim", "timesteps", "cond_drop_prob")}
    assert {k: v for k, v in ck._IMAGEN_DEFAULTS.items() if k not in ("timesteps", "cond_drop_prob")} == ref_im
    assert ck._IMAGEN_DEFAULTS["timesteps"] == 1000 and ck._IMAGEN_DEFAULTS["cond_drop_prob"] == 0.5


def test_imagen_from_config_validation(fixture):
    from imagen_pytorch_amd import ElucidatedImagen, NullUnet
    from imagen_pytorch_amd.checkpoint import imagen_from_config

    params = fixture["checkpoint"]["imagen_params"]
    with pytest.raises(ValueError, match="image sizes length"):                    # configs.py:77-81
        imagen_from_config("original", {**params, "image_sizes": [16, 32]})
    vid = imagen_from_config("original", {**params, "video": True})               # configs.py:87-93: every unet becomes a Unet3D
    assert vid.is_video and type(vid.unets[0]).__name__ == "Unet3D" and vid._config["video"] is True
    dflt = imagen_from_config("original", {**params, "unets": [{k: v for k, v in params["unets"][0].items()
                                                                if k not in ("attn_dim_head", "attn_heads")}]})
    att = [m for m in dflt.unets[0].modules() if type(m).__name__ == "AttentionP"]   # config defaults: 16 heads x 32 dims (configs.py:48-49)
    assert att and all(m.dim_head == 32 and m.heads == 16 for m in att)
    with pytest.raises(NotImplementedError, match="attn_dim_head"):                # head dims other than 64 / 32 have no kernel
        imagen_from_config("original", {**params, "unets": [{**params["unets"][0], "attn_dim_head": 48}]})
    el = imagen_from_config("elucidated", {k: v for k, v in fixture["elucidated_config"].items()
                                           if k != "unets"} | {"unets": [params["unets"][0]], "num_sample_steps": 4})
    assert type(el) is ElucidatedImagen and el._config["num_sample_steps"] == 4 and el._config["S_noise"] == 1.003
    two = imagen_from_config("original", {**params, "unets": [{"is_null": True}, {**params["unets"][0]}], "image_sizes": [16, 32]})
    assert isinstance(two.unets[0], NullUnet) and two.unets[1].lowres_cond


def test_trainer_checkpoint_roundtrip_and_partial_load(fixture, ckpt_path, tmp_path, capsys):
    """load_trainer_checkpoint (tr.py:743-768, only_model) into a hand-built model; save_checkpoint writes the same layout back."""
    from imagen_pytorch_amd import Imagen, Unet, load_trainer_checkpoint, save_checkpoint
    from imagen_pytorch_amd.checkpoint import ema_unet_state_dicts

    params = fixture["checkpoint"]["imagen_params"]
    kw = dict(params["unets"][0])
    imagen = Imagen([Unet(**kw)], image_sizes=(16,), timesteps=2, text_embed_dim=32)
    loaded = load_trainer_checkpoint(imagen, ckpt_path, use_ema=True)
    assert float(loaded["steps"][0]) == 12.0 and loaded["version"]
    want = _unet_tensors(fixture, "ema")
    assert all(torch.equal(v, want[k]) for k, v in imagen.unets[0].state_dict().items())
    assert load_trainer_checkpoint(imagen, tmp_path / "nope.pt", noop_if_not_exist=True) is None
    # a model with one differently-shaped layer: strict load fails -> the shape-tolerant fallback (tr.py:209-220) keeps going
    other = Imagen([Unet(**{**kw, "attn_pool_num_latents": 4})], image_sizes=(16,), timesteps=2, text_embed_dim=32)
    load_trainer_checkpoint(other, ckpt_path)
    assert "Trying partial load" in capsys.readouterr().out
    assert torch.equal(other.unets[0].state_dict()["final_conv.weight"], _unet_tensors(fixture, "model")["final_conv.weight"])
    # write -> read
    src = Imagen([Unet(**kw)], image_sizes=(16,), timesteps=2, text_embed_dim=32)
    load_trainer_checkpoint(src, ckpt_path)
    avg = Unet(**kw)
    avg.load_state_dict(want)
    out = tmp_path / "sub" / "resaved.pt"
    save_checkpoint(src, out, ema_unets=[avg])
    re = torch.load(str(out), map_location="cpu", weights_only=False)
    assert set(re) >= {"model", "version", "steps", "ema"} and "imagen_type" not in re       # hand-built model: no config (tr.py:728)
    assert list(re["model"].keys()) == list(fixture["checkpoint"]["model"].keys())
    assert set(re["ema"].keys()) == set(fixture["checkpoint"]["ema"].keys())
    assert all(torch.equal(v, want[k]) for k, v in ema_unet_state_dicts(re["ema"], 1)[0].items())


def test_config_classes(fixture):
    """ImagenConfig / ElucidatedImagenConfig / UnetConfig / Unet3DConfig / NullUnetConfig(...).create() (configs.py:36-160)."""
    from imagen_pytorch_amd import (ElucidatedImagen, ElucidatedImagenConfig, Imagen, ImagenConfig, NullUnet, NullUnetConfig, Unet, Unet3D,
                                    Unet3DConfig, UnetConfig)

    params = fixture["checkpoint"]["imagen_params"]
    im = ImagenConfig(**params).create()
    assert type(im) is Imagen and im._config["timesteps"] == params["timesteps"]
    el = ElucidatedImagenConfig(**{k: v for k, v in params.items() if k not in ("timesteps", "noise_schedules", "loss_type")}, num_sample_steps=3).create()
    assert type(el) is ElucidatedImagen and el.hparams[0].num_sample_steps == 3
    uk = dict(params["unets"][0])
    assert type(UnetConfig(**uk).create()) is Unet and type(Unet3DConfig(**uk).create()) is Unet3D and type(NullUnetConfig(is_null=True).create()) is NullUnet
