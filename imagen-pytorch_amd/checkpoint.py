"""Checkpoint interchange with the reference trainer (SURVEY.md §8(f) NEXT-4).

A model trained with lucidrains/imagen-pytorch is saved by `ImagenTrainer.save` (tr.py:677-741) as ONE `torch.save` dict:

    model          Imagen.state_dict()                         keys `unets.{i}.<unet key>`
    ema            nn.ModuleList([EMA(unet), ...]).state_dict()  keys `{i}.ema_model.<unet key>`, `{i}.online_model.<unet key>`,
                                                                 `{i}.initted`, `{i}.step`   (only with use_ema, tr.py:723-724)
    imagen_type    'original' | 'elucidated'                     (only when the model was built from a config, tr.py:728-736)
    imagen_params  the constructor kwargs (`ImagenConfig(...).dict()`, configs.py:105-107)
    version, steps, optim{i}, scaler{i}, scheduler{i}, warmup{i}  — training state, ignored here

This module is the sampling-side reader of that file: it rebuilds the MI355X `Imagen` / `ElucidatedImagen` from
`imagen_params` and loads the plain or the EMA weights, so trained checkpoints (not only random-init models) can be sampled
and benchmarked on the HIP path.  It mirrors
  * `load_imagen_from_checkpoint(checkpoint_path, load_weights=True, load_ema_if_available=False)`   utils.py:15-61
  * `ImagenConfig / ElucidatedImagenConfig(**params).create()`                                        configs.py:65-160
  * `ImagenTrainer.load(path, only_model=True, strict=...)` incl. its shape-tolerant fallback         tr.py:743-765, 209-220
`ema_pytorch` is a third-party dependency that is absent from /root/reference (setup.py:34 asks for `ema-pytorch>=0.0.3`);
only the key layout of its `EMA` module is needed and is restated above: the exponential-moving-average copy of unet i
lives under `{i}.ema_model.`.  Nothing here touches the GPU; everything is covered by CPU tests (tests/test_checkpoint.py).
"""
from __future__ import annotations

from pathlib import Path
from typing import Dict, Optional, Sequence

import torch

from .elucidated import ElucidatedImagen
from .imagen import DEFAULT_T5_NAME, T5_DIMS, Imagen
from .unet import NullUnet, Unet
from .unet3d import Unet3D

# configs.py:42-49 — what a `UnetConfig` fills in when the key is missing.  NOTE attn_dim_head / attn_heads: the config
# defaults (32 / 16) differ from the Unet constructor's (64 / 8, ip.py:1127-1128).
_UNET_DEFAULTS = dict(cond_dim=None, channels=3, attn_dim_head=32, attn_heads=16)
# configs.py:65-75
_IMAGEN_DEFAULTS = dict(timesteps=1000, noise_schedules='cosine', text_encoder_name=DEFAULT_T5_NAME, channels=3, loss_type='l2',
                        cond_drop_prob=0.5)
# configs.py:109-127
_ELUCIDATED_DEFAULTS = dict(text_encoder_name=DEFAULT_T5_NAME, channels=3, cond_drop_prob=0.5, num_sample_steps=32, sigma_min=0.002,
                            sigma_max=80, sigma_data=0.5, rho=7, P_mean=-1.2, P_std=1.2, S_churn=80, S_tmin=0.05, S_tmax=50, S_noise=1.003)


def _default_text_embed_dim() -> int:
    return T5_DIMS[DEFAULT_T5_NAME]   # configs.py:44 (`get_encoded_dim(DEFAULT_T5_NAME)`)


def _make_unet(params: dict, video: bool = False):
    if 'is_null' in params:                                   # NullUnetConfig, configs.py:36-40
        return NullUnet()
    for required in ('dim', 'dim_mults'):
        if required not in params:
            raise ValueError(f'unet config needs `{required}` (configs.py:42-44)')
    kw = {'text_embed_dim': _default_text_embed_dim(), **_UNET_DEFAULTS, **params}
    kw['dim_mults'] = tuple(kw['dim_mults'])
    return (Unet3D if video else Unet)(**kw)                   # configs.py:87-93: `video` selects Unet3D for every non-null unet


def imagen_from_config(imagen_type: str, imagen_params: dict):
    """`ImagenConfig(**imagen_params).create()` / `ElucidatedImagenConfig(...).create()` (configs.py:77-107, 129-160) on plain dicts.
    The returned model carries `_config` like the reference's, so it can be re-saved in the same format."""
    if imagen_type == 'original':
        klass, defaults = Imagen, _IMAGEN_DEFAULTS
    elif imagen_type == 'elucidated':
        klass, defaults = ElucidatedImagen, _ELUCIDATED_DEFAULTS
    else:
        raise ValueError(f'unknown imagen type {imagen_type} - you need to instantiate your Imagen with configurations, '
                         'using classes ImagenConfig or ElucidatedImagenConfig')       # utils.py:33-34
    for required in ('unets', 'image_sizes'):
        if required not in imagen_params:
            raise ValueError(f'imagen config needs `{required}`')
    params = {**defaults, **imagen_params}
    unet_params = list(params.pop('unets'))
    video = bool(params.pop('video', False))
    if len(params['image_sizes']) != len(unet_params):         # configs.py:77-81
        raise ValueError(f"image sizes length {len(params['image_sizes'])} must be equivalent to the number of unets {len(unet_params)}")
    params['image_sizes'] = tuple(params['image_sizes'])
    model = klass([_make_unet(dict(u), video) for u in unet_params], **params)
    model._config = {**params, 'unets': [dict(u) for u in unet_params], 'video': video}
    return model


class _Config:
    """Thin stand-ins for the reference's pydantic config classes (configs.py): `Config(**kwargs).create()`."""
    imagen_type = 'original'

    def __init__(self, **kwargs):
        self.kwargs = kwargs

    def dict(self):
        return dict(self.kwargs)

    def create(self):
        return imagen_from_config(self.imagen_type, self.kwargs)


class ImagenConfig(_Config):
    """configs.py:65-107."""


class ElucidatedImagenConfig(_Config):
    """configs.py:109-160."""
    imagen_type = 'elucidated'


class UnetConfig(_Config):
    """configs.py:42-52."""

    def create(self):
        return _make_unet(dict(self.kwargs))


class Unet3DConfig(_Config):
    """configs.py:54-63."""

    def create(self):
        return _make_unet(dict(self.kwargs), video=True)


class NullUnetConfig(_Config):
    """configs.py:36-40."""

    def create(self):
        return NullUnet()


def restore_parts(state_dict_target: Dict[str, torch.Tensor], state_dict_from: Dict[str, torch.Tensor]):
    """tr.py:209-220 — copy every tensor whose name and shape match, report the others."""
    for name, value in state_dict_from.items():
        if name not in state_dict_target:
            continue
        if value.size() == state_dict_target[name].size():
            state_dict_target[name].copy_(value)
        else:
            print(f"layer {name}({value.size()} different than target: {state_dict_target[name].size()}")
    return state_dict_target


def ema_unet_state_dicts(ema_state: Dict[str, torch.Tensor], num_unets: int) -> Sequence[Dict[str, torch.Tensor]]:
    """Split the trainer's `ema` entry into one unet state_dict per stage (the `{i}.ema_model.` sub-trees)."""
    out = []
    for i in range(num_unets):
        prefix = f'{i}.ema_model.'
        out.append({k[len(prefix):]: v for k, v in ema_state.items() if k.startswith(prefix)})
    return out


def _load_model_weights(imagen, loaded: dict, strict: bool):
    try:
        imagen.load_state_dict(loaded['model'], strict=strict)
    except RuntimeError:                                       # tr.py:760-765
        print("Failed loading state dict. Trying partial load")
        imagen.load_state_dict(restore_parts(imagen.state_dict(), loaded['model']))


def _load_ema_weights(imagen, loaded: dict, strict: bool = True):
    per_unet = ema_unet_state_dicts(loaded['ema'], len(imagen.unets))
    for unet, sd in zip(imagen.unets, per_unet):
        if isinstance(unet, NullUnet):
            continue
        if not sd:
            raise KeyError('the checkpoint has an `ema` entry but no `<i>.ema_model.*` weights for every unet')
        unet.load_state_dict(sd, strict=strict)               # utils.py:57-58


def _load_checkpoint_file(path, trust_checkpoint: bool):
    """torch.load restricted to tensors / plain containers (the trainer's dict holds nothing else); a checkpoint that needs the
    full unpickler (arbitrary code execution) is only read with an explicit `trust_checkpoint=True`."""
    import pickle
    try:
        return torch.load(str(path), map_location='cpu', weights_only=True)
    except (pickle.UnpicklingError, RuntimeError) as e:
        if not trust_checkpoint:
            raise RuntimeError(f'{path} holds objects beyond tensors and plain containers; pass trust_checkpoint=True to unpickle it '
                               f'(this executes code stored in the file): {e}') from e
        return torch.load(str(path), map_location='cpu', weights_only=False)


def load_imagen_from_checkpoint(checkpoint_path, load_weights: bool = True, load_ema_if_available: bool = False, *,
                                trust_checkpoint: bool = False):
    """utils.py:15-61.  Returns the model on CPU; move it with `.to('cuda')` before `.sample()`."""
    model_path = Path(checkpoint_path)
    assert model_path.exists(), f'checkpoint not found at {str(model_path.resolve())}'
    loaded = _load_checkpoint_file(model_path, trust_checkpoint)
    imagen_params, imagen_type = loaded.get('imagen_params'), loaded.get('imagen_type')
    if imagen_type not in ('original', 'elucidated'):
        raise ValueError(f'unknown imagen type {imagen_type} - you need to instantiate your Imagen with configurations, '
                         'using classes ImagenConfig or ElucidatedImagenConfig')
    assert imagen_params is not None, 'imagen type and configuration not saved in this checkpoint'
    imagen = imagen_from_config(imagen_type, imagen_params)
    if not load_weights:
        return imagen
    imagen.load_state_dict(loaded['model'])
    if not ('ema' in loaded and load_ema_if_available):
        print('loading non-EMA version of unets')
        return imagen
    _load_ema_weights(imagen, loaded)
    print('loaded EMA version of unets')
    return imagen


def load_trainer_checkpoint(imagen, path, *, use_ema: bool = False, strict: bool = True, noop_if_not_exist: bool = False,
                            trust_checkpoint: bool = False) -> Optional[dict]:
    """`ImagenTrainer.load(path, only_model=True, strict=strict)` (tr.py:743-768) for an already constructed model — the case of
    checkpoints saved without a config (no `imagen_params`): the caller builds `Imagen(...)` with the training-time kwargs and
    this loads `model` (and, with use_ema, overwrites every unet with its EMA copy, as the trainer samples with, tr.py:949-959).
    Returns the loaded dict (steps, version, ... stay available to the caller)."""
    path = Path(path)
    if noop_if_not_exist and not path.exists():
        print(f'trainer checkpoint not found at {str(path)}')
        return None
    assert path.exists(), f'{path} does not exist'
    loaded = _load_checkpoint_file(path, trust_checkpoint)
    _load_model_weights(imagen, loaded, strict)
    if use_ema:
        assert 'ema' in loaded
        _load_ema_weights(imagen, loaded, strict)
    for unet in imagen.unets:                                  # packed weight copies of a previous load are stale now
        if hasattr(unet, 'release_engines'):
            unet.release_engines()
    return loaded


def save_checkpoint(imagen, path, *, ema_unets: Optional[Sequence[torch.nn.Module]] = None, version: str = '2.0.0', **extra):
    """Write the model in the trainer's format (tr.py:696-741 with without_optim_and_sched=True) so the reference can read it back:
    `model`, `version`, `steps`, optional `ema`, and `imagen_type` / `imagen_params` when the model was built from a config."""
    obj = dict(model=imagen.state_dict(), version=version, steps=torch.zeros(len(imagen.unets)), **extra)
    if ema_unets is not None:
        ema = {}
        for i, (unet, avg) in enumerate(zip(imagen.unets, ema_unets)):
            ema.update({f'{i}.online_model.{k}': v for k, v in unet.state_dict().items()})
            ema.update({f'{i}.ema_model.{k}': v for k, v in avg.state_dict().items()})
            ema[f'{i}.initted'] = torch.tensor([True])
            ema[f'{i}.step'] = torch.tensor([0])
        obj['ema'] = ema
    if hasattr(imagen, '_config'):
        obj['imagen_type'] = 'elucidated' if isinstance(imagen, ElucidatedImagen) else 'original'
        obj['imagen_params'] = imagen._config
    path = Path(path)
    path.parent.mkdir(exist_ok=True, parents=True)
    torch.save(obj, str(path))
