"""Sampling half of the reference's `ImagenTrainer` (imagen_pytorch/trainer.py = "tr.py"): the caller on the user's side of the hot
path.  Scripts written against the reference sample through the trainer — `trainer.load(path)`, `trainer.sample(texts=..., batch_size=...,
max_batch_size=..., use_non_ema=...)` (tr.py:743-809, 947-961, 188-209) — so this class keeps exactly that surface over the MI355X
sampler: checkpoint layout of tr.py:677-741 (`model`, `ema`, `version`, `steps`), sampling from the EMA unets unless `use_non_ema`,
batch chunking by `max_batch_size` with the reference's argument-splitting rules.  Everything that trains (`forward`, `update`,
`train_step`, dataloaders, optimizers, accelerate) is outside this build and raises.

EMA weights: the reference keeps an `ema_pytorch.EMA` copy of every unet and swaps it in for sampling (tr.py:862-899).  Here both
weight sets of a loaded checkpoint are kept as state dicts on the host and the one a `sample()` call asks for is made the active one
of `imagen` (a reload + repack of the device weights only when the selection changes — sampling scripts stay on the EMA set).
"""
from __future__ import annotations

from collections.abc import Iterable
from math import ceil
from pathlib import Path
from typing import Optional

import torch

from .checkpoint import _load_checkpoint_file, _load_model_weights, ema_unet_state_dicts
from .elucidated import ElucidatedImagen
from .imagen import Imagen, _out_of_scope
from .unet import NullUnet


def _num_to_groups(num, divisor):
    """tr.py:78-84."""
    groups, remainder = divmod(num, divisor)
    return [divisor] * groups + ([remainder] if remainder > 0 else [])


def _split(t, split_size):
    """tr.py:138-155: tensors along dim 0, other iterables (lists of texts) by slicing."""
    if isinstance(t, torch.Tensor):
        return t.split(split_size, dim=0)
    return [t[i * split_size:(i + 1) * split_size] for i in range(ceil(len(t) / split_size))]


def split_args_and_kwargs(*args, split_size=None, **kwargs):
    """tr.py:163-186: chunk every tensor / iterable argument along the batch, repeat everything else; yields
    (chunk fraction, (args, kwargs)) like the reference."""
    all_args = (*args, *kwargs.values())
    first_tensor = next((t for t in all_args if isinstance(t, torch.Tensor)), None)
    assert first_tensor is not None
    batch_size = len(first_tensor)
    split_size = split_size if split_size is not None else batch_size
    num_chunks = ceil(batch_size / split_size)
    n_kw = len(kwargs)
    keys = list(kwargs.keys())
    n_pos = len(all_args) - n_kw
    split_all = [_split(a, split_size) if a is not None and isinstance(a, (torch.Tensor, Iterable)) and not isinstance(a, str)
                 else ((a,) * num_chunks) for a in all_args]
    chunk_sizes = tuple(len(c) for c in _split(first_tensor, split_size))
    for size, chunked in zip(chunk_sizes, zip(*split_all)):
        yield size / batch_size, (chunked[:n_pos], dict(zip(keys, chunked[n_pos:])))


class ImagenTrainer:
    """`ImagenTrainer(imagen)` or `ImagenTrainer(imagen_checkpoint_path=...)` (tr.py:229-246); training-only keyword arguments of the
    reference (lr, eps, warmup / cosine schedules, fp16, dataloader options, ...) are accepted and ignored."""

    def __init__(self, imagen=None, imagen_checkpoint_path=None, use_ema: bool = True, checkpoint_path=None, device=None, **ignored_training_kwargs):
        assert (imagen is not None) ^ (imagen_checkpoint_path is not None), \
            'either imagen instance is passed into the trainer, or a checkpoint path that contains the imagen config'
        if imagen is None:
            from .checkpoint import load_imagen_from_checkpoint
            imagen = load_imagen_from_checkpoint(imagen_checkpoint_path)
        assert isinstance(imagen, (Imagen, ElucidatedImagen))
        self.imagen = imagen
        self.is_elucidated = isinstance(imagen, ElucidatedImagen)
        self.num_unets = len(imagen.unets)
        self.use_ema = use_ema
        self._online_sd = None          # set by load(): the two weight sets of the checkpoint, host state dicts
        self._ema_sds = None
        self._active = 'online'
        self.steps = torch.zeros(self.num_unets, dtype=torch.long)
        self.is_main = True
        self._device = torch.device(device) if device is not None else None
        self.checkpoint_path = checkpoint_path

    @property
    def device(self):
        return self._device if self._device is not None else self.imagen.device

    @property
    def unets(self):
        return self.imagen.unets

    def to(self, device):
        self._device = torch.device(device)
        self.imagen.to(self._device)
        return self

    def cuda(self):
        return self.to('cuda')

    def _release(self):
        for unet in self.imagen.unets:                       # packed copies of the previous weights are stale now
            if hasattr(unet, 'release_engines'):
                unet.release_engines()

    def _activate(self, which: str):
        if which == self._active:
            return
        if which == 'ema':
            from .unet import NullUnet
            for unet, sd in zip(self.imagen.unets, self._ema_sds):
                if not isinstance(unet, NullUnet):
                    unet.load_state_dict(sd)
        else:
            self.imagen.load_state_dict(self._online_sd)
        self._release()
        self._active = which

    # ---- checkpoints (tr.py:677-809) ------------------------------------------------------------------------------------------
    def load(self, path, only_model: bool = False, strict: bool = True, noop_if_not_exist: bool = False, trust_checkpoint: bool = False):
        path = Path(path)
        if noop_if_not_exist and not path.exists():
            print(f'trainer checkpoint not found at {str(path)}')
            return None
        assert path.exists(), f'{path} does not exist'
        loaded = _load_checkpoint_file(path, trust_checkpoint)
        _load_model_weights(self.imagen, loaded, strict)
        self._release()
        self._active = 'online'
        self._online_sd = {k: v.detach().cpu().clone() for k, v in self.imagen.state_dict().items()}
        self._ema_sds = None
        if only_model:                                       # tr.py:767-768: neither the step counters nor the EMA unets
            return loaded
        if 'steps' in loaded:
            self.steps.copy_(torch.as_tensor(loaded['steps']).long().reshape(-1)[: self.num_unets])
        if self.use_ema:                                     # tr.py:790-800
            assert 'ema' in loaded
            self._ema_sds = ema_unet_state_dicts(loaded['ema'], self.num_unets)
            assert all(sd or isinstance(u, NullUnet) for u, sd in zip(self.imagen.unets, self._ema_sds)), \
                'the checkpoint has an `ema` entry but no `<i>.ema_model.*` weights for every unet'
        print(f'checkpoint loaded from {str(path)}')
        return loaded

    def save(self, path, overwrite: bool = True, **kwargs):
        """The online weights in the trainer's layout (no optimizer / scheduler / EMA state: nothing here updates them), tr.py:677-741."""
        from .checkpoint import save_checkpoint
        path = Path(path)
        assert overwrite or not path.exists()
        self._activate('online')
        save_checkpoint(self.imagen, path, **kwargs)

    # ---- sampling (tr.py:947-961, 188-209) ------------------------------------------------------------------------------------
    @torch.no_grad()
    def sample(self, *args, max_batch_size: Optional[int] = None, use_non_ema: bool = False, **kwargs):
        self._activate('online' if use_non_ema or self._ema_sds is None else 'ema')   # tr.py:951: sample from the EMA unets
        model = self.imagen
        kwargs.setdefault('device', self.device)
        run = lambda *a, **k: model.sample(*a, **k)
        if max_batch_size is None:
            return run(*args, **kwargs)
        if model.unconditional:
            outputs = [run(*args, **{**kwargs, 'batch_size': b}) for b in _num_to_groups(kwargs.get('batch_size'), max_batch_size)]
        else:
            outputs = [run(*a, **k) for _, (a, k) in split_args_and_kwargs(*args, split_size=max_batch_size, **kwargs)]
        if isinstance(outputs[0], torch.Tensor):
            return torch.cat(outputs, dim=0)
        return list(map(lambda t: torch.cat(t, dim=0), list(zip(*outputs))))   # return_all_unet_outputs: one cat per unet (tr.py:204-207)

    # ---- the training half is not part of this build ---------------------------------------------------------------------------
    def forward(self, *args, **kwargs):
        _out_of_scope("ImagenTrainer.forward (training, tr.py:963-989)")

    __call__ = forward

    def update(self, *args, **kwargs):
        _out_of_scope("ImagenTrainer.update (optimizer step, tr.py:902-945)")

    def train_step(self, *args, **kwargs):
        _out_of_scope("ImagenTrainer.train_step (tr.py:479-500)")

    def valid_step(self, *args, **kwargs):
        _out_of_scope("ImagenTrainer.valid_step (tr.py:502-520)")
