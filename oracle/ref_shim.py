"""In-process import shim for the *reference* implementation (TEST INFRASTRUCTURE ONLY).

This file is part of the oracle tooling: it is only ever used inside the build
container (where /root/reference exists) to (a) validate oracle/unet_oracle.py and
(b) generate the golden fixtures under tests/golden/.  Nothing in the product
package, `-m gpu` tests, smoke() or bench.py imports it at run time on the GPU box.

Recipe: the reference cannot be imported as a plain package here
because beartype / torchvision / kornia / ema_pytorch / pytorch_warmup are absent
and `T5Config.from_pretrained` needs the network.  None of those touch hot-path
numerics, so they are stubbed in-process; no reference source is copied.
"""
import importlib
import os
import sys
import types
import typing

REFERENCE_ROOT = os.environ.get("IMAGEN_REFERENCE_ROOT", "/root/reference")


def reference_available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "imagen_pytorch"))


_cached = {}


def load_reference(module: str = "imagen_pytorch"):
    """Return the reference module `imagen_pytorch.<module>` (imagen_pytorch | elucidated_imagen | imagen_video | configs)."""
    if module in _cached:
        return _cached[module]
    if not reference_available():
        raise RuntimeError(f"reference tree not found at {REFERENCE_ROOT}")

    import transformers  # must come first: the reference evaluates T5 config at class-definition time

    class _FakeT5Config:
        d_model = 768

    transformers.T5Config.from_pretrained = staticmethod(lambda *a, **k: _FakeT5Config())

    def _stub(name, **attrs):
        if name in sys.modules:
            return sys.modules[name]
        m = types.ModuleType(name)
        for k, v in attrs.items():
            setattr(m, k, v)
        sys.modules[name] = m
        return m

    _stub("beartype", beartype=lambda f: f)
    _stub("beartype.typing", List=typing.List, Union=typing.Union, Optional=typing.Optional,
          Tuple=typing.Tuple, Dict=typing.Dict, Callable=typing.Callable, Any=typing.Any)
    tv = _stub("torchvision")
    tvt = _stub("torchvision.transforms", ToPILImage=lambda: (lambda x: x))
    tv.transforms = tvt
    ko = _stub("kornia")
    ka = _stub("kornia.augmentation", RandomCrop=None)
    ko.augmentation = ka

    if module in ("configs", "trainer", "utils"):
        # checkpoint-format fixtures only (oracle/make_golden.py --checkpoint): the trainer module's absent dependencies are
        # stubbed so `configs.ImagenConfig(...).create()` runs; no trainer / EMA object is ever constructed through these stubs
        import torch.nn as _nn

        class _AbsentEMA(_nn.Module):
            def __init__(self, *a, **k):
                raise RuntimeError("ema_pytorch is not installed in this image (stub)")

        _stub("ema_pytorch", EMA=_AbsentEMA)
        _stub("pytorch_warmup")
        _stub("imagen_pytorch.data", cycle=None)   # the data loaders pull in `datasets`, which rejects the torchvision stub

    if "imagen_pytorch" not in sys.modules or not hasattr(sys.modules["imagen_pytorch"], "__path__"):
        pkg = types.ModuleType("imagen_pytorch")
        pkg.__path__ = [os.path.join(REFERENCE_ROOT, "imagen_pytorch")]
        sys.modules["imagen_pytorch"] = pkg

    mod = importlib.import_module("imagen_pytorch." + module)
    _cached[module] = mod
    return mod
