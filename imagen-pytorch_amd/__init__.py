"""imagen-pytorch_amd — MI355X-native (gfx950) Imagen cascaded-DDPM sampling path.

Drop-in `Unet(...)` / `Imagen(...)` constructors, `state_dict` layout and
`.forward()` / `.sample()` signatures of lucidrains/imagen-pytorch; behind them the
host code drives hand-written HIP kernels through the C ABI in include/imagen_hip.h.
Importable as `imagen_pytorch_amd` (the hyphenated directory is the real package).
"""
from . import _abi  # noqa: F401
from ._abi import ImagenHipError, load_library  # noqa: F401

__all__ = ["ImagenHipError", "load_library"]

try:  # the model classes need the ops layer; keep the ABI importable on its own for the symbol tests
    from .unet import Unet, NullUnet, BaseUnet64, SRUnet256, SRUnet1024  # noqa: F401
    from .unet3d import Unet3D  # noqa: F401
    from .imagen import Conditioning, Imagen  # noqa: F401
    from .elucidated import ElucidatedImagen  # noqa: F401
    from .schedules import GaussianDiffusionContinuousTimes  # noqa: F401
    from .checkpoint import (ElucidatedImagenConfig, ImagenConfig, NullUnetConfig, Unet3DConfig, UnetConfig,  # noqa: F401
                             load_imagen_from_checkpoint, load_trainer_checkpoint, save_checkpoint)
    __all__ += ["Unet", "Unet3D", "NullUnet", "BaseUnet64", "SRUnet256", "SRUnet1024", "Imagen", "Conditioning", "ElucidatedImagen", "GaussianDiffusionContinuousTimes",
                "load_imagen_from_checkpoint", "load_trainer_checkpoint", "save_checkpoint", "ImagenConfig", "ElucidatedImagenConfig", "UnetConfig",
                "Unet3DConfig", "NullUnetConfig"]
except ModuleNotFoundError as _e:  # pragma: no cover - only while the package is being bootstrapped
    if _e.name not in ("imagen_pytorch_amd.unet", "imagen_pytorch_amd.imagen", "imagen_pytorch_amd.schedules"):
        raise
