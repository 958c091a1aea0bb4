import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "reference: needs /root/reference (build container only)")


# IMAGEN_EMUL_TESTS=1 (with IMAGEN_LIB_PATH = an emulated kernel library, tools/emul): the `-m gpu` tests run on the CPU through the
# functional emulation of the kernel library (every csrc/*.hip but conv_lds.hip, graph capture included) instead of being skipped for
# want of a GPU.  `gpu_device()` is what those tests allocate on: cuda:0, or cpu under emulation.
EMULATED = os.environ.get("IMAGEN_EMUL_TESTS") == "1"


def gpu_device():
    import torch
    return torch.device("cpu" if EMULATED else "cuda:0")


def pytest_collection_modifyitems(config, items):
    import torch

    have_gpu = torch.cuda.is_available()
    skip_gpu = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords and not have_gpu:
            if EMULATED:
                continue
            item.add_marker(skip_gpu)


if EMULATED:
    @pytest.fixture(scope="session", autouse=True)
    def _emulated_backend():
        import torch
        assert "emul" in os.path.basename(os.environ.get("IMAGEN_LIB_PATH", "")), "IMAGEN_EMUL_TESTS=1 needs IMAGEN_LIB_PATH=<emulated library>"
        import contextlib
        from imagen_pytorch_amd import imagen as imagen_mod, ops, unet as unet_mod

        class Stream:
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

        class Event:
            def __init__(self, *a, **k):
                pass

            def record(self, stream=None):
                pass

            def synchronize(self):
                pass

        torch.cuda.synchronize = lambda *a, **k: None
        torch.cuda.Stream, torch.cuda.Event = Stream, Event
        torch.cuda.current_stream = lambda *a, **k: Stream()
        torch.cuda.device = lambda *a, **k: contextlib.nullcontext()
        torch.cuda.stream = lambda *a, **k: contextlib.nullcontext()
        ops.current_stream_handle = lambda: 0
        imagen_mod._SAMPLING_DEVICE_TYPES = ("cuda", "cpu")
        unet_mod._ENGINE_DEVICE_TYPES = ("cuda", "cpu")
        yield


@pytest.fixture(scope="session")
def lib():
    from imagen_pytorch_amd import _abi

    return _abi.load_library()
