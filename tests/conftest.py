import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "reference: needs /root/reference (build container only)")


# IMAGEN_EMUL_TESTS=1 (with IMAGEN_LIB_PATH = an emulated kernel library, tools/emul): the per-launch IGEMM tests of the files below run
# on the CPU through the functional emulation of csrc/igemm.hip instead of being skipped for want of a GPU.
EMULATED = os.environ.get("IMAGEN_EMUL_TESTS") == "1"
EMULATABLE_FILES = ("test_igemm_cfgs_gpu.py",)


def pytest_collection_modifyitems(config, items):
    import torch

    have_gpu = torch.cuda.is_available()
    skip_gpu = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords and not have_gpu:
            if EMULATED and os.path.basename(str(item.fspath)) in EMULATABLE_FILES:
                continue
            item.add_marker(skip_gpu)


if EMULATED:
    @pytest.fixture(scope="session", autouse=True)
    def _emulated_backend():
        import torch
        assert "emul" in os.path.basename(os.environ.get("IMAGEN_LIB_PATH", "")), "IMAGEN_EMUL_TESTS=1 needs IMAGEN_LIB_PATH=<emulated library>"
        from imagen_pytorch_amd import ops
        torch.cuda.synchronize = lambda *a, **k: None
        ops.current_stream_handle = lambda: 0
        yield


@pytest.fixture(scope="session")
def lib():
    from imagen_pytorch_amd import _abi

    return _abi.load_library()
