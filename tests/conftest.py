import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "reference: needs /root/reference (build container only)")


def pytest_collection_modifyitems(config, items):
    import torch

    have_gpu = torch.cuda.is_available()
    skip_gpu = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords and not have_gpu:
            item.add_marker(skip_gpu)


@pytest.fixture(scope="session")
def lib():
    from imagen_pytorch_amd import _abi

    return _abi.load_library()
