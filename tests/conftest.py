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


PARITY_LOG = []   # (test id, {name: measured error}) — filled by record_parity(), printed in the terminal summary so the driver's pytest.log carries the figures


def record_parity(test_id, **vals):
    """Keep measured parity figures of a passing test: printed at the end of the run (also with -q) and written to
    gpurun_out/parity_measured.json (copied to profiles/ by the round's measurement script)."""
    import json
    PARITY_LOG.append((test_id, vals))
    path = os.path.join(ROOT, "gpurun_out", "parity_measured.json")
    try:
        os.makedirs(os.path.dirname(path), exist_ok=True)
        try:
            rec = json.load(open(path))
        except (OSError, ValueError):
            rec = {}
        rec[test_id] = vals
        with open(path, "w") as f:
            json.dump(rec, f, indent=1, sort_keys=True, default=str)
    except OSError:
        pass


def pytest_terminal_summary(terminalreporter):
    if not PARITY_LOG:
        return
    terminalreporter.write_sep("=", "measured parity (normwise relative error vs the oracle / reference fixtures)")
    # whole-denoiser records (a cond AND a null figure) go LAST, so that the tail of the driver's pytest.log holds every one of them
    whole = lambda rec: isinstance(rec[1].get("cond"), float) and isinstance(rec[1].get("null"), float)
    for tid, vals in sorted(PARITY_LOG, key=whole):
        parts = []
        for k, v in vals.items():
            if isinstance(v, float):
                parts.append(f"{k}={v:.2e}")
            elif isinstance(v, dict):
                fl = [x for x in v.values() if isinstance(x, float)]
                if fl:
                    parts.append(f"max({k})={max(fl):.2e}")
            else:
                parts.append(f"{k}={v}")
        terminalreporter.write_line(f"{tid}: " + " ".join(parts))
    rows = [(tid, v) for tid, v in PARITY_LOG if whole((tid, v))]
    if rows:
        terminalreporter.write_sep("-", "whole denoiser forwards vs the fp32 oracle: cond / null (bar)")
        for tid, v in rows:
            terminalreporter.write_line(f"{tid}: cond {v['cond']:.3e} null {v['null']:.3e} (tol {v.get('tol', float('nan')):.2e})")
        terminalreporter.write_line(f"worst whole-denoiser figure: {max(max(v['cond'], v['null']) for _, v in rows):.3e}")


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
