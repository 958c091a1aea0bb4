"""INTEGRATION.md §2 is executable documentation: the ctypes binding snippet a reference maintainer would copy is extracted from the markdown
and run — on CPU the binding, the struct mirror and its size against the library (no launch); under `-m gpu` the launch itself against the
fp32 restatement of Block.forward.  A snippet that drifts from include/imagen_hip.h fails here, not in a maintainer's hands."""
import ctypes
import math
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _snippet():
    md = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    sec = md[md.index("## 2. Binding the C ABI directly"):]
    m = re.search(r"```python\n(.*?)```", sec, flags=re.S)
    assert m, "INTEGRATION.md §2 lost its python snippet"
    return m.group(1)


def test_integration_snippet_binds_and_mirrors_the_header():
    ns = {}
    exec(compile(_snippet(), "INTEGRATION.md#2", "exec"), ns)   # noqa: S102 — our own documentation
    from imagen_pytorch_amd import _abi
    st = ns["ImagenIgemmParams"]
    assert ns["lib"].imagen_sizeof(ns["IMAGEN_OP_IGEMM"]) == ctypes.sizeof(st)
    hdr = open(os.path.join(ROOT, "include", "imagen_hip.h")).read()
    body = re.search(r"typedef struct ImagenIgemmParams \{(.*?)\} ImagenIgemmParams;", hdr, flags=re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    decls = [" ".join(d.split()) for d in body.split(";") if d.strip()]
    names = [n.strip() for d in decls for n in (d.split("*")[-1] if "*" in d else d.split(" ", 1)[1]).split(",")]
    assert [f[0] for f in st._fields_] == names, "the mirror must list the header's fields in the header's order"
    # every symbol the document lists is exported, and the document lists every exported symbol
    md = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    tail = md[md.index("All entry points"):]
    for sym in _abi.EXPORTED_SYMBOLS:
        stem = sym if sym in tail else re.sub(r"_(begin|end|launch|destroy|create|record|elapsed_ms)$", "", sym)
        assert stem in tail, f"INTEGRATION.md does not list {sym}"
    assert "ImagenTrainer` and the `imagen` command line are NOT provided" in md and "imagen_pytorch_amd.cli" not in md


@pytest.mark.gpu
def test_integration_snippet_launches_a_block():
    from conftest import gpu_device
    from imagen_pytorch_amd import ops
    dev = gpu_device()
    ns = {}
    exec(compile(_snippet(), "INTEGRATION.md#2", "exec"), ns)   # noqa: S102
    g = torch.Generator().manual_seed(0)
    B, C, H, W, Cout = 2, 64, 24, 40, 64
    x = (torch.randn(B, C, H, W, generator=g)).half().float()
    w, b = torch.randn(Cout, C, 3, 3, generator=g) / math.sqrt(9 * C), torch.randn(Cout, generator=g) * 0.1
    pa, ps = 1 + 0.2 * torch.randn(B, C, generator=g), 0.2 * torch.randn(B, C, generator=g)
    rs = 1.0 / x.norm(dim=1).clamp(min=1e-12)
    ref = torch.nn.functional.conv2d(torch.nn.functional.silu(x * rs[:, None] * pa[:, :, None, None] + ps[:, :, None, None]), w, b, padding=1)
    xa = ops.act_from_nchw(x.to(dev))
    y = ops.new_act(B, H, W, Cout, dev)
    packed = ops.pack_weight(w, b, dev)
    ns["block_forward"](xa.t, rs.reshape(-1).contiguous().to(dev), pa.contiguous().to(dev), ps.contiguous().to(dev), packed, y.t,
                        stream=ops.current_stream_handle())
    torch.cuda.synchronize()
    got = ops.act_to_nchw(y).cpu()
    assert ((got - ref).norm() / ref.norm()).item() < 1e-3
