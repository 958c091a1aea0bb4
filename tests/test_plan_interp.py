"""CPU: the Unet planner's launch list, executed by the plan interpreter (tests/plan_interp.py: every op kind restated from its
documented contract, fp16 storage), reproduces the reference fixture — i.e. buffer wiring, strides, folded weights and op order
of imagen_pytorch_amd/engine.py are right, independent of the HIP kernels (which the -m gpu tests check)."""
import os

import pytest
import torch

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture()
def reference_weights():
    from imagen_pytorch_amd import ops

    ops.KEEP_REFERENCE_WEIGHTS = True
    try:
        yield ops
    finally:
        ops.KEEP_REFERENCE_WEIGHTS = False
        ops.REFERENCE_WEIGHTS.clear()


def nerr(a, b):
    return float((a.float() - b.float()).norm() / b.float().norm())


@pytest.mark.parametrize("name", ["unet_tiny_base.pt", "unet_tiny_sr.pt"])
@pytest.mark.parametrize("cfg_rows", [False, True])
def test_unet_plan_on_cpu_matches_reference_fixture(name, cfg_rows, reference_weights):
    from imagen_pytorch_amd import Unet
    from imagen_pytorch_amd.engine import UnetEngine
    from plan_interp import Interpreter

    g = torch.load(os.path.join(GOLDEN, name), weights_only=False)
    u = Unet(**g["kwargs"]).eval()
    u.load_state_dict(g["state_dict"])
    B, S = g["x"].shape[0], g["x"].shape[-1]
    rows = 2 * B if cfg_rows else B
    eng = UnetEngine(u, rows, B, S, "cpu", dry=True)
    keep = torch.ones(rows, dtype=torch.bool)
    keep[B:] = False
    eng.set_conditioning(text_embeds=g["text_embeds"], text_mask=g["text_mask"], keep=keep, lowres_noise_times=g["extra"].get("lowres_noise_times"))
    it = Interpreter()
    for t in (eng.x_in, eng.lowres_in, eng.times, eng.lowres_times, eng.out, eng.keep_u8, eng.src_idx, eng.arange_idx, eng.t_const.t):
        it.mem.register(t)
    it.run(eng._static_plans[g["text_embeds"].shape[1]][0])
    eng.x_in.copy_(g["x"])
    if eng.lowres:
        eng.lowres_in.copy_(g["extra"]["lowres_cond_img"])
    eng.times.copy_(g["time"].repeat(rows // B))
    it.run(eng.step_plan)
    out = eng.out
    e_cond = nerr(out[:B], g["out_cond"])
    assert e_cond < 1e-2, e_cond          # fp16 storage of every activation, as on the GPU (UNET_TOL of tests/test_model_gpu.py)
    if cfg_rows:
        e_null = nerr(out[B:], g["out_null"])
        assert e_null < 1e-2, e_null
    assert len(it.trace) == len(eng.step_plan) + len(eng._static_plans[g["text_embeds"].shape[1]][0])
