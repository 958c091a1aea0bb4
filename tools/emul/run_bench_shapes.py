#!/usr/bin/env python
"""Every DISTINCT conv / GEMM launch of the benchmark's denoiser plans (tests/test_bench_shapes_gpu.py: bench_descriptors) executed on the
CPU through an emulated kernel library, at a reduced spatial size — same tile configuration, tile shape, channel counts, kernel size,
stride, fused prologue / epilogue modes; the map shrunk to a few tiles (partial ones included), one batch row.

Two steps, because the tile configurations are the PRODUCT planner's choice and cfg ids depend on which families a library holds:
    python tools/emul/run_bench_shapes.py --enumerate /tmp/desc.json                       # product library: dry-run of the planners
    IMAGEN_LIB_PATH=.../libimagen_emul[_x].so python tools/emul/run_bench_shapes.py --run /tmp/desc.json --out /tmp/x.pt
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import torch  # noqa: E402


def enumerate_descriptors(path):
    from imagen_pytorch_amd import ops
    from test_bench_shapes_gpu import bench_descriptors

    tab = ops.cfg_table()
    lib = ops.load_library()
    out = []
    for d, label, count in bench_descriptors():
        cid, th, tw = d["cfg"]
        tp, bn, g, fam = tab[cid]
        if fam == 1:
            continue                                   # conv_lds.hip is not emulated (and off by default in the planner)
        d = dict(d, family=fam, tile=(tp, bn), ring=lib.imagen_igemm_config_ring(cid), G=g, label=label, launches=count)
        out.append(d)
    json.dump(out, open(path, "w"), indent=0, default=str)
    print(f"{len(out)} distinct launches of families 0 / 2 / 3 -> {path}")


def shrink(d):
    """A few tiles per dimension, one of them partial; 1 x N token maps keep their single row."""
    _, th, tw = d["cfg"]
    s = d["stride"]
    if d["H"] == 1:
        return 1, min(d["W"], 2 * tw * s + tw * s // 2)
    return min(d["H"], (th + th // 2) * s + (d["K"] - 1) * (s > 1)), min(d["W"], (2 * tw + tw // 2) * s + (d["K"] - 1) * (s > 1))


def run(path, out):
    assert "emul" in os.path.basename(os.environ.get("IMAGEN_LIB_PATH", "")), "start with IMAGEN_LIB_PATH=<an emulated library>"
    torch.cuda.synchronize = lambda *a, **k: None
    from imagen_pytorch_amd import ops
    ops.current_stream_handle = lambda: 0
    from igemm_case import run_case

    lib = ops.load_library()
    tab = ops.cfg_table()
    by_key = {}
    for i, (tp, bn, g, fam) in enumerate(tab):
        by_key.setdefault((fam, tp, bn, lib.imagen_igemm_config_ring(i), g), i)
    results = {}
    for n, d in enumerate(json.load(open(path))):
        key = (d["family"], d["tile"][0], d["tile"][1], d["ring"], d["G"])
        cid = by_key[key]
        H, W = shrink(d)
        kw = {k: v for k, v in d.items() if k in ("C1", "C2", "Cout", "K", "stride", "pad", "prologue", "affine", "act_in", "act_out", "epilogue",
                                                   "ssq_out", "bias")}
        if d["prologue"] == "none" and d["has_pa"]:
            kw["prologue"] = "rs"
        captured = {}
        real_to_nchw, real_igemm = ops.act_to_nchw, ops.igemm

        def grab(a, _c=captured, _f=real_to_nchw):
            o = _f(a)
            _c["y"] = o.clone()
            return o

        def grab_y(plan, x1, pw, y, *a, _c=captured, **k):
            if isinstance(y, torch.Tensor):
                _c["y_t"] = y
            return real_igemm(plan, x1, pw, y, *a, **k)

        ops.act_to_nchw, ops.igemm = grab, grab_y
        try:
            r = run_case(ops, torch.device("cpu"), B=1, H=H, W=W, G=d["G"], cfg=(cid, d["cfg"][1], d["cfg"][2]), **kw)
        finally:
            ops.act_to_nchw, ops.igemm = real_to_nchw, real_igemm
        r["y"] = captured["y"] if "y" in captured else captured["y_t"].clone()
        r["desc"] = f"{d['label']}: {d['C1']}+{d['C2']}->{d['Cout']} k{d['K']} s{d['stride']} fam{d['family']} tile{tuple(d['tile'])} t{d['cfg'][1]}x{d['cfg'][2]} " \
                    f"{kw['prologue']}/{d['epilogue']}{'/ssq' if d['ssq_out'] else ''} @{H}x{W} (x{d['launches']})"
        results[n] = r
        print(f"{n:3d} err {r['err']:.2e}  {r['desc']}", flush=True)
    torch.save(results, out)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--enumerate")
    ap.add_argument("--run")
    ap.add_argument("--out")
    a = ap.parse_args()
    if a.enumerate:
        enumerate_descriptors(a.enumerate)
    else:
        run(a.run, a.out)
