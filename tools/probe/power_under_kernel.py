#!/usr/bin/env python
"""Probe (GPU): package power and the clocks rocm-smi reports while ONE kernel shape loops for a few seconds — conv_big on the benchmark's shapes, the
bare MFMA calibration loop, a device copy.  Backs the reading of NOTES_r06 3a (the shader clock under conv_big is 1.61-1.71 GHz by the kernel's own
s_memtime / s_memrealtime stamps): is that the power limit?  Prints one JSON line per case."""
import ctypes
import json
import math
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa: E402

import bench  # noqa: E402
import conv_bench  # noqa: E402
from imagen_pytorch_amd import _abi, ops  # noqa: E402


def loop_for(fn, seconds, samples):
    stop = [False]

    def sampler():
        time.sleep(1.0)
        while not stop[0]:
            samples.append(bench.smi_sample(extra=True))
            time.sleep(0.5)
    th = threading.Thread(target=sampler)
    th.start()
    t0, n = time.perf_counter(), 0
    while time.perf_counter() - t0 < seconds:
        for _ in range(50):
            fn()
        torch.cuda.synchronize()
        n += 50
    dt = time.perf_counter() - t0
    stop[0] = True
    th.join()
    return dt / n * 1e6


def summarise(samples):
    def num(v):
        try:
            return float(str(v).lower().replace("mhz", "").replace("w", "").strip())
        except ValueError:
            return None
    out = {}
    for s in samples:
        for k, v in s.items():
            x = num(v)
            if x is not None and ("power" in k.lower() or k in ("sclk", "fclk", "socclk", "mclk")):
                out.setdefault(k, []).append(x)
    return {k: dict(min=min(v), max=max(v), mean=round(sum(v) / len(v), 1), n=len(v)) for k, v in out.items()}


def main():
    dev = torch.device("cuda:0")
    lib = _abi.load_library()
    g = torch.Generator().manual_seed(0)
    B = 16
    for shp, spec in (("192:128:64", "big:3"), ("128:128:64", "big:3"), ("384:256:32", "big:2")):
        Cin, Cout, H = map(int, shp.split(":"))
        w = torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(9 * Cin)
        pw = ops.pack_weight(w, torch.randn(Cout, generator=g) * 0.1, dev, G=4)
        xs = [ops.act_from_nchw((torch.randn(B, Cin, H, H, generator=g) * 0.7).to(dev)) for _ in range(4)]
        ys = [ops.new_act(B, H, H, Cout, dev) for _ in range(4)]
        cfg = conv_bench.resolve(spec, Cout, H, B)
        plan = ops.Plan("loop")
        for i in range(4):
            ops.igemm(plan, xs[i], pw, ys[i], cfg=cfg, label=spec)
        plan.run()
        torch.cuda.synchronize()
        samples = []
        us = loop_for(plan.run, 6.0, samples) / 4
        fl = 2.0 * B * H * H * Cout * 9 * Cin
        print(json.dumps(dict(case=f"conv_big {shp}", us_per_launch=round(us, 2), tflops=round(fl / us / 1e6, 1), smi=summarise(samples))), flush=True)
    v = ctypes.c_float()
    s = ops.current_stream_handle()
    samples = []
    sink = torch.empty(1 << 20, device=dev)
    loop_for(lambda: _abi.check(lib.imagen_probe_mfma(20000, 1, sink.data_ptr(), s, ctypes.byref(v)), "mfma"), 6.0, samples)
    print(json.dumps(dict(case="bare MFMA loop (calibration probe)", tflops=round(v.value, 1), smi=summarise(samples))), flush=True)
    n = 1 << 30
    src, dst = torch.empty(n, dtype=torch.uint8, device=dev).fill_(1), torch.empty(n, dtype=torch.uint8, device=dev)
    samples = []
    loop_for(lambda: _abi.check(lib.imagen_probe_copy(dst.data_ptr(), src.data_ptr(), n, 1, s, ctypes.byref(v)), "copy"), 6.0, samples)
    print(json.dumps(dict(case="device copy 1 GiB (calibration probe)", gbs=round(v.value, 1), smi=summarise(samples))), flush=True)


if __name__ == "__main__":
    main()
