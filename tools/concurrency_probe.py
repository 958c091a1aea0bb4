#!/usr/bin/env python
"""Does the chip run two independent sampling streams concurrently to an advantage?  Two Imagen instances sample 4 images each on their
own HIP streams from two host threads (every step is one hipGraph launch, so the host side is idle), against one instance sampling 8.
If the kernels' fixed costs (launch boundary, cold prologue, tail) dominate, the two streams hide each other's and the pair is faster.

    python tools/concurrency_probe.py [timesteps]
"""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

T = int(sys.argv[1]) if len(sys.argv) > 1 else 100
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
a, b = bench.build_imagen(T, dev), bench.build_imagen(T, dev)
te = torch.randn(8, 256, 768, device=dev)


def timed(fn, n=2):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def one8():
    a.sample(text_embeds=te, cond_scale=3.0, use_tqdm=False, seed=1)


def one4():
    a.sample(text_embeds=te[:4], cond_scale=3.0, use_tqdm=False, seed=1)


streams = [torch.cuda.Stream(dev), torch.cuda.Stream(dev)]


def pair():
    def work(m, s, x):
        with torch.cuda.stream(s):
            m.sample(text_embeds=x, cond_scale=3.0, use_tqdm=False, seed=1)
    th = [threading.Thread(target=work, args=(m, s, x)) for m, s, x in ((a, streams[0], te[:4]), (b, streams[1], te[4:]))]
    for t in th:
        t.start()
    for t in th:
        t.join()


print(f"T={T}: one instance, batch 8: {timed(one8):.1f} ms | one instance, batch 4: {timed(one4):.1f} ms | two instances x batch 4, two streams: {timed(pair):.1f} ms")
