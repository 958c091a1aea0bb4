#!/usr/bin/env python
"""Phase timeline of the GlobalContext gate derivation (csrc/elementwise.hip: gca_final_fast_body inside gca_final_fast_kernel and gca_tail_kernel) on
the benchmark's shapes, every launch behind a 256 MiB device copy (weights and partial rows as cold as inside the sampling loop):

    IMAGEN_LIB_PATH=<a -DGCA_TRACE library> python tools/gca_bench.py --trace

prints one JSON line: per case the mean s_memtime deltas (cycles of the 100 MHz-class constant clock are NOT assumed: the tool also times a known
delay to convert) between the stamps: 0 kernel entry, 1 body entry, 2 every load requested, 3 chunk maxima merged (first barrier: the loads have
landed), 4 weights of the chunks known, 5 pooled context ready, 6 first MLP layer done, 7 gate ready, 8 (tail) rows streamed."""
import argparse
import ctypes
import json
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

# (C, HW, chunks) at 16 rows: README unet2's 64^2 / 32^2 / 128^2 / 256^2 levels and unet1's 16^2 / 8^2
CASES = [(128, 4096, 16), (256, 1024, 8), (64, 16384, 64), (32, 65536, 256), (128, 256, 8), (64, 1024, 4)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--trace", action="store_true")
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--tag", default="")
    args = ap.parse_args()
    import torch
    from imagen_pytorch_amd import _abi, ops

    dev = torch.device("cuda:0")
    lib = _abi.load_library()
    assert hasattr(lib, "imagen_debug_gca_trace"), "not a -DGCA_TRACE library"
    lib.imagen_debug_gca_trace.argtypes = [ctypes.c_void_p]
    buf = torch.zeros(8192, 16, dtype=torch.int64, device=dev)
    assert lib.imagen_debug_gca_trace(buf.data_ptr()) == 0
    n = 256 << 20
    src, dst = torch.empty(n, dtype=torch.uint8, device=dev).fill_(1), torch.empty(n, dtype=torch.uint8, device=dev)
    v = ctypes.c_float()
    flush = lambda: _abi.check(lib.imagen_probe_copy(dst.data_ptr(), src.data_ptr(), n, 1, ops.current_stream_handle(), ctypes.byref(v)), "flush")
    B = 16
    out = {}
    g = torch.Generator().manual_seed(0)
    rn = lambda *s: torch.randn(*s, generator=g)
    for C, HW, chunks in CASES:
        hidden = max(3, C // 2)
        w1t, b1 = (rn(C, hidden) / math.sqrt(C)).to(dev), (rn(hidden) * 0.1).to(dev)
        w2t, b2 = (rn(hidden, C) / math.sqrt(hidden)).to(dev), (rn(C) * 0.1).to(dev)
        part = torch.rand(B, chunks, C + 2, generator=g).to(dev)
        gate = torch.empty(B, C, device=dev)
        S = int(math.isqrt(HW))
        h, x, o = ops.new_act(B, S, S, C, dev), ops.new_act(B, S, S, C, dev), ops.new_act(B, S, S, C, dev)
        h.t.normal_(); x.t.normal_()
        ssq = torch.empty(B * HW, device=dev)
        plans = {}
        pf = ops.Plan("final")
        ops.gca_final(pf, part, w1t, b1, w2t, b2, gate, B=B, C=C, chunks=chunks, label="gca")
        plans["final"] = pf
        pt = ops.Plan("tail")
        ops.gca_tail(pt, h, x, o, part=part, chunks=chunks, w1t=w1t, b1=b1, w2t=w2t, b2=b2, gate=gate, ssq_out=ssq, label="tail")
        plans["tail"] = pt
        pg = ops.Plan("tail_gate_in")
        ops.gca_tail(pg, h, x, o, gate_in=gate, ssq_out=ssq, label="tail")
        plans["tail_gate_in"] = pg
        for name, plan in plans.items():
            plan.run()
            torch.cuda.synchronize()
            acc, cnt = None, 0
            for _ in range(args.reps):
                flush()
                buf.zero_()
                plan.run()
                torch.cuda.synchronize()
                t = buf.cpu().double()
                t = t[t[:, 0] > 0]
                last = 8 if name.startswith("tail") else 7
                idx = [0, 1, 2, 3, 4, 5, 6, 7] + ([8] if last == 8 else [])
                if name == "tail_gate_in":
                    idx = [0, 8]
                d = torch.stack([t[:, b_] - t[:, a_] for a_, b_ in zip(idx[:-1], idx[1:])], 1).mean(0)
                row = torch.cat((d, torch.tensor([(t[:, last] - t[:, 0]).mean(), t[:, last].max() - t[:, 0].min(), float(t.shape[0])])))
                acc = row if acc is None else acc + row
                cnt += 1
            acc = (acc / cnt).tolist()
            out[f"C{C}_HW{HW}_chunks{chunks}:{name}"] = dict(phase_ticks=[round(a, 1) for a in acc[:-3]], per_wg_ticks=round(acc[-3], 1),
                                                            first_to_last_ticks=round(acc[-2], 1), wgs=round(acc[-1]))
    # s_memtime runs on a constant clock: convert with a timed idle kernel-free interval
    print(json.dumps(dict(tag=args.tag, note="ticks of s_memtime = shader clock cycles (MI355X_MICROARCH.md: per-instruction cycle constants)", trace=out)))


if __name__ == "__main__":
    main()
