#!/usr/bin/env python
"""Kernel-level A/B of the prologue-free 3x3 conv families on the benchmark's C >= 128 layer shapes (rows 16 = batch 8 under CFG):
each candidate tile configuration is launched `--iters` times over four rotating input / output buffer sets (same weights), timed with
events on the launch stream, and its output compared with the first candidate's (the planner's current pick).

    python tools/conv_bench.py [--iters 40] [--shapes 128:128:64 ...] [--out file.jsonl]

Candidates: "pick" (ops.pick_cfg as the planner calls it), "dma:<tile px>x<tile couts>", "big:<n>" (n-th configuration of family 5),
"cfg:<id>:<th>:<tw>".  One JSON line per (shape, candidate): us per launch, TFLOP/s, normwise distance to the first candidate."""
import argparse
import json
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from imagen_pytorch_amd import ops  # noqa: E402

DEFAULT_SHAPES = ["128:128:64", "192:128:64", "256:256:32", "384:256:32", "128:128:32"]
DEFAULT_CANDS = {64: ["pick", "big:0", "big:3"], 32: ["pick", "big:1", "big:2"]}


def resolve(spec, Cout, H, B):
    tab = ops.cfg_table()
    if spec == "pick":
        return ops.pick_cfg(4, Cout, H, H, B, 3, 3, 1, full_cout=False, raw=True, family=2)
    kind, rest = spec.split(":", 1)
    if kind == "cfg":
        i, th, tw = map(int, rest.split(":"))
        return (i, th, tw)
    if kind == "big":
        i = [j for j, c in enumerate(tab) if c[3] == 5][int(rest)]
    else:
        tp, bn = map(int, rest.split("x"))
        i = next(j for j, c in enumerate(tab) if c[3] == 2 and (c[0], c[1]) == (tp, bn))
    sh = ops.launchable_shapes(i, H, H, 3, 3, 1)
    return (i, sh[0][2], sh[0][3])


def trace(args):
    """--trace (a -DCB_TRACE variant library): phase timeline of conv_big_kernel per shape.  Every launch runs behind a 256 MiB device copy (cold
    weights and inputs, as inside the sampling loop) and is timed with events; thread 0 of every workgroup reports s_memtime at 0 entry, 1 addresses
    set up, 2 first copies requested (code + L2 warm-up issued), 3 first stage landed + barrier, 4 K loop done, 5 look-ahead drained (+ K-group
    sum), 6 epilogue done — and the constant 100 MHz clock at entry / exit, from which the workgroups' start skew and the kernel's span follow."""
    import ctypes

    from imagen_pytorch_amd import _abi
    dev = torch.device("cuda:0")
    lib = _abi.load_library()
    assert hasattr(lib, "imagen_debug_conv_big_trace"), "not a -DCB_TRACE library"
    lib.imagen_debug_conv_big_trace.argtypes = [ctypes.c_void_p]
    buf = torch.zeros(8, 4096, 16, dtype=torch.int64, device=dev)   # (the launcher numbers its launches mod 8: one slot each)
    assert lib.imagen_debug_conv_big_trace(buf.data_ptr()) == 0
    n = 256 << 20
    src, dst = torch.empty(n, dtype=torch.uint8, device=dev).fill_(1), torch.empty(n, dtype=torch.uint8, device=dev)
    v = ctypes.c_float()
    flush = lambda: _abi.check(lib.imagen_probe_copy(dst.data_ptr(), src.data_ptr(), n, 1, ops.current_stream_handle(), ctypes.byref(v)), "flush")
    B = args.rows
    g = torch.Generator().manual_seed(0)
    out = {}
    for shp in args.shapes:
        Cin, Cout, H = map(int, shp.split(":"))
        w = torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(9 * Cin)
        pw = ops.pack_weight(w, torch.randn(Cout, generator=g) * 0.1, dev, G=4)
        xs = [ops.act_from_nchw((torch.randn(B, Cin, H, H, generator=g) * 0.7).to(dev)) for _ in range(4)]
        ys = [ops.new_act(B, H, H, Cout, dev) for _ in range(4)]
        x, y = xs[0], ys[0]
        wk = torch.randn(Cout, generator=g).to(dev) * 0.3
        for spec in (args.cands or DEFAULT_CANDS[H]):
            cfg = resolve(spec, Cout, H, B)
            if ops.cfg_table()[cfg[0]][3] != 5:
                continue
            plan = ops.Plan("trace")
            kw = dict(gca=dict(wk=wk, bk=0.1)) if (args.gca and Cout <= 128) else {}
            ops.igemm(plan, x, pw, y, cfg=cfg, label=spec, **kw)
            plan.run()
            torch.cuda.synchronize()
            # eight launches back to back (four rotating buffer sets): what the device sees between one launch's last workgroup and the next one's first
            p8 = ops.Plan("trace8")
            for i in range(7):   # (+ the launch above = 8: every slot written once per round)
                ops.igemm(p8, xs[i % 4], pw, ys[i % 4], cfg=cfg, label=spec, **kw)
            for _ in range(2):
                plan.run(); p8.run()
            torch.cuda.synchronize()
            agg = None
            for _ in range(args.iters):
                buf.zero_()
                plan.run(); p8.run()
                torch.cuda.synchronize()
                t = buf.cpu().double()
                L = sorted([t[s_][t[s_][:, 8] > 0] for s_ in range(8)], key=lambda a: a[:, 8].min().item())
                st, en = [a[:, 8].min().item() * 10.0 for a in L], [a[:, 9].max().item() * 10.0 for a in L]
                mid = L[2:7]
                row = torch.tensor([sum(en[i] - st[i] for i in range(2, 7)) / 5, sum(st[i + 1] - en[i] for i in range(2, 7)) / 5, (st[7] - st[2]) / 5,
                                    sum(((a[:, 6] - a[:, 0]).mean() / ((a[:, 9] - a[:, 8]).mean() * 10.0)).item() for a in mid) / 5]
                                   + [sum((a[:, i + 1] - a[:, i]).mean().item() for a in mid) / 5 for i in range(6)])
                agg = row if agg is None else agg + row
            a = (agg / args.iters).tolist()
            out[f"{shp}:{spec}:back-to-back"] = dict(cfg=list(cfg), span_first_start_to_last_exit_ns=round(a[0]), last_exit_to_next_first_start_ns=round(a[1]),
                                                     period_ns=round(a[2]), shader_clock_ghz=round(a[3], 3), phase_cycles=[round(z) for z in a[4:]])
            acc, us_acc = None, 0.0
            for cold in (True, False):
                acc, us_acc = None, 0.0
                for _ in range(args.iters):
                    if cold:
                        flush()
                    buf.zero_()
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    plan.run()
                    e1.record()
                    torch.cuda.synchronize()
                    us_acc += e0.elapsed_time(e1) * 1e3
                    t = buf.cpu().double().view(-1, 16)
                    t = t[t[:, 8] > 0]
                    ph = (t[:, 1:7] - t[:, 0:6]).mean(0)
                    r0, r1 = t[:, 8], t[:, 9]
                    ids = set(zip(t[:, 10].long().tolist(), (t[:, 11].long() & 0xff00).tolist()))
                    row = torch.cat((ph, torch.tensor([(t[:, 6] - t[:, 0]).mean(), (r1 - r0).mean() * 10.0, (r0 - r0.min()).mean() * 10.0, (r0 - r0.min()).max() * 10.0,
                                                       (r1 - r0.min()).mean() * 10.0, (r1.max() - r0.min()) * 10.0, float(t.shape[0]), float(len(ids))])))
                    acc = row if acc is None else acc + row
                a = (acc / args.iters).tolist()
                out[f"{shp}:{spec}:{'cold' if cold else 'warm'}"] = dict(
                    cfg=list(cfg), phase_cycles=[round(z) for z in a[:6]], per_wg_cycles=round(a[6]), per_wg_ns=round(a[7]),
                    start_skew_ns=dict(mean=round(a[8]), max=round(a[9])), end_after_first_start_ns=dict(mean=round(a[10]), max=round(a[11])),
                    event_us=round(us_acc / args.iters, 2), wgs=round(a[12]), distinct_xcc_cu=round(a[13]))
    print(json.dumps(dict(tag=args.tag, phases=["address setup", "warm-ups + first copies requested", "first stage landed + barrier", "K loop", "drain (+ K-group sum)", "epilogue"],
                          trace=out)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--trace", action="store_true", help="phase timeline of conv_big_kernel (needs a -DCB_TRACE library)")
    ap.add_argument("--tag", default="")
    ap.add_argument("--iters", type=int, default=40)
    ap.add_argument("--rows", type=int, default=16)
    ap.add_argument("--shapes", nargs="*", default=DEFAULT_SHAPES)
    ap.add_argument("--cands", nargs="*", default=None)
    ap.add_argument("--gca", action="store_true", help="GlobalContext partials from the epilogue (Cout <= 128 shapes)")
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    if args.trace:
        return trace(args)
    dev = torch.device("cuda:0")
    B = args.rows
    g = torch.Generator().manual_seed(0)
    lines = []
    for shp in args.shapes:
        Cin, Cout, H = map(int, shp.split(":"))
        w = torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(9 * Cin)
        pw = ops.pack_weight(w, torch.randn(Cout, generator=g) * 0.1, dev, G=4)
        xs = [ops.act_from_nchw((torch.randn(B, Cin, H, H, generator=g) * 0.7).to(dev)) for _ in range(4)]
        ys = [ops.new_act(B, H, H, Cout, dev) for _ in range(4)]
        wk = torch.randn(Cout, generator=g).to(dev) * 0.3
        first = None
        for spec in (args.cands or DEFAULT_CANDS[H]):
            try:
                cfg = resolve(spec, Cout, H, B)
                plan = ops.Plan("bench")
                for i in range(4):
                    kw = dict(gca=dict(wk=wk, bk=0.1)) if (args.gca and Cout <= 128) else {}
                    ops.igemm(plan, xs[i], pw, ys[i], cfg=cfg, label=spec, **kw)
                plan.run()
                torch.cuda.synchronize()
            except Exception as e:   # noqa: BLE001
                lines.append(dict(shape=shp, cand=spec, error=str(e)[:200]))
                print(json.dumps(lines[-1]), flush=True)
                continue
            out = ys[0].t.float().clone()
            if first is None:
                first = out
            err = ((out - first).norm() / first.norm()).item()
            for _ in range(3):
                plan.run()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.iters):
                plan.run()
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / (4 * args.iters)
            fl = 2.0 * B * H * H * Cout * 9 * Cin
            lines.append(dict(shape=shp, cand=spec, cfg=list(cfg), fam=ops.cfg_table()[cfg[0]][3], us=round(us, 2), tflops=round(fl / us / 1e6, 1),
                              dist_to_first=float(f"{err:.3e}"), gca=bool(args.gca and Cout <= 128)))
            print(json.dumps(lines[-1]), flush=True)
    if args.out:
        with open(args.out, "a") as f:
            for ln in lines:
                f.write(json.dumps(ln) + "\n")


if __name__ == "__main__":
    main()
