#!/usr/bin/env python
"""Per-launch timer of the ROWCHAIN kernels on the benchmark's shapes (the A/B tool of csrc/rowchain.hip).

    [IMAGEN_LIB_PATH=<variant .so>] rocprofv3 --kernel-trace --output-format csv -d /tmp/cb -- python tools/chain_bench.py [--tag name] --list /tmp/cases.json
    python tools/chain_bench.py --parse /tmp/cb /tmp/cases.json        -> one JSON line: {case: median us per launch}

Every case is launched `--reps` times, each behind a 256 MiB device copy (so the weights are no more L2-resident than inside the
sampling loop, where a launch last ran a whole denoiser step earlier); the kernel durations come from the rocprofv3 trace of the run
(HIP events around a 10 us launch measure the launch path, not the kernel).  Cases: (mode, rows per image N, channels C) at 16 rows
of the CFG batch, with 32- and 64-row tiles."""
import argparse
import csv
import glob
import json
import math
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

CASES = [  # (mode, N, C): README unet1's 32^2 / 16^2 / 8^2 / mid levels and unet2's 32^2 level, 16 rows
    ("xattn", 1024, 32), ("xattn", 1024, 64), ("xattn", 256, 64), ("xattn", 256, 128), ("xattn", 64, 128), ("xattn", 64, 256), ("xattn", 1024, 256),
    ("ff", 1024, 32), ("ff", 1024, 64), ("ff", 256, 64), ("ff", 256, 128), ("ff", 64, 128), ("ff", 64, 256), ("ff", 1024, 256),
    ("qkv", 1024, 32), ("qkv", 256, 128), ("qkv", 64, 256), ("qkv", 1024, 256),
    ("resprep", 4096, 128), ("resprep", 1024, 256),
]


def build_case(ops, torch, dev, mode, N, C, tile64, B=16):
    heads, dh, inner = 8, 64, 512
    plan = ops.Plan(f"{mode}-{N}-{C}")
    rnd = lambda *s: (torch.randn(*s) * 0.5).half().to(dev)
    act = lambda t: ops.Act(t, B, 1, N, t.shape[-1], t.shape[-1], N * t.shape[-1])
    pw = lambda co, ci, bias=False: ops.pack_weight(torch.randn(co, ci) / math.sqrt(ci), (torch.randn(co) * 0.1) if bias else None, dev)
    g = lambda n: (1 + 0.1 * torch.randn(n)).to(dev)
    ops.CHAIN_TILE64_MIN_ROWS = 1 if tile64 else 1 << 30
    if mode == "ff":
        ops.rowchain_ff(plan, act(rnd(B, 1, N, inner)), act(rnd(B, 1, N, C)), act(rnd(B, 1, N, C)), pw(C, inner), g(C), pw(2 * C, C), g(C), pw(C, 2 * C),
                        g(2 * C), rows_per_batch=N, ssq_out=torch.empty(B * N, device=dev))
    elif mode == "xattn":
        J, Jp = 41, 64
        khat, vt = rnd(B, heads, Jp, dh), rnd(B, heads, dh, Jp)
        ops.rowchain_xattn(plan, act(rnd(B, 1, N, C)), act(rnd(B, 1, N, C)), pw(inner, C), g(C), pw(C, inner), g(C), khat, vt, heads=heads, J=J,
                           k_strides=(heads * Jp * dh, Jp * dh, dh), vt_strides=(heads * dh * Jp, dh * Jp, Jp), q_scale=g(dh), q_mult=8 * ops.LOG2E,
                           rows_per_batch=N, ssq_out=torch.empty(B * N, device=dev))
    elif mode == "qkv":
        Jp = ops._round_up(42 + N, 32)
        khat, vt = torch.zeros(B, Jp, dh, dtype=torch.float16, device=dev), torch.zeros(B, dh, Jp, dtype=torch.float16, device=dev)
        ops.rowchain_qkv(plan, act(rnd(B, 1, N, C)), act(rnd(B, 1, N, inner + 128)), pw(inner + 128, C), g(C), khat, vt, g(dh), heads=heads, r0=42,
                         k_strides=(Jp * dh, 0, dh), vt_strides=(dh * Jp, 0, Jp), rows_per_batch=N)
    else:
        C1, C2 = C, C // 2
        out = act(rnd(B, 1, N, C))
        ops.rowchain_resprep(plan, act(rnd(B, 1, N, C1)), act(rnd(B, 1, N, C2)), act(rnd(B, 1, N, C)), torch.rand(B, C).to(dev), out, pw(C, C1 + C2, True),
                             rows_per_batch=N, ssq_out=torch.empty(B * N, device=dev))
        assert ops.request_prep(out, act(rnd(B, 1, N, C2)), torch.rand(B * N).to(dev), 0.5, g(C + C2)) is not None
    return plan


def run(args):
    import ctypes

    import torch
    from imagen_pytorch_amd import _abi, ops

    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    lib = _abi.load_library()
    n = 256 << 20
    src, dst = torch.empty(n, dtype=torch.uint8, device=dev).fill_(1), torch.empty(n, dtype=torch.uint8, device=dev)
    v = ctypes.c_float()
    order = []
    have_resprep = hasattr(ops, "rowchain_resprep")
    for mode, N, C in CASES:
        for tile64 in (0, 1):
            if N % 64 and tile64:
                continue
            if mode == "resprep" and (not have_resprep or os.environ.get("CHAIN_BENCH_NO_RESPREP")):
                continue
            plan = build_case(ops, torch, dev, mode, N, C, tile64)
            plan.run()
            torch.cuda.synchronize()
            for _ in range(args.reps):
                _abi.check(lib.imagen_probe_copy(dst.data_ptr(), src.data_ptr(), n, 1, ops.current_stream_handle(), ctypes.byref(v)), "flush")
                plan.run()
            torch.cuda.synchronize()
            order.append(dict(case=f"{mode}:N{N}:C{C}:t{64 if tile64 else 32}", launches=args.reps + 1))
    json.dump(dict(tag=args.tag, lib=os.path.basename(_abi.LIB_PATH), cases=order), open(args.list, "w"))
    print(f"chain_bench: {len(order)} cases x {args.reps} launches done ({os.path.basename(_abi.LIB_PATH)})", flush=True)


def trace(args):
    """--trace (a -DROWCHAIN_TRACE variant library): per-phase s_memtime deltas of thread 0 of every workgroup, averaged over workgroups and launches."""
    import ctypes

    import torch
    from imagen_pytorch_amd import _abi, ops

    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    lib = _abi.load_library()
    assert hasattr(lib, "imagen_debug_rowchain_trace"), "not a -DROWCHAIN_TRACE library"
    lib.imagen_debug_rowchain_trace.argtypes = [ctypes.c_void_p]
    buf = torch.zeros(8192, 16, dtype=torch.int64, device=dev)
    assert lib.imagen_debug_rowchain_trace(buf.data_ptr()) == 0
    n = 256 << 20
    src, dst = torch.empty(n, dtype=torch.uint8, device=dev).fill_(1), torch.empty(n, dtype=torch.uint8, device=dev)
    v = ctypes.c_float()
    out = {}
    for mode, N, C in CASES:
        if mode not in ("ff", "xattn"):
            continue
        for tile64 in (0, 1):
            if (N % 64 and tile64) or (tile64 != (16 * N >= 16384)):
                continue
            plan = build_case(ops, torch, dev, mode, N, C, tile64)
            plan.run()
            torch.cuda.synchronize()
            wgs = 16 * N // (64 if tile64 else 32)
            acc = None
            for _ in range(args.reps):
                _abi.check(lib.imagen_probe_copy(dst.data_ptr(), src.data_ptr(), n, 1, ops.current_stream_handle(), ctypes.byref(v)), "flush")
                buf.zero_()
                plan.run()
                torch.cuda.synchronize()
                t = buf[:wgs].cpu().double()
                nst = int((t[0] > 0).sum())
                d = torch.cat(((t[:, 1:nst] - t[:, :nst - 1]).mean(0), torch.tensor([t[:, nst - 1].max() - t[:, 0].min(), (t[:, nst - 1] - t[:, 0]).mean()])))
                acc = d if acc is None else acc + d
            acc = (acc / args.reps).tolist()
            out[f"{mode}:N{N}:C{C}:t{64 if tile64 else 32}"] = dict(phase_cycles=[round(x) for x in acc[:-2]], first_to_last_cycles=round(acc[-2]),
                                                                    per_wg_cycles=round(acc[-1]), wgs=wgs)
    print(json.dumps(dict(tag=args.tag, trace=out)))


def parse(trace_dir, list_path):
    meta = json.load(open(list_path))
    f = next(iter(sorted(glob.glob(os.path.join(trace_dir, "**", "*kernel_trace.csv"), recursive=True))))
    rows = [r for r in csv.DictReader(open(f)) if "rowchain_kernel" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    out, i = {}, 0
    for c in meta["cases"]:
        d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows[i:i + c["launches"]]][1:]   # (the first launch is the warm-up)
        i += c["launches"]
        out[c["case"]] = round(statistics.median(d), 2) if d else None
    assert i == len(rows), (i, len(rows))
    print(json.dumps(dict(tag=meta["tag"], lib=meta["lib"], us=out)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tag", default="")
    ap.add_argument("--reps", type=int, default=12)
    ap.add_argument("--list", default="/tmp/chain_cases.json")
    ap.add_argument("--parse", nargs=2, metavar=("TRACE_DIR", "LIST"))
    ap.add_argument("--trace", action="store_true", help="phase timeline of the ff / xattn chains (needs a -DROWCHAIN_TRACE library)")
    args = ap.parse_args()
    if args.parse:
        parse(*args.parse)
    elif args.trace:
        trace(args)
    else:
        run(args)


if __name__ == "__main__":
    main()
