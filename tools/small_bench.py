#!/usr/bin/env python
"""Per-launch timer of the small-map 3x3 convs (the A/B tool of csrc/conv_small.hip): every case is planned twice in one process — with the
small-map family (ops.CONV_SMALL = 1) and without (the planner's previous pick) — and launched `--reps` times, each behind a 256 MiB
device copy (weights and inputs as cold as inside the sampling loop).  Durations come from the rocprofv3 trace of the run:

    rocprofv3 --kernel-trace --output-format csv -d /tmp/sb -- python tools/small_bench.py --list /tmp/small_cases.json
    python tools/small_bench.py --parse /tmp/sb /tmp/small_cases.json        -> one JSON line per case: median us with / without, family picked
    IMAGEN_LIB_PATH=<a -DCS_TRACE library> python tools/small_bench.py --trace      -> s_memtime phase deltas of conv_small_kernel per case

Cases: (C1, C2, Cout, H, form) at 16 rows of the CFG batch — README unet1's 8^2 / 16^2 layers, C2's 512- / 1024-channel ones, a few 32^2 ones.
form: "block1" (ssq statistics + gain + SiLU, ssq_out), "block2" (the same + per-batch scale / shift), "raw" (input activated by its
producer), "post" (raw + the next Block's prologue applied to the output), "gca" (raw + GlobalContext partials)."""
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

CASES = [
    (128, 0, 128, 8, "block1"), (128, 0, 128, 8, "gca"), (128, 0, 128, 8, "post"), (256, 0, 256, 8, "block2"), (256, 128, 256, 8, "block1"),
    (64, 0, 64, 16, "gca"), (64, 0, 64, 16, "post"), (128, 64, 128, 16, "block1"), (128, 0, 128, 16, "block2"), (128, 0, 128, 16, "gca"),
    (512, 0, 512, 16, "block2"), (1024, 0, 1024, 8, "block2"), (1024, 0, 512, 8, "block1"),
    (64, 32, 64, 32, "block1"), (64, 0, 64, 32, "gca"), (128, 0, 128, 32, "gca"), (128, 0, 128, 32, "post"),
]


def build_case(ops, torch, dev, case, small, B=16):
    C1, C2, Cout, H, form = case
    g = torch.Generator().manual_seed(0)
    rn = lambda *s: torch.randn(*s, generator=g)
    plan = ops.Plan(f"{C1}+{C2}->{Cout}@{H}:{form}")
    pw = ops.pack_weight(rn(Cout, C1 + C2, 3, 3) / math.sqrt(9 * (C1 + C2)), rn(Cout) * 0.1, dev, G=4)
    x1 = ops.act_from_nchw((rn(B, C1, H, H) * 0.7).to(dev))
    x2 = ops.act_from_nchw((rn(B, C2, H, H) * 0.7).to(dev)) if C2 else None
    y = ops.new_act(B, H, H, Cout, dev)
    kw = {}
    if form in ("block1", "block2"):
        kw.update(ssq_a=(x1.t.float() ** 2).sum(-1).reshape(-1).contiguous(), pa=(1 + 0.1 * rn(B, C1 + C2)).to(dev), act_in=ops.ACT_SILU)
        if x2 is not None:
            kw.update(ssq_b=(x2.t.float() ** 2).sum(-1).reshape(-1).contiguous(), ssq_wb=0.5)
        if form == "block2":
            kw.update(ps=(0.1 * rn(B, C1 + C2)).to(dev), pstride=C1 + C2)
        else:
            kw["pa"] = kw["pa"][:1].contiguous()
            kw.update(ssq_out=torch.empty(B * H * H, device=dev))
    elif form == "post":
        kw.update(post=dict(pa=(1 + 0.1 * rn(B, Cout)).to(dev), ps=(0.1 * rn(B, Cout)).to(dev), pstride=Cout))
    elif form == "gca":
        kw.update(gca=dict(wk=(rn(Cout) / math.sqrt(Cout)).to(dev), bk=0.1))
    ops.CONV_SMALL, ops.SMALL_MAX_ROWS = (1 if small else 0), 1 << 30   # (every case is offered to the family: the 32^2 ones too)
    p = ops.igemm(plan, x1, pw, y, x2=x2, label=plan.name, **kw)
    ops.CONV_SMALL = 1
    return plan, ops.cfg_table()[p.cfg], (p.cfg, p.TH, p.TW)


def flusher(torch, lib, _abi, ops, dev):
    import ctypes
    n = 256 << 20
    src, dst = torch.empty(n, dtype=torch.uint8, device=dev).fill_(1), torch.empty(n, dtype=torch.uint8, device=dev)
    v = ctypes.c_float()
    return lambda: _abi.check(lib.imagen_probe_copy(dst.data_ptr(), src.data_ptr(), n, 1, ops.current_stream_handle(), ctypes.byref(v)), "flush")


def run(args):
    import torch
    from imagen_pytorch_amd import _abi, ops

    dev = torch.device("cpu" if "emul" in os.path.basename(os.environ.get("IMAGEN_LIB_PATH", "")) else "cuda:0")   # (cpu: a dry check of this tool on the emulated library)
    if dev.type == "cpu":
        torch.cuda.synchronize = lambda *a, **k: None
        ops.current_stream_handle = lambda: 0
    lib = _abi.load_library()
    flush = flusher(torch, lib, _abi, ops, dev) if dev.type != "cpu" else (lambda: None)
    order = []
    for case in CASES:
        for small in (1, 0):
            plan, tab, cfg = build_case(ops, torch, dev, case, small)
            plan.run()
            torch.cuda.synchronize()
            for _ in range(args.reps):
                flush()
                plan.run()
            torch.cuda.synchronize()
            order.append(dict(case="{}+{}->{}@{}:{}".format(*case), small=small, family=tab[3], tile=[tab[0], tab[1]], cfg=list(cfg), launches=args.reps + 1))
    json.dump(dict(tag=args.tag, lib=os.path.basename(_abi.LIB_PATH), cases=order), open(args.list, "w"))
    print(f"small_bench: {len(order)} plans x {args.reps} launches done ({os.path.basename(_abi.LIB_PATH)})", flush=True)


def trace(args):
    import ctypes

    import torch
    from imagen_pytorch_amd import _abi, ops

    dev = torch.device("cuda:0")
    lib = _abi.load_library()
    assert hasattr(lib, "imagen_debug_conv_small_trace"), "not a -DCS_TRACE library"
    lib.imagen_debug_conv_small_trace.argtypes = [ctypes.c_void_p]
    buf = torch.zeros(4096, 8, dtype=torch.int64, device=dev)
    assert lib.imagen_debug_conv_small_trace(buf.data_ptr()) == 0
    flush = flusher(torch, lib, _abi, ops, dev)
    out = {}
    for case in CASES:
        plan, tab, cfg = build_case(ops, torch, dev, case, 1)
        if tab[3] != 8:
            continue
        plan.run()
        torch.cuda.synchronize()
        acc = None
        for _ in range(args.reps):
            flush()
            buf.zero_()
            plan.run()
            torch.cuda.synchronize()
            t = buf.cpu().double()
            t = t[t[:, 0] > 0]
            d = torch.cat(((t[:, 1:6] - t[:, 0:5]).mean(0), torch.tensor([t[:, 5].max() - t[:, 0].min(), (t[:, 5] - t[:, 0]).mean(), float(t.shape[0])])))
            acc = d if acc is None else acc + d
        acc = (acc / args.reps).tolist()
        out["{}+{}->{}@{}:{}".format(*case)] = dict(phases=["requests issued", "staged + barrier", "K loop", "K-split sum", "epilogue"],
                                                    phase_cycles=[round(x) for x in acc[:5]], first_to_last_cycles=round(acc[5]), per_wg_cycles=round(acc[6]),
                                                    wgs=round(acc[7]), tile=[tab[0], tab[1]])
    print(json.dumps(dict(tag=args.tag, trace=out)))


def parse(trace_dir, list_path):
    meta = json.load(open(list_path))
    f = next(iter(sorted(glob.glob(os.path.join(trace_dir, "**", "*kernel_trace.csv"), recursive=True))))
    rows = [r for r in csv.DictReader(open(f)) if ("igemm_kernel" in r["Kernel_Name"] or "conv_" in r["Kernel_Name"]) and "at::" not in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    i, res = 0, {}
    for c in meta["cases"]:
        d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows[i:i + c["launches"]]][1:]   # (the first launch is the warm-up)
        i += c["launches"]
        e = res.setdefault(c["case"], {})
        e["small_us" if c["small"] else "before_us"] = round(statistics.median(d), 2) if d else None
        e["small_tile" if c["small"] else "before"] = (f"fam{c['family']} {c['tile'][0]}x{c['tile'][1]}")
    assert i == len(rows), (i, len(rows))
    for k, v in res.items():
        print(json.dumps(dict(case=k, tag=meta["tag"], **v)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tag", default="")
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--list", default="/tmp/small_cases.json")
    ap.add_argument("--parse", nargs=2, metavar=("TRACE_DIR", "LIST"))
    ap.add_argument("--trace", action="store_true", help="phase timeline of conv_small_kernel (needs a -DCS_TRACE library)")
    args = ap.parse_args()
    if args.parse:
        parse(*args.parse)
    elif args.trace:
        trace(args)
    else:
        run(args)


if __name__ == "__main__":
    main()
