"""In-GRAPH per-op timing of the denoiser steps: a rocprofv3 kernel trace of a short sampling run joined with the plan's op labels.

Event-timed eager launches (round-2 probe step_profile.py) carry the event records' own cost and run every kernel cold behind an idle queue;
the hipGraph replays of the real run do not.  This tool takes the dispatch timestamps of the replays themselves:

    cd /tmp && export TMPDIR=/tmp
    rocprofv3 --kernel-trace --output-format csv -d /tmp/gp -- python $REPO/tools/graph_profile.py run --steps 12 --plan-out /tmp/plan.json
    python $REPO/tools/graph_profile.py analyze $(find /tmp/gp -name '*kernel_trace.csv') /tmp/plan.json [--top 50]

`run` samples `--steps` DDPM steps per stage (graph replays of the step plan) and writes the op list of each stage's step plan;
`analyze` finds, per stage, the run of dispatches that repeats with the plan's length, averages each position over the replays and
prints duration + the idle gap in front of the kernel by op kind, by label class and for the top launches.
"""
import argparse
import collections
import csv
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(args):
    import torch

    import bench
    from imagen_pytorch_amd import _abi, ops

    dev = torch.device("cuda", 0)
    if args.dup_igemm:
        # timing experiment (results garbage where an op accumulates): every conv / GEMM launch is issued twice back to back, so the
        # trace holds each one cold (operands last touched by other kernels) and warm (weights and input tile just read by itself)
        add0 = ops.Plan.add

        def add_twice(self, struct, label="", keep=()):
            r = add0(self, struct, label, keep)
            if _abi.STRUCT_KIND[type(struct)] == _abi.ENUMS["IMAGEN_OP_IGEMM"]:
                self.ops.append((_abi.STRUCT_KIND[type(struct)], struct, (label or "igemm") + ".again"))
            return r
        ops.Plan.add = add_twice
    imagen = bench.build_imagen(1000, dev)
    te = torch.randn(args.batch, 256, 768, device=dev)
    imagen.sample(text_embeds=te, cond_scale=3.0, use_tqdm=False, seed=1, max_steps=args.steps)
    torch.cuda.synchronize()
    K_IGEMM = _abi.ENUMS["IMAGEN_OP_IGEMM"]
    kind_name = {v: k.replace("IMAGEN_OP_", "").lower() for k, v in _abi.ENUMS.items() if k.startswith("IMAGEN_OP_")}
    tab = ops.cfg_table()
    out = []
    for sidx, st in imagen._stages.items():
        rows = []
        for kind, p, label in st["plan"].ops:
            fl = by = 0.0
            desc, cls = "", re.sub(r"\d+", "#", label)
            if kind == K_IGEMM:
                fl = bench.igemm_flops(p)
                cin = p.C1 + p.C2
                by = 2.0 * p.B * (p.H * p.W * cin + p.OH * p.OW * p.Cout * (2 if p.out_mode == 2 else 1)) + 2.0 * p.KH * p.KW * cin * p.Cout
                by += 2.0 * p.B * p.OH * p.OW * p.Cout * (bool(p.res) + bool(p.addend))
                desc = (f"{cin}->{p.Cout} k{p.KH} s{p.stride} @{p.H}x{p.W} B{p.B} cfg{p.cfg}{tab[p.cfg]} t{p.TH}x{p.TW}"
                        f"{' pro' if (p.pa or p.rs or p.ssq_a) else ''}{' gca' if p.gca_part else ''}")
                cls += f" [{cin}->{p.Cout} k{p.KH} @{p.H}]"
            rows.append(dict(kind=kind_name.get(kind, str(kind)), label=label, cls=cls, desc=desc, flops=fl, bytes=by,
                             cfg=int(p.cfg) if kind == K_IGEMM else -1))
        out.append(dict(stage=str(sidx[:3]), ops=rows))
    json.dump(dict(steps=args.steps, stages=out), open(args.plan_out, "w"))
    print(f"wrote {args.plan_out}: " + ", ".join(f"{len(s['ops'])} ops" for s in out))


def short(name):
    m = re.search(r"(\w+)\s*(<|\()", name.replace("void ", "").replace("(anonymous namespace)::", ""))
    return m.group(1) if m else name[:40]


MULTI = {"quantile"}        # op kinds that launch several kernels


def matches(kind, kname):
    if kind == "igemm":
        return kname in ("igemm_kernel", "conv_dma_kernel", "conv_stream_kernel", "conv_pw_kernel", "conv_big_kernel", "conv_pro_kernel", "conv_gemm_kernel", "conv_small_kernel")
    return kname.startswith(kind)


def find_period(names, lo, hi, reps, start):
    """(offset, period): first offset >= start from which the name sequence repeats `reps - 1` times with a period in [lo, hi]."""
    for off in range(start, len(names) - lo * reps):
        for per in range(lo, hi + 1):
            if off + per * reps > len(names):
                break
            if names[off:off + per * (reps - 1)] == names[off + per:off + per * reps] and len(set(names[off:off + per])) > 4:
                return off, per      # (more than a handful of distinct kernels: a long run of ONE kernel — the upload copies of the weight
                                     # packing, ~1000 __amd_rocclr_copyBuffer dispatches since round 5 — is periodic with every period)
    return -1, 0


def join(names, plan):
    """Per stage: (offset of the first replay in the dispatch list, kernels per replay, owner[i] = plan op of kernel i of a replay)."""
    steps = plan["steps"]
    start = 0
    out = []
    for st in plan["stages"]:
        ops = st["ops"]
        n = len(ops)
        off, per = find_period(names, n, n + 12, steps, start)
        if off < 0:
            print(f"stage {st['stage']}: no run of {steps} replays of ~{n} kernels found after dispatch {start} ({len(names)} dispatches)", file=sys.stderr)
            out.append(None)
            continue
        # rotate the period so that it starts at the first op's kernel (the search may land mid-replay on a periodic tail)
        while not matches(ops[0]["kind"], names[off]) or not matches(ops[1]["kind"], names[off + 1]):
            off += 1
        start = off + per * (steps - 1)
        # kernel -> op assignment inside one period
        owner, j, had = [], 0, False
        for i in range(per):
            k = names[off + i]
            if matches(ops[j]["kind"], k) and (not had or ops[j]["kind"] in MULTI):
                owner.append(j); had = True
            elif j + 1 < n and matches(ops[j + 1]["kind"], k):
                j += 1; owner.append(j); had = True
            else:
                owner.append(-1)          # engine bookkeeping (step_advance)
        assert j == n - 1, (j, n, per)
        out.append((off, per, owner))
    return out


def pmc_join(csv_path, plan):
    """rocprofv3 --pmc counter_collection.csv of a `run` joined with its plan: per stage a list (one entry per plan op) of
    {label, kind, desc, flops, bytes, <COUNTER>: value per launch averaged over the replays}."""
    disp = {}
    for r in csv.DictReader(open(csv_path)):
        d = disp.setdefault(int(r["Dispatch_Id"]), [short(r["Kernel_Name"]), {}])
        d[1][r["Counter_Name"]] = d[1].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    order = [disp[k] for k in sorted(disp)]
    names = [d[0] for d in order]
    steps = plan["steps"]
    res = []
    for st, j in zip(plan["stages"], join(names, plan)):
        if j is None:
            res.append(None)
            continue
        off, per, owner = j
        reps = range(1, steps - 1)
        ops = [dict(o) for o in st["ops"]]
        for r in reps:
            for i in range(per):
                if owner[i] < 0:
                    continue
                for c, v in order[off + r * per + i][1].items():
                    ops[owner[i]][c] = ops[owner[i]].get(c, 0.0) + v / len(reps)
        res.append(ops)
    return res


def pmc(args):
    plan = json.load(open(args.plan))
    res = pmc_join(args.csv_in, plan)
    counters = sorted({k for ops in res if ops for o in ops for k in o if k.isupper()})
    out = dict(counters=counters, note="per-launch averages over the graph replays of a short sampling run; FETCH_SIZE / WRITE_SIZE in KiB as rocprofv3 "
               "reports them (gfx950: FETCH_SIZE counts 128-byte requests as 64 bytes, consumers double it)", stages=[])
    for st, ops in zip(plan["stages"], res):
        if not ops:
            continue
        by_kind = collections.defaultdict(lambda: collections.defaultdict(float))
        for o in ops:
            by_kind[o["kind"]]["launches"] += 1
            for c in counters:
                by_kind[o["kind"]][c] += o.get(c, 0.0)
        igemm = [dict(label=o["label"], desc=o["desc"], flops=o["flops"], bytes=o["bytes"], **{c: round(o.get(c, 0.0), 2) for c in counters})
                 for o in ops if o["kind"] == "igemm"]
        out["stages"].append(dict(stage=st["stage"], by_kind={k: dict(v) for k, v in by_kind.items()}, igemm=igemm))
    json.dump(out, open(args.out, "w"), indent=1)
    for s_ in out["stages"]:
        print("stage", s_["stage"], {k: {c: round(v, 1) for c, v in d.items()} for k, d in list(s_["by_kind"].items())[:4]})


def analyze(args):
    plan = json.load(open(args.plan))
    rows = list(csv.DictReader(open(args.trace)))
    ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])) for r in rows), key=lambda t: t[0])
    names = [e[2] for e in ev]
    steps = plan["steps"]
    for st, jn in zip(plan["stages"], join(names, plan)):
        if jn is None:
            continue
        off, per, owner = jn
        ops = st["ops"]
        n = len(ops)
        reps = range(1, steps - 1)        # the first replay follows host work; average the rest
        dur = [0.0] * (n + 1)
        gap = [0.0] * (n + 1)
        for r in reps:
            for i in range(per):
                s, e, _ = ev[off + r * per + i]
                dur[owner[i]] += (e - s) / 1e3 / len(reps)
                gap[owner[i]] += max(0, s - ev[off + r * per + i - 1][1]) / 1e3 / len(reps)
        span = sum((ev[off + (r + 1) * per][0] - ev[off + r * per][0]) for r in reps) / 1e3 / len(reps)
        ops = ops + [dict(kind="bookkeeping", label="step_advance", cls="step_advance", desc="", flops=0.0, bytes=0.0)]
        n += 1
        tot = sum(dur) + sum(gap)
        print(f"\n=== stage {st['stage']}: {per} kernels per replay, replay period {span / 1e3:.3f} ms  busy {sum(dur) / 1e3:.3f} ms  gaps {sum(gap) / 1e3:.3f} ms"
              f"  (mean kernel {sum(dur) / per:.2f} us, mean gap {sum(gap) / per:.2f} us)")
        by_kind = collections.defaultdict(lambda: [0, 0.0, 0.0])
        by_cls = collections.defaultdict(lambda: [0, 0.0, 0.0, 0.0])
        for o, d, g in zip(ops, dur, gap):
            k = by_kind[o["kind"]]
            k[0] += 1; k[1] += d; k[2] += g
            c = by_cls[o["cls"]]
            c[0] += 1; c[1] += d; c[2] += g; c[3] += o["flops"]
        print("-- by kind: launches, kernel us, gap us, share of span")
        for k, (c, d, g) in sorted(by_kind.items(), key=lambda kv: -(kv[1][1] + kv[1][2])):
            print(f"  {k:16s} {c:4d} {d:9.1f} {g:8.1f} {100 * (d + g) / tot:5.1f}%   ({d / c:6.2f} us each)")
        print("-- by label class: launches x mean kernel us (+ mean gap) = total, share, TFLOP/s")
        for k, (c, d, g, fl) in sorted(by_cls.items(), key=lambda kv: -(kv[1][1] + kv[1][2]))[: args.top]:
            tf = f"{fl / d / 1e6:7.1f} TF" if fl else ""
            print(f"  {k:58s} {c:3d} x {d / c:7.2f} (+{g / c:4.2f}) = {d + g:8.1f} us {100 * (d + g) / tot:5.1f}% {tf}")
        print("-- top launches")
        for d, g, o in sorted(zip(dur, gap, ops), key=lambda t: -t[0])[: args.top]:
            extra = f"{o['flops'] / d / 1e6:7.1f} TF {o['bytes'] / d / 1e3:7.0f} GB/s" if o["flops"] else ""
            print(f"  {o['label']:34s} {d:7.2f} us {extra}  {o['desc']}")
        if args.csv:
            with open(args.csv + f".stage{plan['stages'].index(st) + 1}.csv", "w") as f:
                f.write("label,kind,desc,kernel_us,gap_us,flops,bytes\n")
                for o, d, g in zip(ops, dur, gap):
                    f.write(",".join(str(x).replace(",", ";") for x in (o["label"], o["kind"], o["desc"], round(d, 3), round(g, 3), o["flops"], o["bytes"])) + "\n")


def main():
    ap = argparse.ArgumentParser()
    sub = ap.add_subparsers(dest="cmd", required=True)
    r = sub.add_parser("run")
    r.add_argument("--steps", type=int, default=12)
    r.add_argument("--batch", type=int, default=8)
    r.add_argument("--plan-out", default="/tmp/plan.json")
    r.add_argument("--dup-igemm", action="store_true", help="issue every conv / GEMM launch twice (cold vs warm timing experiment)")
    a = sub.add_parser("analyze")
    a.add_argument("trace")
    a.add_argument("plan")
    a.add_argument("--top", type=int, default=50)
    a.add_argument("--csv", default="")
    c = sub.add_parser("pmc")
    c.add_argument("csv_in")
    c.add_argument("plan")
    c.add_argument("out")
    args = ap.parse_args()
    dict(run=run, analyze=analyze, pmc=pmc)[args.cmd](args)


if __name__ == "__main__":
    main()
