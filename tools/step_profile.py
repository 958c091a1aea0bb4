"""Per-op timing of one denoiser step of each cascade stage (C3 workload), HIP events around every launch.

    python tools/step_profile.py [--reps 5] [--top 40] [--csv gpurun_out/step_profile.csv]

Prints, per stage: the time by op kind, by label class (layer names with indices collapsed) and the top individual
launches with their achieved TFLOP/s / GB/s (algorithmic bytes: inputs + outputs once).  Eager launches on the current
stream, so the sum is a little above the hipGraph step time (launch gaps) but the split is the same.
"""
import argparse
import collections
import ctypes
import os
import re
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--top", type=int, default=40)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--csv", default="")
    args = ap.parse_args()
    from imagen_pytorch_amd import _abi, ops

    dev = torch.device("cuda", 0)
    imagen = bench.build_imagen(1000, dev)
    te = torch.randn(args.batch, 256, 768, device=dev)
    imagen.sample(text_embeds=te, cond_scale=3.0, use_tqdm=False, seed=1, max_steps=3)
    lib = _abi.load_library()
    K_IGEMM = _abi.ENUMS["IMAGEN_OP_IGEMM"]
    kind_name = {v: k.replace("IMAGEN_OP_", "").lower() for k, v in _abi.ENUMS.items() if k.startswith("IMAGEN_OP_")}
    h = torch.cuda.current_stream().cuda_stream
    tab = ops.cfg_table()
    rows = []
    for sidx, st in imagen._stages.items():
        plan = st["plan"]
        n = len(plan.ops)
        acc = [0.0] * n
        for rep in range(args.reps + 1):
            st["step_ptr"].zero_()
            evs = []
            for kind, struct, label in plan.ops:
                e0, e1 = ctypes.c_void_p(), ctypes.c_void_p()
                lib.imagen_event_create(ctypes.byref(e0))
                lib.imagen_event_create(ctypes.byref(e1))
                lib.imagen_event_record(e0, h)
                _abi.check(lib.imagen_launch(kind, ctypes.addressof(struct), h))
                lib.imagen_event_record(e1, h)
                evs.append((e0, e1))
            torch.cuda.synchronize()
            for i, (e0, e1) in enumerate(evs):
                ms = ctypes.c_float()
                lib.imagen_event_elapsed_ms(e0, e1, ctypes.byref(ms))
                if rep > 0:
                    acc[i] += ms.value * 1e3 / args.reps
                lib.imagen_event_destroy(e0)
                lib.imagen_event_destroy(e1)
        total = sum(acc)
        print(f"\n=== stage {sidx}: {n} launches, {total / 1e3:.3f} ms per step (eager, event-timed)")
        by_kind = collections.defaultdict(lambda: [0, 0.0])
        by_class = collections.defaultdict(lambda: [0, 0.0, 0.0])
        items = []
        for (kind, p, label), us in zip(plan.ops, acc):
            by_kind[kind_name.get(kind, str(kind))][0] += 1
            by_kind[kind_name.get(kind, str(kind))][1] += us
            fl = by = 0.0
            desc = ""
            if kind == K_IGEMM:
                fl = bench.igemm_flops(p)
                cin = p.C1 + p.C2
                by = 2.0 * p.B * (p.H * p.W * cin + p.OH * p.OW * p.Cout * (2 if p.out_mode == 2 else 1)) + 2.0 * p.KH * p.KW * cin * p.Cout
                if p.res:
                    by += 2.0 * p.B * p.OH * p.OW * p.Cout
                if p.addend:
                    by += 2.0 * p.B * p.OH * p.OW * p.Cout
                desc = (f"{cin}->{p.Cout} k{p.KH} s{p.stride} @{p.H}x{p.W} B{p.B} cfg{p.cfg}{tab[p.cfg]} t{p.TH}x{p.TW}"
                        f"{' pro' if (p.pa or p.rs or p.ssq_a) else ''}")
                cls = re.sub(r"\d+", "#", label) + f" [{cin}->{p.Cout} k{p.KH} @{p.H}]"
            else:
                cls = re.sub(r"\d+", "#", label)
            c = by_class[cls]
            c[0] += 1
            c[1] += us
            c[2] += fl
            items.append((us, label, desc, fl, by))
            rows.append((sidx, label, kind_name.get(kind, str(kind)), desc, us, fl, by))
        print("-- by kind")
        for k, (cnt, us) in sorted(by_kind.items(), key=lambda kv: -kv[1][1]):
            print(f"  {k:16s} {cnt:4d} launches {us:9.1f} us {100 * us / total:5.1f}%")
        print("-- by label class")
        for k, (cnt, us, fl) in sorted(by_class.items(), key=lambda kv: -kv[1][1])[: args.top]:
            tf = f"{fl / us / 1e6:7.1f} TF" if fl else ""
            print(f"  {k:60s} {cnt:3d} x {us / cnt:7.1f} us = {us:8.1f} us {100 * us / total:5.1f}% {tf}")
        print("-- top launches")
        for us, label, desc, fl, by in sorted(items, key=lambda t: -t[0])[: args.top]:
            extra = f"{fl / us / 1e6:7.1f} TF {by / us / 1e3:7.0f} GB/s" if fl else ""
            print(f"  {label:34s} {us:7.1f} us {extra}  {desc}")
    if args.csv:
        with open(args.csv, "w") as f:
            f.write("stage,label,kind,desc,us,flops,bytes\n")
            for r in rows:
                f.write(",".join(str(x).replace(",", ";") for x in r) + "\n")


if __name__ == "__main__":
    main()
