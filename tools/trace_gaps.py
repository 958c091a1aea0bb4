"""Launch-gap analysis of a rocprofv3 --kernel-trace CSV: per kernel name the summed duration, and the idle time between
consecutive kernels on the GPU (graph replay), split by the kernel that FOLLOWS the gap.

    python tools/trace_gaps.py <kernel_trace.csv> [skip_first_n]
"""
import collections
import csv
import sys


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    skip = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows), key=lambda t: t[0])[skip:]
    busy = sum(e - s for s, e, _ in ev)
    span = ev[-1][1] - ev[0][0]
    gaps = collections.defaultdict(lambda: [0, 0])
    dur = collections.defaultdict(lambda: [0, 0])
    big = 0
    for (s0, e0, n0), (s1, e1, n1) in zip(ev, ev[1:]):
        g = s1 - e0
        if g > 200_000:      # host-side pause between graph launches / stages
            big += g
            continue
        k = n1.split("(")[0][-60:]
        gaps[k][0] += 1
        gaps[k][1] += max(g, 0)
    for s, e, n in ev:
        k = n.split("(")[0][-60:]
        dur[k][0] += 1
        dur[k][1] += e - s
    tot_gap = sum(v[1] for v in gaps.values())
    print(f"kernels {len(ev)}  span {span / 1e6:.2f} ms  busy {busy / 1e6:.2f} ms  small gaps {tot_gap / 1e6:.2f} ms  long pauses {big / 1e6:.2f} ms")
    print(f"mean gap {tot_gap / max(1, sum(v[0] for v in gaps.values())) / 1e3:.2f} us")
    print("-- by kernel: count, mean duration us, mean preceding gap us")
    for k, (c, d) in sorted(dur.items(), key=lambda kv: -kv[1][1])[:30]:
        gc, gg = gaps.get(k, (0, 0))
        print(f"  {k:62s} {c:6d} {d / c / 1e3:8.2f} {gg / max(gc, 1) / 1e3:7.2f}")


if __name__ == "__main__":
    main()
