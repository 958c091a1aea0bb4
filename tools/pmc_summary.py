#!/usr/bin/env python
"""Summarise a rocprofv3 `--pmc` counter_collection.csv (one counter pass) per kernel symbol.

    python tools/pmc_summary.py <counter_collection.csv> <out.json>

Writes {kernel_name: {"launches": n, "<COUNTER>": average value per launch, ...}} so the small summary can be committed under
profiles/ and read by bench.py (roofline.traffic).  FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in KiB; on gfx950
FETCH_SIZE counts 128-B requests as 64 B, i.e. HALF the bytes of a wide coalesced read (MI355X_MICROARCH.md §HBM) — the
correction (x2) is applied by the consumer (bench.py), the raw averages are stored here.
"""
import collections
import csv
import json
import sys


def main(path, out):
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.defaultdict(lambda: collections.defaultdict(int))
    with open(path) as f:
        for r in csv.DictReader(f):
            k = r["Kernel_Name"]
            c = r["Counter_Name"]
            agg[k][c] += float(r["Counter_Value"])
            cnt[k][c] += 1
    res = {}
    for k in agg:
        res[k] = {"launches": max(cnt[k].values())}
        for c in agg[k]:
            res[k][c] = agg[k][c] / cnt[k][c]
    with open(out, "w") as f:
        json.dump(res, f, indent=1, sort_keys=True)
    top = sorted(res.items(), key=lambda kv: -sum(v for n, v in kv[1].items() if n != "launches") * kv[1]["launches"])[:8]
    for k, v in top:
        print(k[:80], v)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
