#!/usr/bin/env python
"""Probe (GPU): what a fork / join inside the step graph could buy.  In every ResnetBlock with a res_conv, the 1x1 res_conv depends only on the block's
input, block1 -> block2 (-> GlobalContext) only on each other: the probe takes those launches out of the benchmark's real step plans and times
[block1, block2, gca*] followed by [res_conv] on one stream against the two lists on two streams between a fork and a join (events), per block and for
all blocks of a stage.  Prints one JSON line.  No product code changes: if the overlap is not there, the planner keeps one stream."""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from imagen_pytorch_amd import ops  # noqa: E402


def sub(plan, idxs, name):
    p = ops.Plan(name)
    for i in idxs:
        p.ops.append(plan.ops[i])
    return p


def timeit(fn, reps=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


def main():
    dev = torch.device("cuda", 0)
    imagen = bench.build_imagen(1000, dev)
    te = torch.randn(8, 256, 768, device=dev)
    imagen.sample(text_embeds=te, cond_scale=3.0, use_tqdm=False, seed=1, max_steps=2)
    torch.cuda.synchronize()
    s1, s2 = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
    out = {}
    for sidx, st in imagen._stages.items():
        plan = st["plan"]
        labels = [l for _, _, l in plan.ops]
        blocks = sorted({l[:-len(".res_conv")] for l in labels if l.endswith(".res_conv")})
        rows, tot_seq, tot_par = [], 0.0, 0.0
        for b in blocks:
            main_i = [i for i, l in enumerate(labels) if l.startswith(b + ".") and re.search(r"\.(block1|block2)(\.prep)?$|\.gca\.(partial|final)$", l)]
            res_i = [i for i, l in enumerate(labels) if l == b + ".res_conv"]
            if not main_i or not res_i:
                continue
            A, B = sub(plan, main_i, b + ".main"), sub(plan, res_i, b + ".res")
            ev1, ev2 = torch.cuda.Event(), torch.cuda.Event()

            def seq():
                with torch.cuda.stream(s1):
                    A.run(s1.cuda_stream)
                    B.run(s1.cuda_stream)

            def par():
                with torch.cuda.stream(s1):
                    ev1.record(s1)
                    s2.wait_event(ev1)
                    B.run(s2.cuda_stream)
                    A.run(s1.cuda_stream)
                    ev2.record(s2)
                    s1.wait_event(ev2)

            with torch.cuda.stream(s1):
                t_seq, t_par = timeit(seq), timeit(par)
                t_a, t_b = timeit(lambda: A.run(s1.cuda_stream)), timeit(lambda: B.run(s1.cuda_stream))
            rows.append(dict(block=b, main=[labels[i].split(".")[-1] for i in main_i], main_us=round(t_a, 1), res_us=round(t_b, 1), seq_us=round(t_seq, 1), fork_join_us=round(t_par, 1)))
            tot_seq += t_seq
            tot_par += t_par
        out[str(sidx[:3])] = dict(blocks=rows, seq_us=round(tot_seq, 1), fork_join_us=round(tot_par, 1))
    print(json.dumps(out))


if __name__ == "__main__":
    main()
