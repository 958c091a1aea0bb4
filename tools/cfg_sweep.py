#!/usr/bin/env python
"""In-loop sweep of igemm tile configurations: for the heaviest conv / GEMM layer classes of the benchmark's two denoisers, every launchable
candidate configuration is timed INSIDE the sampling loop (graph replay of the whole stage, one layer class overridden at a time), because
isolated-kernel timings have mispredicted the in-step ranking before (round 2: 128x128 tiles +3-9 % alone, -3 % in the model).

    python tools/cfg_sweep.py [--steps 40] [--top 8] [--out gpurun_out/cfg_sweep.json]

Output: per class the baseline pick and every candidate's ms per DDPM step of its stage; the candidates that beat the baseline by more
than --min-gain are collected as `picks` — the format of imagen-pytorch_amd/tuned_cfgs.json (ops.CFG_OVERRIDE), after a combined run has
confirmed them together."""
import argparse
import collections
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--top", type=int, default=8, help="layer classes per stage (by estimated time)")
    ap.add_argument("--max-cand", type=int, default=7)
    ap.add_argument("--min-gain", type=float, default=0.004, help="fraction of the stage's step time a candidate must save to be picked")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "cfg_sweep.json"))
    ap.add_argument("--budget-s", type=float, default=420.0)
    args = ap.parse_args()
    from imagen_pytorch_amd import _abi, ops

    dev = torch.device("cuda:0")
    ops.CFG_OVERRIDE.clear()
    imagen = bench.build_imagen(1000, dev)
    te = torch.randn(8, 256, 768, generator=torch.Generator().manual_seed(1234)).to(dev)
    K_IGEMM = _abi.ENUMS["IMAGEN_OP_IGEMM"]
    tab = ops.cfg_table()
    t_start = time.perf_counter()

    low = [None]

    def time_stage(stage: int, reps: int = 2) -> float:
        """ms per DDPM step of `stage` (0 | 1) with the current overrides (stages rebuilt)."""
        imagen._stages.clear()
        for u in imagen.unets:
            u.release_engines() if hasattr(u, "release_engines") else None
        kw = dict(text_embeds=te, cond_scale=3.0, use_tqdm=False)
        if stage == 0:
            run = lambda n, seed: imagen.sample(seed=seed, max_steps=n, stop_at_unet_number=1, **kw)
        else:
            if low[0] is None:
                low[0] = torch.rand(8, 3, 64, 64, device=dev)
            run = lambda n, seed: imagen.sample(seed=seed, max_steps=n, start_at_unet_number=2, start_image_or_video=low[0], **kw)
        run(2, 1)
        torch.cuda.synchronize()
        best = 1e9
        for r in range(reps):
            t0 = time.perf_counter()
            run(args.steps, 2 + r)
            torch.cuda.synchronize()
            best = min(best, (time.perf_counter() - t0) / args.steps * 1e3)
        return best

    def classes(stage: int):
        st = next(s for k, s in imagen._stages.items() if k[0] == stage)
        agg = collections.OrderedDict()
        for kind, p, label in st["plan"].ops:
            if kind != K_IGEMM:
                continue
            pro = bool(p.pa or p.rs or p.ssq_a or p.mu or p.ps or p.act_in)
            key = ops.layer_key(p.C1 + p.C2, p.Cout, p.KH, p.stride, p.OH, p.OW, p.B, pro)
            fl = bench.igemm_flops(p)
            by = bench.igemm_bytes(p)
            est = max(fl / 600e12, by / 3e12) * 1e6 + 4.0     # us: the rates the kernels reach today + a launch floor
            a = agg.setdefault(key, dict(count=0, est_us=0.0, cfg=(int(p.cfg), int(p.TH), int(p.TW)), G=tab[p.cfg][2], fam=tab[p.cfg][3],
                                         K=int(p.KH), stride=int(p.stride), OH=int(p.OH), OW=int(p.OW), raw=not pro and not p.x2 and p.C1 % 32 == 0,
                                         labels=[]))
            a["count"] += 1
            a["est_us"] += est
            a["labels"].append(label)
        return agg

    result = dict(steps=args.steps, stages=[])
    picks = {}
    for stage in (1, 0):
        base = time_stage(stage)
        base2 = time_stage(stage, reps=1)
        cls = classes(stage)
        print(f"stage {stage}: baseline {base:.4f} ms per step (again {base2:.4f}); {len(cls)} igemm classes", flush=True)
        rec = dict(stage=stage, baseline_ms=base, baseline_again_ms=base2, classes=[])
        order = sorted(cls.items(), key=lambda kv: -kv[1]["est_us"])[: args.top]
        for key, c in order:
            if c["fam"] in (3, 4):
                continue      # the streaming families are chosen by their own rules
            cands = []
            for i, (tp, bn, g, fam) in enumerate(tab):
                if g != c["G"] or fam in (3, 4) or i == c["cfg"][0]:
                    continue
                if fam == 2 and not (c["raw"] and c["K"] == 3 and c["stride"] == 1 and g == 4):
                    continue
                sh = ops.launchable_shapes(i, c["OH"], c["OW"], c["K"], c["K"], c["stride"])
                if sh:
                    cands.append((i, sh[0][2], sh[0][3]))
            # nearest tile sizes first
            tp0, bn0 = tab[c["cfg"][0]][0], tab[c["cfg"][0]][1]
            cands.sort(key=lambda ov: (abs(tab[ov[0]][0] * tab[ov[0]][1] - tp0 * bn0), tab[ov[0]][3]))
            crec = dict(key=key, count=c["count"], est_us=round(c["est_us"], 1), baseline_cfg=list(c["cfg"]), candidates=[])
            best = (base, None)
            for ov in cands[: args.max_cand]:
                if time.perf_counter() - t_start > args.budget_s:
                    break
                ops.CFG_OVERRIDE.clear()
                ops.CFG_OVERRIDE[key] = ov
                try:
                    ms = time_stage(stage, reps=1)
                except Exception as e:  # noqa: BLE001 — an unlaunchable combination must not end the sweep
                    crec["candidates"].append(dict(cfg=list(ov), error=f"{type(e).__name__}: {e}"[:200]))
                    continue
                crec["candidates"].append(dict(cfg=list(ov), tile=list(tab[ov[0]][:2]), family=tab[ov[0]][3], ms=round(ms, 4), delta_ms=round(ms - base, 4)))
                if ms < best[0]:
                    best = (ms, ov)
            ops.CFG_OVERRIDE.clear()
            if best[1] is not None and base - best[0] > args.min_gain * base:
                picks[key] = list(best[1])
                crec["picked"] = list(best[1])
            print(f"  {key:44s} x{c['count']:2d} base cfg{c['cfg']}: " + ", ".join(
                f"cfg{d['cfg'][0]}({d.get('tile')},f{d.get('family')}) {d.get('delta_ms', 'err'):+.3f}" if "ms" in d else f"cfg{d['cfg'][0]} err" for d in crec["candidates"]),
                flush=True)
            rec["classes"].append(crec)
        # all picks of this stage together
        ops.CFG_OVERRIDE.clear()
        ops.CFG_OVERRIDE.update({k: tuple(v) for k, v in picks.items()})
        rec["combined_ms"] = time_stage(stage) if picks else base
        print(f"stage {stage}: all picks together {rec['combined_ms']:.4f} ms per step (baseline {base:.4f})", flush=True)
        result["stages"].append(rec)
    result["picks"] = picks
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    json.dump(result, open(args.out, "w"), indent=1)
    print(json.dumps(dict(picks=picks)))


if __name__ == "__main__":
    main()
