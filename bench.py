#!/usr/bin/env python
"""bench.py — images/sec of the Imagen 64->256 cascade on MI355X (BASELINE.json metric), one process per GPU.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A "step" is ONE full pass of the hot path over one batch: `Imagen.sample()` of the README cascade (unet1 @64^2 ->
unet2 @256^2), 1000 DDPM steps per stage, classifier-free guidance 3.0, batch 8 PER GPU (weak scaling: each rank
samples its own 8 prompts and the final images are all-gathered over RCCL).  Inputs are synthetic and resident in
HBM before the timed region: random-init weights (final_conv ~ N(0, 0.05^2)), random text_embeds,
in-kernel Philox noise.  Rank 0 prints one JSON line.

Batch schedule (--mode): the K timed batches are scheduled `lanes` (default; --lanes 3 whole cascades side by side, one HIP stream +
hipGraph set each — the 64^2 stage and the 32^2/64^2 levels of the 256^2 stage launch a few dozen workgroups per kernel and are
latency-bound at batch 8, so independent batches fill the idle CUs), `pipeline` (cascade stages overlapped across batches,
Imagen.sample_pipelined) or `sequential` (one sample() call after the other — also measured once after the timed region and
reported under "sequential").  Every batch runs the full cascade at batch 8; nothing is shared between batches but the weights.

Extra legs (rank 0, N = 1 only):
  roofline     — HIP-event timing, on the launch stream, of every implicit-GEMM launch of one denoiser step of each stage, grouped
                 by kernel symbol (tile configuration, taps, epilogue / prologue instantiation) and by bound (algorithmic FLOP/byte vs the 312 FLOP/B ridge).  "roofline" is the dominant
                 group overall, "roofline_other_bound" the dominant group under the other roof: achieved = algorithmic FLOPs
                 (2*MACs) or bytes (inputs + output + weights + epilogue operand) / time, vs 2.5 PFLOP/s dense fp16 MFMA or 8 TB/s
                 HBM (/opt/skills/guides/MI355X_MICROARCH.md).  `traffic` (HBM bytes per launch) and `mfma_busy_frac` come from
                 three `rocprofv3 --pmc` passes (FETCH_SIZE | WRITE_SIZE | SQ MFMA-busy) that THIS invocation runs over a short
                 sampling run in a child process (tools/graph_profile.py), joined with the plan by dispatch order.
  cpu_baseline — the reference itself on the host cores where its tree exists (kind "reference"; the build container only), else
                 the CPU oracle (oracle/: fp32 torch restatement of the reference path, "port") for ONE DDPM step per stage at
                 batch 8 (2 CFG forwards each), linearly extrapolated to 1000 steps.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

README_U1 = dict(dim=32, cond_dim=512, dim_mults=(1, 2, 4, 8), num_resnet_blocks=3, layer_attns=(False, True, True, True),
                 layer_cross_attns=(False, True, True, True))
README_U2 = dict(dim=32, cond_dim=512, dim_mults=(1, 2, 4, 8), num_resnet_blocks=(2, 4, 8, 8), layer_attns=(False, False, False, True),
                 layer_cross_attns=(False, False, False, True))
FLOPS_PER_IMAGE_REFERENCE = 287.3e12   # 2000 * (12.71 + 130.94) GF, as the reference executes the path
MFMA_PEAK_TFLOPS = 2500.0              # dense fp16/bf16, MI355X_MICROARCH.md
LANES_DEFAULT = 6


def build_imagen(timesteps: int, device):
    from imagen_pytorch_amd import Imagen, Unet

    torch.manual_seed(0)
    u1, u2 = Unet(**README_U1), Unet(**README_U2)
    imagen = Imagen((u1, u2), image_sizes=(64, 256), timesteps=timesteps, cond_drop_prob=0.1)
    for u in imagen.unets:
        torch.nn.init.normal_(u.final_conv.weight, std=0.05)
        torch.nn.init.normal_(u.final_conv.bias, std=0.05)
    return imagen.to(device).eval()


# igemm tile configuration id -> template arguments <MI, NI, WM, WN, G> (csrc/igemm.hip kCfgs), to match rocprof kernel names
CFG_TEMPLATE = {0: (2, 1, 4, 1, 4), 1: (4, 1, 1, 4, 4), 2: (4, 1, 2, 2, 4), 3: (2, 1, 1, 4, 4), 4: (1, 1, 4, 1, 4), 5: (2, 1, 4, 1, 1),
                6: (1, 1, 2, 2, 4), 7: (1, 2, 2, 2, 1), 8: (1, 1, 2, 2, 1), 9: (1, 1, 4, 1, 1), 10: (2, 1, 1, 4, 16), 11: (4, 1, 1, 4, 8),
                12: (1, 1, 2, 2, 16), 13: (1, 1, 4, 1, 8), 14: (2, 1, 1, 4, 8), 15: (1, 1, 2, 2, 8)}


def igemm_flops(p) -> float:
    """Algorithmic FLOPs of one igemm launch: 2 * output pixels * Cout * (KH*KW*Cin)."""
    return 2.0 * p.B * p.OH * p.OW * p.Cout * p.KH * p.KW * (p.C1 + p.C2)


HBM_PEAK_GBS = 8000.0                  # HBM3E, MI355X_MICROARCH.md (achievable copy rate ~6300)
RIDGE = MFMA_PEAK_TFLOPS * 1e12 / (HBM_PEAK_GBS * 1e9)   # FLOP per byte above which a launch is MFMA-bound (312)


def igemm_bytes(p) -> float:
    """Algorithmic HBM bytes of one igemm launch: inputs once + output once (fp32 for the NCHW image) + packed weights +
    the epilogue's addend / residual operand (per-unit figures)."""
    cin = p.C1 + p.C2
    by = 2.0 * p.B * (p.H * p.W * cin + p.OH * p.OW * p.Cout * (2 if p.out_mode == 2 else 1)) + 2.0 * p.KH * p.KW * cin * p.Cout
    return by + 2.0 * p.B * p.OH * p.OW * p.Cout * (bool(p.res) + bool(p.addend))


def pmc_traffic_leg(log, budget_s: float = 300.0):
    """HBM traffic / MFMA-busy counters of every igemm launch of the denoiser steps, measured NOW: separate `rocprofv3 --pmc`
    passes (FETCH_SIZE | WRITE_SIZE | SQ MFMA-busy) over a short sampling run in a child process (tools/graph_profile.py run),
    joined with the plan by dispatch order.  Returns {(stage index, label): {counter: value per launch}} or None."""
    import shutil
    import subprocess
    import tempfile
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import graph_profile

    rocprof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(rocprof):
        return None
    out = {}
    passes = (["FETCH_SIZE"], ["WRITE_SIZE"], ["SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "GRBM_GUI_ACTIVE"])
    t_leg = time.perf_counter()
    for counters in passes:
        left = budget_s - (time.perf_counter() - t_leg)      # the three passes share one wall-clock budget (a pass takes ~40 s)
        if left < 30:
            log(f"pmc pass {counters} skipped: {budget_s:.0f} s counter budget spent")
            continue
        tmp = tempfile.mkdtemp(prefix="imagen_pmc_", dir="/tmp")
        try:
            plan_path = os.path.join(tmp, "plan.json")
            cmd = [rocprof, "--pmc", *counters, "--output-format", "csv", "-d", os.path.join(tmp, "out"), "--",
                   sys.executable, os.path.join(ROOT, "tools", "graph_profile.py"), "run", "--steps", "5", "--plan-out", plan_path]
            env = dict(os.environ, TMPDIR="/tmp")
            r = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=min(150, left))
            csvs = [os.path.join(d, f) for d, _, fs in os.walk(os.path.join(tmp, "out")) for f in fs if f.endswith("counter_collection.csv")]
            if r.returncode != 0 or not csvs or not os.path.exists(plan_path):
                log(f"pmc pass {counters}: rocprofv3 failed (rc {r.returncode}): {r.stdout.decode(errors='replace')[-300:]}")
                continue
            plan = json.load(open(plan_path))
            for si, ops_ in enumerate(graph_profile.pmc_join(csvs[0], plan)):
                for o in ops_ or []:
                    if o["kind"] == "igemm":
                        out.setdefault((si, o["label"]), {}).update({c: o[c] for c in counters if c in o})
            log(f"pmc pass {counters} done")
        except Exception as e:  # noqa: BLE001 — the leg is optional
            log(f"pmc pass {counters} failed: {e}")
        finally:
            shutil.rmtree(tmp, ignore_errors=True)
    return out or None


def calibration_leg(device):
    """Self-anchoring numbers measured in THIS process: the GPU's current shader / memory clocks (rocm-smi, best effort), a fixed 1 GiB
    device copy (GB/s of read + written bytes) and a fixed dense fp16 MFMA loop on every CU (TFLOP/s) — imagen_probe_copy /
    imagen_probe_mfma of the kernel library, HIP-event timed.  Boxes of the pool differ by up to +-20 % on the sampling workload; these
    two figures let one round's line be normalised against another's."""
    import ctypes
    import shutil
    import subprocess
    from imagen_pytorch_amd import _abi, ops

    lib = _abi.load_library()
    out = {}
    n = 1 << 30
    src = torch.empty(n, dtype=torch.uint8, device=device).fill_(1)
    dst = torch.empty(n, dtype=torch.uint8, device=device)
    sink = torch.zeros(4, dtype=torch.float32, device=device)
    torch.cuda.synchronize()
    h = ops.current_stream_handle()
    v = ctypes.c_float()
    _abi.check(lib.imagen_probe_copy(dst.data_ptr(), src.data_ptr(), n, 5, h, ctypes.byref(v)), "probe_copy")
    out["copy_1GiB_GBs"] = round(v.value, 1)
    _abi.check(lib.imagen_probe_mfma(20000, 3, sink.data_ptr(), h, ctypes.byref(v)), "probe_mfma")
    out["mfma_f16_loop_TFLOPs"] = round(v.value, 1)
    out["mfma_f16_loop_frac_of_peak"] = round(v.value / MFMA_PEAK_TFLOPS, 4)
    del src, dst
    # latency side (round 5): the path is a chain of dependent round trips and dependent launches, and boxes that agree on the two throughput
    # figures above differ by 20 % on it.  One lane chases a random cycle of 128-byte nodes through a working set that lives in the XCD's L2
    # (1 MiB), in the Infinity Cache (64 MiB) or in HBM (2 GiB): ns per dependent load; and a captured chain of 300 dependent one-wave launches
    # (three kernel symbols in rotation): us per dependent launch inside a hipGraph.
    word = torch.zeros(4, dtype=torch.int32, device=device)
    lat = {}
    for name, nbytes, hops in (("l2_1MiB", 1 << 20, 20000), ("infinity_cache_64MiB", 64 << 20, 20000), ("hbm_2GiB", 2 << 30, 20000)):
        nodes = nbytes // 128
        perm = torch.randperm(nodes, device=device, generator=torch.Generator(device=device).manual_seed(7))
        chain = torch.zeros(nodes, 32, dtype=torch.int32, device=device)
        chain[perm, 0] = perm.roll(-1).to(torch.int32)       # node perm[k] -> perm[k + 1]: one cycle over all nodes
        del perm
        torch.cuda.synchronize()
        _abi.check(lib.imagen_probe_latency(chain.data_ptr(), hops, word.data_ptr(), h, ctypes.byref(v)), "probe_latency")
        lat[name] = round(v.value, 1)
        del chain
    out["dependent_load_ns"] = lat
    # ... and the same HBM chase while a copy kernel on a second stream keeps every memory channel busy (round 6): the idle chase reads the same on
    # boxes whose sampling step differs by 14 % — what the denoiser's dependent round trips pay is the LOADED latency
    try:
        import threading
        nbytes, hops = 2 << 30, 6000
        nodes = nbytes // 128
        perm = torch.randperm(nodes, device=device, generator=torch.Generator(device=device).manual_seed(11))
        chain = torch.zeros(nodes, 32, dtype=torch.int32, device=device)
        chain[perm, 0] = perm.roll(-1).to(torch.int32)
        del perm
        src = torch.empty(n, dtype=torch.uint8, device=device).fill_(1)
        dst = torch.empty(n, dtype=torch.uint8, device=device)
        side = torch.cuda.Stream(device=device)
        torch.cuda.synchronize()
        v2, rc = ctypes.c_float(), []
        th = threading.Thread(target=lambda: rc.append(lib.imagen_probe_copy(dst.data_ptr(), src.data_ptr(), n, 120, side.cuda_stream, ctypes.byref(v2))))
        th.start()
        time.sleep(0.003)                                    # (the copy launches are queued by now: ~50 ms of saturated HBM)
        _abi.check(lib.imagen_probe_latency(chain.data_ptr(), hops, word.data_ptr(), h, ctypes.byref(v)), "probe_latency (loaded)")
        th.join()
        torch.cuda.synchronize()
        out["dependent_load_ns_under_copy"] = {"hbm_2GiB": round(v.value, 1), "copy_GBs_meanwhile": round(v2.value, 1)}
        del chain, src, dst
    except Exception as e:  # noqa: BLE001 — a probe must never cost the line
        out["dependent_load_ns_under_copy"] = {"error": f"{type(e).__name__}: {e}"}
    _abi.check(lib.imagen_probe_launch_chain(300, 20, word.data_ptr(), h, ctypes.byref(v)), "probe_launch_chain")
    out["graph_dependent_launch_us"] = round(v.value, 3)
    out.update(smi_sample(("sclk", "mclk")))
    return out


def _mhz(text):
    """'1200Mhz' -> 1200.0 (None if unreadable)."""
    import re
    m = re.search(r"([0-9.]+)\s*mhz", str(text or ""), flags=re.I)
    return float(m.group(1)) if m else None


def smi_sample(clocks=("sclk", "mclk", "fclk", "socclk"), extra=False):
    """Best-effort reading of rocm-smi: the named clocks ("sclk clock speed:": "(2400Mhz)" -> {"sclk": "2400Mhz"}) and, with `extra`, every power /
    temperature field verbatim.  Idle readings say little (every box of the pool reports the same levels); `sequential.rocm_smi_under_load` is the
    same reading taken two seconds into a sampling pass."""
    import shutil
    import subprocess

    smi = shutil.which("rocm-smi") or "/opt/rocm/bin/rocm-smi"
    out = {}
    try:
        cmd = [smi, "--showclocks", "--json"] + (["--showpower", "--showtemp"] if extra else [])
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=20)
        card = next(iter(json.loads(r.stdout.decode()).values()))
        for k, val in card.items():
            kl = k.lower()
            for name in clocks:
                if kl.startswith(name) and "speed" in kl:
                    out[name] = str(val).strip("()")
            if extra and ("power" in kl or "temperature" in kl):
                out[k.strip(": ")] = str(val)
    except Exception as e:  # noqa: BLE001 — best effort
        out["clocks_error"] = f"{type(e).__name__}: {e}"
    return out


def stage_replay_leg(imagen, reps: int = 50):
    """In-graph time of one denoiser + sampler step of each stage: HIP events on the launch stream around `reps` back-to-back replays of the
    stage's captured per-timestep graph (the default lane's stages: what a sequential sample() replays 1000 times).  The step counter runs
    past the schedule during the probe (the tables clamp it); the next sample() call resets the stage's state anyway."""
    import ctypes
    from imagen_pytorch_amd import _abi

    lib = _abi.load_library()
    out = {}
    torch.cuda.synchronize()
    for key, st in imagen._stages.items():
        if key[-1] != 0 or st.get("graph") is None:      # lane 0 = the lane sequential sample() calls run on
            continue
        h = st["graph"].stream.cuda_stream               # the stream the graph was captured on and replays on
        st["step_ptr"].zero_()
        torch.cuda.synchronize()
        st["graph"].launch()
        e0, e1 = ctypes.c_void_p(), ctypes.c_void_p()
        lib.imagen_event_create(ctypes.byref(e0))
        lib.imagen_event_create(ctypes.byref(e1))
        lib.imagen_event_record(e0, h)
        for _ in range(reps):
            st["graph"].launch()
        lib.imagen_event_record(e1, h)
        torch.cuda.synchronize()
        ms = ctypes.c_float()
        lib.imagen_event_elapsed_ms(e0, e1, ctypes.byref(ms))
        lib.imagen_event_destroy(e0)
        lib.imagen_event_destroy(e1)
        out[f"stage{key[0]}_{st['S']}px"] = {"ms_per_step": round(ms.value / reps, 4), "launches_per_step": len(st["plan"].ops)}
    return out


def roofline_leg(imagen, batch: int, device, pmc=None):
    """Event-time every igemm launch of one denoiser step of both stages (eager, same stream), grouped by kernel symbol — the template
    instantiation a launch takes, i.e. the rows of `rocprofv3 --stats` (round 3 grouped by tile configuration, which merges the plain- and
    generic-epilogue instantiations of one configuration into a group no rocprof row corresponds to).  Two roofs are reported: the symbol
    with the largest total time among the MFMA-bound launches (algorithmic FLOP/byte above the ridge) and the one among the HBM-bound."""
    import ctypes
    from imagen_pytorch_amd import _abi, ops

    lib = _abi.load_library()
    K_IGEMM = _abi.ENUMS["IMAGEN_OP_IGEMM"]
    groups = {}
    tab_fam = [c[3] for c in ops.cfg_table()]
    stream = torch.cuda.current_stream()
    h = stream.cuda_stream
    seen = set()
    stage_no = -1
    for key, st in imagen._stages.items():
        if key[:3] in seen or key[1] != batch:      # one engine per (stage, batch, size): lanes hold copies of the same plan; the merged-requests
            continue                                # leg's batch-24 stages are not the metric's workload
        seen.add(key[:3])
        stage_no = key[0]
        plan = st["plan"]
        st["step_ptr"].zero_()
        evs = []
        for kind, struct, label in plan.ops:
            if kind == K_IGEMM:
                e0, e1 = ctypes.c_void_p(), ctypes.c_void_p()
                lib.imagen_event_create(ctypes.byref(e0))
                lib.imagen_event_create(ctypes.byref(e1))
                lib.imagen_event_record(e0, h)
                _abi.check(lib.imagen_launch(kind, ctypes.addressof(struct), ctypes.sizeof(struct), h))
                lib.imagen_event_record(e1, h)
                # the kernel SYMBOL a launch takes (what `rocprofv3 --stats` lists, profiles/rNN_kernel_stats.csv): tile configuration +
                # taps (the unrolled k-loop instantiation) + generic / plain epilogue + (family 6) the prologue instantiation
                fam = tab_fam[struct.cfg]
                gen = fam in (0, 2, 3, 5) and bool(struct.act_out or struct.out_mode or struct.addend or struct.res)
                sym = (struct.cfg, struct.KH * struct.KW, gen, fam in (3, 6) and bool(struct.ssq_a))
                evs.append((sym, igemm_flops(struct), igemm_bytes(struct), (stage_no, label), e0, e1))
            else:
                _abi.check(lib.imagen_launch(kind, ctypes.addressof(struct), ctypes.sizeof(struct), h))
        torch.cuda.synchronize()
        for cfg, fl, by, ident, e0, e1 in evs:
            ms = ctypes.c_float()
            lib.imagen_event_elapsed_ms(e0, e1, ctypes.byref(ms))
            bound = "mfma" if fl / by >= RIDGE else "hbm"
            d = groups.setdefault((bound, cfg), dict(n=0, fl=0.0, by=0.0, sec=0.0, ids=[]))
            d["n"] += 1
            d["fl"] += fl
            d["by"] += by
            d["sec"] += ms.value * 1e-3
            d["ids"].append(ident)
            lib.imagen_event_destroy(e0)
            lib.imagen_event_destroy(e1)
    tab = ops.cfg_table()
    fam_name = {0: "igemm_kernel", 2: "conv_dma_kernel", 3: "conv_stream_kernel", 4: "conv_pw_kernel", 5: "conv_big_kernel", 6: "conv_pro_kernel", 7: "conv_gemm_kernel", 8: "conv_small_kernel"}

    def describe(bound):
        cands = {k: v for k, v in groups.items() if k[0] == bound}
        if not cands:
            return None
        (_, sym), g = max(cands.items(), key=lambda kv: kv[1]["sec"])
        cfg, taps, gen, pro = sym
        traffic = mfma_util = None
        if pmc:
            rows = [pmc[i] for i in g["ids"] if i in pmc]
            if rows and all("FETCH_SIZE" in r and "WRITE_SIZE" in r for r in rows):
                # KiB units; FETCH_SIZE doubled: gfx950 counts a 128-byte request as 64 bytes (MI355X_MICROARCH.md, HBM section)
                traffic = round(sum((2.0 * r["FETCH_SIZE"] + r["WRITE_SIZE"]) * 1024.0 for r in rows) / len(rows))
            if rows and all("SQ_VALU_MFMA_BUSY_CYCLES" in r and r.get("GRBM_GUI_ACTIVE") for r in rows):
                # MFMA-busy cycles summed over the 1024 SIMDs / (kernel cycles x 1024); GRBM_GUI_ACTIVE is summed over the 8 XCDs
                mfma_util = round(sum(r["SQ_VALU_MFMA_BUSY_CYCLES"] / (r["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0) for r in rows) / len(rows), 4)
        if bound == "mfma":
            ach, peak, unit = g["fl"] / g["sec"] / 1e12, MFMA_PEAK_TFLOPS, "TFLOP/s"
        else:
            ach, peak, unit = g["by"] / g["sec"] / 1e9, HBM_PEAK_GBS, "GB/s"
        return {"bound": bound, "achieved": round(ach, 1), "peak": peak, "unit": unit, "frac": round(ach / peak, 4), "traffic": traffic,
                "traffic_source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes run by this bench.py invocation" if traffic is not None else None,
                "mfma_busy_frac": mfma_util,
                "kernel": f"{fam_name.get(tab[cfg][3], 'igemm family ' + str(tab[cfg][3]))} cfg {cfg} ({tab[cfg][0]} px x {tab[cfg][1]} cout tile, G={tab[cfg][2]}), "
                          f"{taps}-tap instantiation, {'generic' if gen else 'plain'} epilogue{', with the Block prologue' if pro else ''}",
                "launches_per_denoiser_step_pair": g["n"], "avg_launch_us": round(g["sec"] / g["n"] * 1e6, 2),
                "avg_launch_gflop": round(g["fl"] / g["n"] / 1e9, 3), "avg_launch_algorithmic_bytes": round(g["by"] / g["n"]),
                "timing": "HIP events around each eager launch of one denoiser step per stage (cold caches, as inside the sampling loop)"}

    mf, hb = describe("mfma"), describe("hbm")
    tsum = lambda b: sum(v["sec"] for k, v in groups.items() if k[0] == b)
    # "roofline" = the single (bound, kernel symbol) group with the largest total time; the other bound's dominant group beside it
    top_bound = max(groups.items(), key=lambda kv: kv[1]["sec"])[0][0]
    main, other = (mf, hb) if top_bound == "mfma" else (hb, mf)
    main = dict(main)
    main["share_of_igemm_time"] = {"mfma_bound_launches": round(tsum("mfma") / (tsum("mfma") + tsum("hbm")), 3),
                                   "hbm_bound_launches": round(tsum("hbm") / (tsum("mfma") + tsum("hbm")), 3)}
    main["per_kernel"] = {f"{b}:cfg{c[0]}:k{c[1]}{':gen' if c[2] else ''}{':pro' if c[3] else ''}":
                          {"launches": v["n"], "tflops": round(v["fl"] / max(v["sec"], 1e-12) / 1e12, 1),
                           "gbytes_per_s": round(v["by"] / max(v["sec"], 1e-12) / 1e9, 1), "ms_total": round(v["sec"] * 1e3, 3)}
                          for (b, c), v in sorted(groups.items())}
    return main, other


def cpu_reference_leg(batch: int, steps: int = 2):
    """The LIVE reference (imagen_pytorch at /root/reference, imported through oracle/ref_shim.py) sampling the same cascade on the
    host cores: `steps` DDPM steps per stage at batch `batch` with CFG 3, extrapolated to 1000.  Only where the reference tree
    exists (the build container) — it cannot travel to the GPU box, which times the oracle port instead."""
    from oracle import ref_shim

    ip = ref_shim.load_reference("imagen_pytorch")
    threads = min(os.cpu_count() or 1, 64)
    torch.set_num_threads(threads)
    torch.manual_seed(0)
    u1, u2 = ip.Unet(**README_U1), ip.Unet(**README_U2)
    imagen = ip.Imagen((u1, u2), image_sizes=(64, 256), timesteps=steps, cond_drop_prob=0.1).eval()
    te = torch.randn(batch, 256, 768)
    t0 = time.time()
    with torch.no_grad():
        out = imagen.sample(text_embeds=te, cond_scale=3.0, use_tqdm=False)
    dt = time.time() - t0
    assert tuple(out.shape) == (batch, 3, 256, 256)
    T = 1000
    return {"value": batch / (dt / steps * T), "unit": "images/s", "cores": threads, "kind": "reference",
            "sample": f"lucidrains/imagen-pytorch Imagen.sample on CPU (fp32), README unet1+unet2 64->256, batch {batch}, CFG 3.0, {steps} DDPM steps "
                      f"per stage in {dt:.1f} s, linearly extrapolated to {T} steps/stage"}


def cpu_baseline_leg(imagen, batch: int):
    """The reference itself where its tree is present, else the oracle ("port" of the reference path, fp32 torch CPU) for one DDPM
    step per stage, extrapolated."""
    try:
        from oracle import ref_shim
        if ref_shim.reference_available():
            return cpu_reference_leg(batch)
    except Exception as e:  # noqa: BLE001 — fall back to the port
        print(f"[bench] live reference unavailable ({e}); timing the oracle port", file=sys.stderr)
    from oracle import sampler_oracle as so
    from oracle import unet_oracle as uo

    if imagen is None:
        imagen = build_imagen(1000, "cpu")
    threads = min(os.cpu_count() or 1, 64)   # oneDNN convs stop scaling (and can thrash) far beyond this on many-core hosts
    torch.set_num_threads(threads)
    torch.manual_seed(0)
    te = torch.randn(batch, 256, 768)
    mask = torch.ones(batch, 256, dtype=torch.bool)
    per_step = []
    for idx, (u, kw, S) in enumerate(zip(imagen.unets, (README_U1, {**README_U2, "lowres_cond": True}), (64, 256))):
        sd = {k: v.detach().float().cpu() for k, v in u.state_dict().items()}
        x = torch.randn(batch, 3, S, S)
        t = torch.full((batch,), 0.5)
        extra = dict(lowres_cond_img=torch.randn(batch, 3, S, S), lowres_noise_times=torch.full((batch,), 1.0)) if idx else {}
        t0 = time.time()
        with torch.no_grad():
            pred = uo.unet_forward_with_cond_scale(sd, kw, x, t, cond_scale=3.0, text_embeds=te, text_mask=mask, **extra)
            so.ddpm_step(x, pred, torch.full((batch,), 0.5), torch.full((batch,), 0.499), torch.randn_like(x), "cosine")
        per_step.append(time.time() - t0)
    T = 1000
    value = batch / (T * sum(per_step))
    return {"value": value, "unit": "images/s", "cores": threads, "kind": "port",
            "sample": f"oracle (fp32 torch CPU restatement of the reference path), 1 DDPM step per stage at batch {batch} with CFG "
                      f"(u1 {per_step[0]:.2f} s, u2 {per_step[1]:.2f} s), linearly extrapolated to {T} steps/stage"}


# BASELINE.json configs beside the headline (C3), informational legs printing the same JSON shape (`--config c2 | c4 | c5`): algorithmic
# FLOPs per unit and the per-unit ceilings are SURVEY.md §8d's
OTHER_CONFIGS = {
    "c2": dict(flops_per_unit=186.1e12, ceiling=(12.47, 14.0), unit="images/s",
               workload="C2: base Unet dim 128 (README unet1 kwargs x4 channels), 64^2, 1000 DDPM steps, CFG 3.0, batch 8 (BASELINE says bf16; this path "
                        "computes in fp16 with fp32 accumulation: same MFMA rate, 8x finer mantissa)"),
    "c4": dict(flops_per_unit=18.1e12, ceiling=(79.4, 140.8), unit="images/s",
               workload="C4 (one GPU's shard): ElucidatedImagen, README unet1 + unet2, 64 -> 256, 32 Karras steps (63 denoiser evaluations per stage), "
                        "CFG 3.0, 4 images per GPU (= 32 over 8 GPUs)"),
    "c5": dict(flops_per_unit=163.0e12, ceiling=(15.3, 15.3), unit="clips/s",
               workload="C5: Imagen-Video Unet3D(dim 64, dim_mults (1, 2, 4, 8)), one 16 x 64 x 64 clip, 250 DDPM steps, CFG 3.0"),
}


def other_config_leg(name: str, steps: int, reps: int):
    """One of BASELINE.json's other configs on one GPU: `steps` sampling steps timed (graph replays, after a 2-step warm-up that packs the
    weights and captures the graphs), linearly extrapolated to the config's full schedule where `steps` is smaller (said in the record).
    Prints the bench JSON shape with the SURVEY ceiling and the fraction of the dense MFMA peak beside it."""
    from imagen_pytorch_amd import ElucidatedImagen, Imagen, Unet, Unet3D

    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    torch.manual_seed(0)
    meta = OTHER_CONFIGS[name]
    kw = dict(cond_scale=3.0, use_tqdm=False)
    if name == "c2":
        model = Imagen((Unet(**dict(README_U1, dim=128)),), image_sizes=(64,), timesteps=1000, cond_drop_prob=0.1)
        B, T, per = 8, 1000, "image"
    elif name == "c4":
        model = ElucidatedImagen((Unet(**README_U1), Unet(**README_U2)), image_sizes=(64, 256), num_sample_steps=32, cond_drop_prob=0.1)
        B, T, per, steps = 4, 32, "image", 32
    else:
        model = Imagen((Unet3D(dim=64, dim_mults=(1, 2, 4, 8)),), image_sizes=(64,), timesteps=250, cond_drop_prob=0.1)
        B, T, per = 1, 250, "clip"
        kw["video_frames"] = 16
    for u in model.unets:
        torch.nn.init.normal_(u.final_conv.weight, std=0.05)
        torch.nn.init.normal_(u.final_conv.bias, std=0.05)
    model = model.to(dev).eval()
    te = torch.randn(B, 256, 768, generator=torch.Generator().manual_seed(1234)).to(dev)
    steps = min(steps, T)
    run_kw = dict(kw) if name == "c4" else dict(kw, max_steps=steps)
    model.sample(text_embeds=te, seed=1, **(kw if name == "c4" else dict(kw, max_steps=2)))
    torch.cuda.synchronize()
    best = 1e30
    for r in range(reps):
        t0 = time.perf_counter()
        out = model.sample(text_embeds=te, seed=2 + r, **run_kw)
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    assert torch.isfinite(out).all()
    full = best * (T / steps)          # seconds per batch at the config's full schedule
    value = B / full
    launches = {f"stage{k[0]}": len(st["plan"].ops) for k, st in getattr(model, "_stages", {}).items()}
    rec = {"metric": f"{meta['unit']} ({name.upper()}, BASELINE.json configs)", "value": round(value, 4), "unit": meta["unit"], "n_gpus": 1, "steps": reps, "warmup": 1,
           "ms_per_step": round(full * 1e3, 2), "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f16", "data": "synthetic",
           "config": {"workload": meta["workload"], "global_batch": B, "timed_sampling_steps": steps, "schedule_steps": T,
                      "extrapolated": steps < T, "launches_per_step": launches},
           "ms_per_sampling_step": round(best / steps * 1e3, 4),
           "ceiling_per_unit": {"value": list(meta["ceiling"]), "unit": meta["unit"], "source": "SURVEY.md §8d: max(MFMA, HBM) per unit at 8 | 6.29 TB/s .. 100 % MFMA"},
           "path_tflops_reference_count": round(value * meta["flops_per_unit"] / 1e12, 1),
           "path_frac_of_mfma_peak": round(value * meta["flops_per_unit"] / 1e12 / MFMA_PEAK_TFLOPS, 4),
           "note": f"one request at a time (batch {B}), best of {reps}; per {per}: {meta['flops_per_unit'] / 1e12:.1f} TFLOP by the reference's own count"}
    print(json.dumps(rec), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=8, help="images per GPU (BASELINE: 8)")
    ap.add_argument("--timesteps", type=int, default=1000, help="DDPM steps per stage (BASELINE: 1000); other values are NOT the headline metric")
    ap.add_argument("--mode", choices=("sequential", "pipeline", "lanes"), default=os.environ.get("IMAGEN_BENCH_MODE", "lanes"),
                    help="how successive batches are scheduled on the GPU: one sample() after the other | cascade stages overlapped across "
                         "batches (Imagen.sample_pipelined) | --lanes whole cascades side by side (one thread + stream each)")
    ap.add_argument("--lanes", type=int, default=int(os.environ.get("IMAGEN_BENCH_LANES", "0")),
                    help="concurrent cascades in --mode lanes; 0 = min(6, --steps): throughput still rises from 3 to 6 lanes (round 3, call C: "
                         "8.90 / 8.00 / 7.50 ms per DDPM step pair at 3 / 4 / 6 lanes on one box); batches are handed out dynamically")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-merged", action="store_true", help="skip the informational merged-requests leg (six requests as two merged batches of 24)")
    ap.add_argument("--no-pmc", action="store_true", help="skip the rocprofv3 --pmc child passes (roofline.traffic / mfma_busy_frac = null)")
    ap.add_argument("--cpu-baseline-only", action="store_true", help="time the CPU baseline leg only (no GPU needed) and print its JSON")
    ap.add_argument("--config", choices=("c3", "c2", "c4", "c5"), default="c3",
                    help="c3 (default): the headline metric; c2 / c4 / c5: BASELINE.json's other configs as informational one-GPU legs (same JSON shape)")
    ap.add_argument("--config-steps", type=int, default=50, help="sampling steps timed by the c2 / c5 legs (extrapolated to the full schedule)")
    args = ap.parse_args()
    if args.config != "c3":
        other_config_leg(args.config, args.config_steps, max(1, args.steps))
        return
    if args.cpu_baseline_only:
        print(json.dumps(cpu_baseline_leg(None, args.batch)), flush=True)
        return

    if args.lanes <= 0:
        args.lanes = max(1, min(LANES_DEFAULT, args.steps))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}"
    device = torch.device("cuda", local_rank)
    torch.cuda.set_device(device)

    from imagen_pytorch_amd.distributed import all_gather_images

    def log(msg):
        if rank == 0:
            print(f"[bench +{time.perf_counter() - t_start:7.1f}s] {msg}", file=sys.stderr, flush=True)

    t_start = time.perf_counter()
    imagen = build_imagen(args.timesteps, device)
    log("model built")
    B = args.batch
    gen = torch.Generator().manual_seed(1234 + rank)
    text_embeds = torch.randn(B, 256, 768, generator=gen).to(device)

    def one_pass(i):
        img = imagen.sample(text_embeds=text_embeds, cond_scale=3.0, use_tqdm=False, seed=1000 + i, sample_offset=rank * B)
        if world > 1:
            img = all_gather_images(img, B * world)
        return img

    def passes(first, count, mode=None):
        """`count` whole batches (seeds 1000 + first ...), every batch the full cascade; returns the last batch's images."""
        mode = mode or args.mode
        if count <= 0:
            return None
        if mode == "sequential":
            for i in range(count):
                out = one_pass(first + i)
            return out
        if mode == "pipeline":
            outs = imagen.sample_pipelined([dict(seed=1000 + first + i) for i in range(count)], text_embeds=text_embeds, cond_scale=3.0,
                                           sample_offset=rank * B)
        else:
            import threading
            outs, nxt, lock, errors = [None] * count, [0], threading.Lock(), []

            def run(lane):
                try:
                    with imagen.lane(lane), torch.cuda.device(device):
                        while not errors:
                            with lock:
                                i = nxt[0]
                                nxt[0] += 1
                            if i >= count:
                                return
                            outs[i] = imagen.sample(text_embeds=text_embeds, cond_scale=3.0, use_tqdm=False, seed=1000 + first + i,
                                                    sample_offset=rank * B)
                except BaseException as e:   # noqa: BLE001 — re-raised in the main thread: a dead lane must fail the bench loudly
                    errors.append(e)

            th = [threading.Thread(target=run, args=(1 + l,)) for l in range(args.lanes)]
            [t.start() for t in th]
            [t.join() for t in th]
            if errors:
                raise errors[0]
        if world > 1:
            outs = [all_gather_images(o, B * world) for o in outs]
        return outs[-1]

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # every lane's stages / graphs are built by the warm-up (a lane only exists once a batch has run on it)
    n_warm = max(args.warmup, min(args.lanes, args.steps)) if args.mode == "lanes" else args.warmup
    passes(0, n_warm)
    log(f"{n_warm} warmup passes done ({args.mode})")
    fence()
    t0 = time.perf_counter()
    out = passes(n_warm, args.steps)
    fence()
    elapsed = time.perf_counter() - t0
    log(f"timed region done: {elapsed:.2f}s for {args.steps} passes")
    if world > 1:
        tt = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    assert torch.isfinite(out).all() and out.shape == (B * world, 3, 256, 256)

    if rank == 0:
        images = B * world * args.steps
        value = images / elapsed
        headline = args.timesteps == 1000 and B == 8
        rec = {
            "metric": "images/sec (64->256 cascade, 1000 steps, bs=8)" if headline else f"images/sec (64->256 cascade, {args.timesteps} steps, bs={B}) [NOT the headline config]",
            "value": round(value, 4), "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": n_warm,
            "ms_per_step": round(elapsed / args.steps * 1e3, 2), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f16", "data": "synthetic",
            "config": {"workload": "C3: README unet1 (dim 32, 64^2) + unet2 (dim 32, 256^2, lowres_cond) cascade, "
                                   f"{args.timesteps} DDPM steps/stage, CFG 3.0, dynamic thresholding, batch {B} per GPU",
                       "global_batch": B * world, "parallelism": f"batch-sharded x{world}, one RCCL all-gather of final images",
                       "batch_schedule": {"sequential": "one sample() call after the other",
                                          "pipeline": "successive batches with the cascade stages overlapped (stage 1 of batch k+1 || stage 2 of "
                                                      "batch k, one stream + hipGraph per stage); every batch runs the full cascade, pipeline fill "
                                                      "and drain are inside the timed region",
                                          "lanes": f"{args.lanes} whole cascades side by side (one stream each)"}[args.mode],
                       "images_in_flight": B * (args.lanes if args.mode == "lanes" else (2 if args.mode == "pipeline" else 1)),
                       "denoiser_evals_per_image": 2 * 2 * args.timesteps},
            "path_tflops_reference_count": round(value * FLOPS_PER_IMAGE_REFERENCE * (args.timesteps / 1000) / 1e12, 1),
            "path_frac_of_mfma_peak": round(value * FLOPS_PER_IMAGE_REFERENCE * (args.timesteps / 1000) / 1e12 / (MFMA_PEAK_TFLOPS * world), 4),
        }
        # A/B runs describe themselves: which kernel library was loaded and which probe knobs were set (none in a production run)
        from imagen_pytorch_amd import _abi as _abi_mod
        knobs = {k: v for k, v in os.environ.items() if k.startswith("IMAGEN_") and k not in ("IMAGEN_LIB_PATH",)}
        if os.path.basename(_abi_mod.LIB_PATH) != "libimagen_hip.so" or knobs:
            rec["config"]["kernel_library"] = os.path.basename(_abi_mod.LIB_PATH)
            rec["config"]["probe_knobs"] = knobs
        if world == 1 and args.mode != "sequential":
            # the same cascade as ONE request at a time (latency view): three sequential passes outside the timed region, the median reported
            import threading
            dts, under_load = [], {}
            for k in range(3):
                # clocks / power under load: one rocm-smi reading two seconds into the FIRST pass (the pass that also re-warms the default lane)
                timer = threading.Timer(2.0, lambda: under_load.update(smi_sample(extra=True))) if k == 0 else None
                if timer:
                    timer.daemon = True
                    timer.start()
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                one_pass(n_warm + args.steps + k)
                torch.cuda.synchronize()
                dts.append(time.perf_counter() - t1)
                if timer:
                    timer.join(timeout=30)
            dt = sorted(dts)[1]
            rec["sequential"] = {"ms_per_step": round(dt * 1e3, 2), "value": round(B / dt, 4), "unit": "images/s",
                                 "ms_per_ddpm_step_pair": round(dt * 1e3 / args.timesteps, 4),
                                 "passes_ms": [round(x * 1e3, 1) for x in dts],
                                 "path_frac_of_mfma_peak": round(B / dt * FLOPS_PER_IMAGE_REFERENCE * (args.timesteps / 1000) / 1e12 / MFMA_PEAK_TFLOPS, 4),
                                 "note": "one sample() call at a time (no overlap between batches; round-over-round comparisons use THIS "
                                         "figure): the median of three passes after the timed region",
                                 "rocm_smi_under_load": dict(under_load)}
            # the readings a line is compared by (filled in once the calibration leg has run: rec["sequential"]["box"]).  Round 5 read socclk under load
            # as the separator of the pool's two kinds of box (1200 vs ~130 MHz on one pair); round 6's pairs contradict it (1200 MHz: 7.57 ms per step
            # pair, 107 MHz: 7.36 ms) — the MFMA loop figure (2044 vs 2176 TFLOP/s) and the chase under a copy (1272 vs 1248 ns) ordered THOSE two, so
            # the line carries the raw readings and no class
            try:
                rec["sequential"]["in_graph_step_ms"] = stage_replay_leg(imagen)
            except Exception as e:  # noqa: BLE001
                rec["sequential"]["in_graph_step_error"] = f"{type(e).__name__}: {e}"
            log("sequential passes done")
        if world == 1 and args.mode == "lanes" and headline and not args.no_merged:
            # the same 48 images in flight as six requests MERGED three by three into two batches of 24 (Imagen.sample_requests: every row keeps its
            # request's noise), one lane each — informational, never the headline: the metric's requests are batch 8, and `value` above runs them as such
            try:
                import threading
                M, L = 3, 2
                reqs = lambda first: [dict(text_embeds=text_embeds, seed=5000 + first + j) for j in range(M)]
                errs = []

                def merged_lane(lane, first, **kw):
                    try:
                        with imagen.lane(0x40 + lane), torch.cuda.device(device):
                            imagen.sample_requests(reqs(first), cond_scale=3.0, **kw)
                    except BaseException as e:   # noqa: BLE001
                        errs.append(e)

                def merged_round(first, **kw):
                    th = [threading.Thread(target=merged_lane, args=(l, first + M * l), kwargs=kw) for l in range(L)]
                    [t.start() for t in th]
                    [t.join() for t in th]
                    if errs:
                        raise errs[0]
                merged_round(0, max_steps=8)            # stages, time tables and graphs of the two batch-24 lanes
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                merged_round(M * L)
                torch.cuda.synchronize()
                dtm = time.perf_counter() - t1
                rec["merged_requests"] = {"value": round(B * M * L / dtm, 4), "unit": "images/s", "requests": M * L, "merged_per_batch": M, "lanes": L,
                                          "images_in_flight": B * M * L, "seconds": round(dtm, 2),
                                          "note": "informational: six batch-8 requests sampled as two merged batches of 24 (one set of launches per "
                                                  "denoiser step for three requests), every row with its own request's Philox key"}
                log("merged-requests leg done")
            except Exception as e:  # noqa: BLE001
                rec["merged_requests_error"] = f"{type(e).__name__}: {e}"
        if world == 1:
            try:
                rec["calibration"] = calibration_leg(device)
                ul = (rec.get("sequential") or {}).get("rocm_smi_under_load") or {}
                rec["calibration"]["under_load"] = {"socclk_mhz": _mhz(ul.get("socclk")), "sclk_mhz": _mhz(ul.get("sclk")), "fclk_mhz": _mhz(ul.get("fclk")),
                                                    "power": next((v for k, v in ul.items() if "power" in k.lower()), None)}
                if rec.get("sequential"):
                    cal = rec["calibration"]
                    rec["sequential"]["box"] = {"value": rec["sequential"]["value"], "mfma_f16_loop_TFLOPs": cal.get("mfma_f16_loop_TFLOPs"),
                                                "hbm_chase_under_copy_ns": (cal.get("dependent_load_ns_under_copy") or {}).get("hbm_2GiB"),
                                                "socclk_mhz_under_load": cal["under_load"]["socclk_mhz"], "power_under_load": cal["under_load"]["power"]}
                log("calibration probes done")
            except Exception as e:  # noqa: BLE001
                rec["calibration"] = None
                rec["calibration_error"] = f"{type(e).__name__}: {e}"
        # the extra legs must never cost the headline line: a failure is reported in the record instead
        if world == 1 and not args.no_roofline:
            try:
                # the counter passes are evidence beside the headline, never at its expense: skipped when the run is already long
                spent = time.perf_counter() - t_start
                if not args.no_pmc and spent > 480:
                    log(f"pmc passes skipped: {spent:.0f} s into the run")
                pmc = None if (args.no_pmc or spent > 480) else pmc_traffic_leg(log)
                rec["roofline"], rec["roofline_other_bound"] = roofline_leg(imagen, B, device, pmc)
                log("roofline leg done")
            except Exception as e:  # noqa: BLE001
                rec["roofline"] = None
                rec["roofline_error"] = f"{type(e).__name__}: {e}"
                log(f"roofline leg failed: {e}")
        if world == 1 and not args.no_cpu_baseline:
            try:
                rec["cpu_baseline"] = cpu_baseline_leg(imagen, B)
                log("cpu baseline leg done")
            except Exception as e:  # noqa: BLE001
                rec["cpu_baseline"] = None
                rec["cpu_baseline_error"] = f"{type(e).__name__}: {e}"
                log(f"cpu baseline leg failed: {e}")
        print(json.dumps(rec), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
