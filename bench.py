#!/usr/bin/env python
"""bench.py — images/sec of the Imagen 64->256 cascade on MI355X (BASELINE.json metric), one process per GPU.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A "step" is ONE full pass of the hot path over one batch: `Imagen.sample()` of the README cascade (unet1 @64^2 ->
unet2 @256^2), 1000 DDPM steps per stage, classifier-free guidance 3.0, batch 8 PER GPU (weak scaling: each rank
samples its own 8 prompts and the final images are all-gathered over RCCL).  Inputs are synthetic and resident in
HBM before the timed region: random-init weights (final_conv ~ N(0, 0.05^2)), random text_embeds,
in-kernel Philox noise.  Rank 0 prints one JSON line.

Extra legs (rank 0, N = 1 only):
  roofline     — HIP-event timing, on the launch stream, of every launch of the dominant kernel symbol (the 128x128 MFMA
                 implicit-GEMM tile `igemm_kernel<2,2,2,2,4>`) in one denoiser step of each stage; achieved = algorithmic
                 FLOPs (2*MACs of the convolution / linear it computes) per launch / average duration, vs the 2.5 PFLOP/s
                 dense fp16 MFMA peak (/opt/skills/guides/MI355X_MICROARCH.md).
  cpu_baseline — the CPU oracle (oracle/: fp32 torch restatement of the reference path, "port") timed on this box's host
                 cores for ONE DDPM step per stage at batch 8 (2 CFG forwards each), linearly extrapolated to 1000 steps.
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


def roofline_leg(imagen, batch: int, device):
    """Event-time every launch of the dominant igemm tile symbol in one denoiser step of both stages (eager, same stream)."""
    import ctypes
    from imagen_pytorch_amd import _abi, ops

    lib = _abi.load_library()
    K_IGEMM = _abi.ENUMS["IMAGEN_OP_IGEMM"]
    per_cfg = {}
    stream = torch.cuda.current_stream()
    h = stream.cuda_stream
    seen = set()
    for key, st in imagen._stages.items():
        if key[:3] in seen:      # one engine per (stage, batch, size): lanes hold copies of the same plan
            continue
        seen.add(key[:3])
        plan = st["plan"]
        st["step_ptr"].zero_()
        evs = []
        for kind, struct, label in plan.ops:
            if kind == K_IGEMM:
                e0, e1 = ctypes.c_void_p(), ctypes.c_void_p()
                lib.imagen_event_create(ctypes.byref(e0))
                lib.imagen_event_create(ctypes.byref(e1))
                lib.imagen_event_record(e0, h)
                _abi.check(lib.imagen_launch(kind, ctypes.addressof(struct), h))
                lib.imagen_event_record(e1, h)
                evs.append((struct.cfg, igemm_flops(struct), e0, e1))
            else:
                _abi.check(lib.imagen_launch(kind, ctypes.addressof(struct), h))
        torch.cuda.synchronize()
        for cfg, fl, e0, e1 in evs:
            ms = ctypes.c_float()
            lib.imagen_event_elapsed_ms(e0, e1, ctypes.byref(ms))
            d = per_cfg.setdefault(cfg, [0, 0.0, 0.0])
            d[0] += 1
            d[1] += fl
            d[2] += ms.value * 1e-3
            lib.imagen_event_destroy(e0)
            lib.imagen_event_destroy(e1)
    # dominant symbol = the tile configuration with the largest total time
    cfg, (n, fl, sec) = max(per_cfg.items(), key=lambda kv: kv[1][2])
    tab = ops.cfg_table()
    achieved = fl / sec / 1e12
    # HBM traffic per launch of that symbol from the committed PMC summaries (separate FETCH_SIZE / WRITE_SIZE passes of
    # `rocprofv3 --pmc`, tools/pmc_summary.py); KiB units, FETCH_SIZE doubled (gfx950 counts 128-B requests as 64 B)
    traffic = None
    try:
        fetch = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_fetch.json")))
        write = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_write.json")))
        mi, ni, wm, wn, g = CFG_TEMPLATE[cfg]
        pat = f"igemm_kernel<{mi}, {ni}, {wm}, {wn}, {g},"
        fs = [(v["launches"], v.get("FETCH_SIZE", 0.0)) for k, v in fetch.items() if pat in k]
        ws = [(v["launches"], v.get("WRITE_SIZE", 0.0)) for k, v in write.items() if pat in k]
        if fs and ws:
            favg = sum(a * b for a, b in fs) / sum(a for a, _ in fs)
            wavg = sum(a * b for a, b in ws) / sum(a for a, _ in ws)
            traffic = round((2.0 * favg + wavg) * 1024.0)
    except (OSError, ValueError, KeyError):
        pass
    summary = {str(c): {"launches": v[0], "tflops": round(v[1] / max(v[2], 1e-12) / 1e12, 1), "ms_total": round(v[2] * 1e3, 3)}
               for c, v in sorted(per_cfg.items())}
    return {
        "bound": "mfma", "achieved": round(achieved, 1), "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(achieved / MFMA_PEAK_TFLOPS, 4),
        "traffic": traffic,
        "kernel": f"{'conv_lds_kernel' if tab[cfg][3] else 'igemm_kernel'} cfg {cfg} ({tab[cfg][0]} px x {tab[cfg][1]} cout tile, G={tab[cfg][2]})",
        "launches_per_denoiser_step_pair": n, "avg_launch_us": round(sec / n * 1e6, 2), "avg_launch_gflop": round(fl / n / 1e9, 3),
        "per_cfg": summary,
    }


def cpu_baseline_leg(imagen, batch: int):
    """The oracle ("port" of the reference path, fp32 torch CPU) for one DDPM step per stage, extrapolated."""
    from oracle import sampler_oracle as so
    from oracle import unet_oracle as uo

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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=8, help="images per GPU (BASELINE: 8)")
    ap.add_argument("--timesteps", type=int, default=1000, help="DDPM steps per stage (BASELINE: 1000); other values are NOT the headline metric")
    ap.add_argument("--mode", choices=("sequential", "pipeline", "lanes"), default=os.environ.get("IMAGEN_BENCH_MODE", "pipeline"),
                    help="how successive batches are scheduled on the GPU: one sample() after the other | cascade stages overlapped across "
                         "batches (Imagen.sample_pipelined) | --lanes whole cascades side by side (one thread + stream each)")
    ap.add_argument("--lanes", type=int, default=2)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    args = ap.parse_args()

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

    def passes(first, count):
        """`count` whole batches (seeds 1000 + first ...), every batch the full cascade; returns the last batch's images."""
        if count <= 0:
            return None
        if args.mode == "sequential":
            for i in range(count):
                out = one_pass(first + i)
            return out
        if args.mode == "pipeline":
            outs = imagen.sample_pipelined([dict(seed=1000 + first + i) for i in range(count)], text_embeds=text_embeds, cond_scale=3.0,
                                           sample_offset=rank * B)
        else:
            import threading
            outs, nxt, lock = [None] * count, [0], threading.Lock()

            def run(lane):
                with imagen.lane(lane), torch.cuda.device(device):
                    while True:
                        with lock:
                            i = nxt[0]
                            nxt[0] += 1
                        if i >= count:
                            return
                        outs[i] = imagen.sample(text_embeds=text_embeds, cond_scale=3.0, use_tqdm=False, seed=1000 + first + i,
                                                sample_offset=rank * B)

            th = [threading.Thread(target=run, args=(1 + l,)) for l in range(args.lanes)]
            [t.start() for t in th]
            [t.join() for t in th]
        if world > 1:
            outs = [all_gather_images(o, B * world) for o in outs]
        return outs[-1]

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    passes(0, args.warmup)
    log(f"{args.warmup} warmup passes done ({args.mode})")
    fence()
    t0 = time.perf_counter()
    out = passes(args.warmup, args.steps)
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
            "value": round(value, 4), "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
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
                       "denoiser_evals_per_image": 2 * 2 * args.timesteps},
            "path_tflops_reference_count": round(value * FLOPS_PER_IMAGE_REFERENCE * (args.timesteps / 1000) / 1e12, 1),
            "path_frac_of_mfma_peak": round(value * FLOPS_PER_IMAGE_REFERENCE * (args.timesteps / 1000) / 1e12 / (MFMA_PEAK_TFLOPS * world), 4),
        }
        if world == 1 and not args.no_roofline:
            rec["roofline"] = roofline_leg(imagen, B, device)
            log("roofline leg done")
        if world == 1 and not args.no_cpu_baseline:
            rec["cpu_baseline"] = cpu_baseline_leg(imagen, B)
            log("cpu baseline leg done")
        print(json.dumps(rec), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
