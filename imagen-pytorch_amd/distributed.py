"""Multi-GPU sampling: one process per GPU, batch sharded by independent samples, ONE collective.

The sampling path has no cross-sample interaction (all norms are per pixel / token / sample, the dynamic
threshold quantile is per sample — SURVEY.md §8e), so every rank runs the whole cascade on its shard with its
own hipGraphs and the only exchange is a single RCCL all-gather of the final images over xGMI
(`torch.distributed` backend "nccl" = RCCL on ROCm).  The reference has no counterpart: under `accelerate
launch` each rank just samples independently (trainer.py:947-961).

Noise is counter-based and keyed by the GLOBAL sample index (`sample_offset`), so the gathered result equals the
single-GPU result for the same seed regardless of the world size.
"""
from __future__ import annotations

from typing import Callable, Optional, Tuple

import torch
import torch.distributed as dist


def shard_bounds(total: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous, balanced split of `total` samples: the first (total % world) ranks get one extra."""
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def sample_sharded(sample_fn: Callable[..., torch.Tensor], text_embeds: torch.Tensor, *, text_masks: Optional[torch.Tensor] = None,
                   group=None, gather: bool = True, **kwargs) -> torch.Tensor:
    """Run `sample_fn(text_embeds=shard, text_masks=shard, sample_offset=lo, **kwargs)` on this rank's shard and all-gather.

    `sample_fn` is normally `Imagen.sample`; it must return a (b_local, C, H, W) tensor on the rank's device.
    Returns the full (B, C, H, W) batch on every rank (or the local shard if gather=False).
    """
    if not dist.is_available() or not dist.is_initialized():
        return sample_fn(text_embeds=text_embeds, text_masks=text_masks, sample_offset=0, **kwargs)
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    B = text_embeds.shape[0]
    lo, hi = shard_bounds(B, rank, world)
    local = None
    if hi > lo:
        local = sample_fn(text_embeds=text_embeds[lo:hi], text_masks=None if text_masks is None else text_masks[lo:hi],
                          sample_offset=lo, **kwargs)
    if not gather:
        return local
    return all_gather_images(local, B, group=group)


def all_gather_images(local: Optional[torch.Tensor], total: int, *, group=None) -> torch.Tensor:
    """One all-gather of the final images.  Equal shards use all_gather_into_tensor (a single ring collective over
    xGMI); ragged shards are padded to the largest shard first (the payload is ~0.8 MB/image, bandwidth is irrelevant)."""
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    sizes = [shard_bounds(total, r, world)[1] - shard_bounds(total, r, world)[0] for r in range(world)]
    if local is not None and all(s == sizes[0] for s in sizes):
        # the common case (B divisible by the world size): exactly one collective, no metadata exchange
        out = torch.empty((total, *local.shape[1:]), dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(out, local.contiguous(), group=group)
        return out
    shapes = [None]
    if local is not None:
        shapes = [tuple(local.shape[1:]), str(local.dtype), str(local.device)]
    # every rank needs the image shape (ranks with an empty shard learn it from rank 0, which always has the first shard)
    obj = [shapes if rank == 0 else None]
    dist.broadcast_object_list(obj, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
    img_shape = obj[0][0]
    dtype = local.dtype if local is not None else torch.float32
    device = local.device if local is not None else (torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu"))
    mx = max(sizes)
    if all(s == mx for s in sizes):
        out = torch.empty((total, *img_shape), dtype=dtype, device=device)
        dist.all_gather_into_tensor(out, local.contiguous(), group=group)
        return out
    padded = torch.zeros((mx, *img_shape), dtype=dtype, device=device)
    if local is not None:
        padded[: local.shape[0]] = local
    parts = [torch.empty_like(padded) for _ in range(world)]
    dist.all_gather(parts, padded, group=group)
    return torch.cat([p[:s] for p, s in zip(parts, sizes)], dim=0)
