"""Clip sharding for one-process-per-GPU inference.

Inference needs no collective: clips are independent units, weights are replicated, each rank runs the HIP path on its
contiguous slice of the batch (what ``nn.DataParallel`` does on dim 0 in the reference,
test_emotion_gesture_diversity_iterative.py:137-138).  ``gather_poses`` is the optional metric-side collection
(17 KB per clip); it is never on the timed data path.
"""
from __future__ import annotations

from typing import Tuple

import torch


def shard_range(n_items: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous [begin, end) of ``n_items`` clips owned by ``rank``: the first ``n_items % world`` ranks get one extra."""
    if world <= 0 or not (0 <= rank < world) or n_items < 0:
        raise ValueError(f"bad shard request n={n_items} rank={rank} world={world}")
    base, extra = divmod(n_items, world)
    begin = rank * base + min(rank, extra)
    return begin, begin + base + (1 if rank < extra else 0)


def shard_batch(tensors, rank: int, world: int):
    """Slice every tensor of a dict/tuple on dim 0 to this rank's clips."""
    if isinstance(tensors, dict):
        n = next(iter(tensors.values())).shape[0]
        b, e = shard_range(n, rank, world)
        return {k: (v[b:e] if v is not None else None) for k, v in tensors.items()}
    n = tensors[0].shape[0]
    b, e = shard_range(n, rank, world)
    return tuple(t[b:e] if t is not None else None for t in tensors)


def gather_poses(pose: torch.Tensor, n_total: int, group=None) -> torch.Tensor:
    """All-gather per-rank pose shards [n_r, F, D] (ragged over ranks) into [n_total, F, D] on every rank."""
    import torch.distributed as dist

    world, rank = dist.get_world_size(group), dist.get_rank(group)
    sizes = [shard_range(n_total, r, world)[1] - shard_range(n_total, r, world)[0] for r in range(world)]
    cap = max(sizes)
    pad = torch.zeros((cap,) + tuple(pose.shape[1:]), dtype=pose.dtype, device=pose.device)
    pad[: pose.shape[0]] = pose
    outs = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(outs, pad, group=group)
    return torch.cat([o[:s] for o, s in zip(outs, sizes)], 0)
