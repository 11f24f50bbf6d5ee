"""Clip sharding for one-process-per-GPU inference.

Inference needs no collective: clips are independent units, weights are replicated, each rank runs the HIP path on its
contiguous slice of the batch (what ``nn.DataParallel`` does on dim 0 in the reference,
test_emotion_gesture_diversity_iterative.py:137-138).  ``gather_poses`` is the optional metric-side collection
(17 KB per clip); it is never on the timed data path.

``init_process_group`` is the one rendezvous of every one-process-per-GPU launcher of this package (bench.py's self-launch, the
K-fold loop's workers, the two-rank tests): a launcher that starts its own ranks hands them a FILE (``EG_DIST_STORE``) instead of a
probed TCP port, so there is no window in which another process can take "the free port" between the probe and rank 0's bind (round 5's
flaky two-rank test).  Under an outer launcher (torchrun: MASTER_ADDR / MASTER_PORT in the environment) it is plain ``env://``.
"""
from __future__ import annotations

from typing import Tuple

import torch


def new_store_path(tag: str = "eg") -> str:
    """A fresh rendezvous file name for `EG_DIST_STORE` (the launcher passes it to the ranks it starts; rank 0 creates the file)."""
    import os
    import tempfile
    import uuid
    return os.path.join(tempfile.gettempdir(), f"{tag}_store_{os.getpid()}_{uuid.uuid4().hex}")


def init_process_group(backend: str, rank: int, world: int, device=None, timeout_s: float = 600.0):
    """torch.distributed.init_process_group over a FileStore when the launcher published one in EG_DIST_STORE, else env:// (MASTER_ADDR /
    MASTER_PORT).  backend "nccl" is RCCL on ROCm; `device` binds the communicator to the rank's GPU at once (eager init)."""
    import datetime
    import os
    import torch.distributed as dist
    kw = {"timeout": datetime.timedelta(seconds=timeout_s)}
    if backend == "nccl" and device is not None:
        kw["device_id"] = device
    path = os.environ.get("EG_DIST_STORE")
    if path:
        kw["store"] = dist.FileStore(path, world)
    dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return dist


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
