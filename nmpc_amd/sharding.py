"""Multi-GPU sharding of a batch of independent DDP problems (one process per GPU).

Every instance is a self-contained solve (the reference keeps all solver state per object, DDPSolver.h:329-374),
so a batch shards embarrassingly: rank r owns the contiguous slice `shard_range(B, r, world)`; there is no
exchange during the iterations and exactly ONE collective at the end of a job — an all-gather of the packed
result records (RCCL over xGMI when the process group's backend is "nccl"; "gloo" in the CPU tests).
"""
from __future__ import annotations

from typing import Tuple

import numpy as np


def shard_range(batch: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous, balanced slice [lo, hi) of `batch` instances owned by `rank` (sizes differ by at most 1)."""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    base, rem = divmod(batch, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_sizes(batch: int, world: int):
    return [shard_range(batch, r, world)[1] - shard_range(batch, r, world)[0] for r in range(world)]


def pack_results(X: np.ndarray, U: np.ndarray, cost: np.ndarray, status: np.ndarray, iters: np.ndarray) -> np.ndarray:
    """One float64 record per instance: [X | U | cost | status | iters] (what the final gather moves)."""
    B = X.shape[0]
    return np.concatenate([X.reshape(B, -1), U.reshape(B, -1), cost.reshape(B, -1),
                           status.reshape(B, 1).astype(np.float64), iters.reshape(B, 1).astype(np.float64)], axis=1)


def unpack_results(rec: np.ndarray, T: int, n: int, mm: int):
    B = rec.shape[0]
    o = 0
    X = rec[:, o:o + (T + 1) * n].reshape(B, T + 1, n)
    o += (T + 1) * n
    U = rec[:, o:o + T * mm].reshape(B, T, mm)
    o += T * mm
    cost = rec[:, o:o + T + 1]
    o += T + 1
    status = rec[:, o].astype(np.int32)
    iters = rec[:, o + 1].astype(np.int32)
    return X, U, cost, status, iters


def record_width(T: int, n: int, mm: int) -> int:
    return (T + 1) * n + T * mm + (T + 1) + 2


def all_gather_records(local, batch: int, group=None):
    """The one collective of a sharded solve.  `local` is this rank's (shard, width) torch tensor; returns the
    (batch, width) tensor of all shards in rank order.  Uneven shards are padded to the largest shard so that a
    single fixed-size all_gather suffices."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    sizes = shard_sizes(batch, world)
    width = local.shape[1]
    pad = max(sizes)
    send = local
    if local.shape[0] < pad:
        send = torch.zeros((pad, width), dtype=local.dtype, device=local.device)
        send[: local.shape[0]] = local
    out = torch.empty((world * pad, width), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, send.contiguous(), group=group)
    parts = [out[r * pad: r * pad + sizes[r]] for r in range(world)]
    return torch.cat(parts, dim=0)
