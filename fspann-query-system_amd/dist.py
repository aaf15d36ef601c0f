"""Multi-GPU data path: query batches shard embarrassingly over ranks (index replicated on
every GPU); the ONLY exchange is one all-gather of the per-rank [Q_local x k] (id, distance)
top-k over RCCL/xGMI (torch.distributed backend "nccl" on ROCm; "gloo" in the CPU tests).

The message is tiny (k = 10, Q_local = 1024: 160 KB per rank), so the collective is
latency-bound; ring vs direct and per-link bandwidth do not matter.
Queries that returned fewer than k results are padded with id = -1, dist = +inf.
"""
from __future__ import annotations

from typing import Tuple


def shard_bounds(n_queries: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous slice [lo, hi) of the global query batch owned by `rank` (balanced, deterministic)."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError("bad world/rank")
    base, rem = divmod(n_queries, world)
    lo = rank * base + min(rank, rem)
    hi = lo + base + (1 if rank < rem else 0)
    return lo, hi


def pack_topk(out_ids, out_dist):
    """[Q, k] int32 ids + [Q, k] float64 distances -> [Q, k, 2] float64 (ids < 2^31 are exact in fp64)."""
    import torch
    return torch.stack((out_ids.to(torch.float64), out_dist.to(torch.float64)), dim=-1).contiguous()


def unpack_topk(packed):
    import torch
    return packed[..., 0].to(torch.int32), packed[..., 1]


def allgather_topk(out_ids, out_dist, group=None, out=None):
    """All ranks contribute the same Q_local (the caller pads the last shard); returns
    ([world*Q_local, k] ids, [world*Q_local, k] dist) in rank order == global query order."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    packed = pack_topk(out_ids, out_dist)
    q, k, _ = packed.shape
    if out is None:
        out = torch.empty((world * q, k, 2), dtype=torch.float64, device=packed.device)
    dist.all_gather_into_tensor(out, packed, group=group)
    return unpack_topk(out)
