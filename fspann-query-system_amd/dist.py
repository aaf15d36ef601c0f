"""Multi-GPU data path: query batches shard embarrassingly over ranks (index replicated on
every GPU); the ONLY exchange is one all-gather of the per-rank [Q_local x k] (id, distance)
top-k over RCCL/xGMI (torch.distributed backend "nccl" on ROCm; "gloo" in the CPU tests).

The message is tiny (k = 10, Q_local = 1024: 120 KB per rank), so the collective is
latency-bound; ring vs direct and per-link bandwidth do not matter.
Queries that returned fewer than k results are padded with id = -1, dist = +inf.

Refine writes ids and distances straight into ONE byte buffer (TopkBuffer), so the merge is a
single all_gather_into_tensor with no packing kernels.
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


class TopkBuffer:
    """[Q, k] int32 ids followed by [Q, k] float64 distances in one contiguous uint8 tensor."""

    def __init__(self, q: int, k: int, device="cpu"):
        import torch
        self.q, self.k = q, k
        self.id_bytes = ((q * k * 4 + 7) // 8) * 8          # keep the fp64 part 8-byte aligned
        self.nbytes = self.id_bytes + q * k * 8
        self.raw = torch.zeros(self.nbytes, dtype=torch.uint8, device=device)
        self.ids = self.raw[: q * k * 4].view(torch.int32).view(q, k)
        self.dist = self.raw[self.id_bytes:].view(torch.float64).view(q, k)


class GatheredTopk:
    """Receive side: world x TopkBuffer layout; `.ids` / `.dist` are [world*Q, k] in rank (= global query) order."""

    def __init__(self, world: int, q: int, k: int, device="cpu"):
        import torch
        self.world, self.q, self.k = world, q, k
        self.proto = TopkBuffer(q, k, "cpu")
        self.raw = torch.zeros(world * self.proto.nbytes, dtype=torch.uint8, device=device)

    def split(self):
        import torch
        nb, ib = self.proto.nbytes, self.proto.id_bytes
        per = self.raw.view(self.world, nb)
        ids = torch.stack([per[r, : self.q * self.k * 4].view(torch.int32).view(self.q, self.k) for r in range(self.world)])
        dist = torch.stack([per[r, ib:].view(torch.float64).view(self.q, self.k) for r in range(self.world)])
        return ids.reshape(self.world * self.q, self.k), dist.reshape(self.world * self.q, self.k)


def allgather_topk(local: TopkBuffer, out: GatheredTopk, group=None):
    """One collective; every rank contributes the same Q_local (the caller pads the last shard)."""
    import torch.distributed as dist
    dist.all_gather_into_tensor(out.raw, local.raw, group=group)
    return out
