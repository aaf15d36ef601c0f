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
        self.id_bytes = ((q * k * 4 + 7) // 8) * 8          # == fspann_topk_dist_offset(q, k): the library's packed layout
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


class LibComm:
    """The merge collective of the C library (`fspann_comm_*`, include/fspann.h): ncclAllGather on the context's own
    stream, a few microseconds of host time per call (torch.distributed spends ~80 us per call in ProcessGroup
    bookkeeping).  This class only does the BOOTSTRAP over an existing torch.distributed group — shipping rank 0's
    unique id — which a JVM deployment does over its own transport; every step that can fail on one rank alone is
    followed by an agreement over the bootstrap group, so all ranks end up with `ok` True or all with False
    (then the caller stays on `allgather_topk` above) and nobody is left waiting inside a collective."""

    def __init__(self, ctx, world: int, rank: int, device="cpu", group=None):
        import ctypes as C
        import sys
        import torch
        import torch.distributed as dist
        from . import _native as N
        self.ok = False
        self.ctx, self.world, self.rank = ctx, world, rank
        self._h = None
        self._L = L = N.lib()

        def agree(flag: bool) -> bool:
            f = torch.tensor([1 if flag else 0], dtype=torch.int32, device=device)
            dist.all_reduce(f, op=dist.ReduceOp.MIN, group=group)
            return int(f.item()) == 1

        def note(msg):
            print(f"[fspann] library collective disabled: {msg}", file=sys.stderr)

        try:
            if not agree(bool(L.fspann_comm_available())):           # librccl missing on some rank: nobody enters the create
                return note("librccl not found on every rank")
            uid = (C.c_char * 128)()
            status = 1
            if rank == 0 and L.fspann_comm_unique_id(uid) != 0:
                status = 0
            t = torch.frombuffer(bytearray(bytes(uid) + bytes([status])), dtype=torch.uint8).to(device)
            dist.broadcast(t, src=0, group=group)
            raw = t.cpu().numpy().tobytes()
            if raw[128] != 1:                                         # the same byte on every rank
                return note("ncclGetUniqueId failed on rank 0")
            h = C.c_void_p()
            rc = L.fspann_comm_create(ctx.handle, raw[:128], world, rank, C.byref(h))
            if rc == 0:
                self._h = h
            if not agree(rc == 0):
                self.close()
                return note("ncclCommInitRank failed on some rank")
            # one verified exchange before anything relies on it
            k, nq = 3, 5
            nb = int(L.fspann_topk_bytes(nq, k))
            probe = ((torch.arange(nb, dtype=torch.int32, device=device) * 7 + rank * 131) % 251).to(torch.uint8)
            got = torch.zeros(world * nb, dtype=torch.uint8, device=device)
            ref = torch.zeros_like(got)
            if device != "cpu":
                torch.cuda.synchronize(device)
            rc = L.fspann_allgather_topk_dev(self._h, nq, k, probe.data_ptr(), got.data_ptr())
            ctx.sync()
            dist.all_gather_into_tensor(ref, probe, group=group)       # every rank takes this one whatever rc was
            if device != "cpu":
                torch.cuda.synchronize(device)
            if not agree(rc == 0 and bool(torch.equal(got, ref))):
                self.close()
                return note("verification exchange differs from all_gather_into_tensor")
            lib = C.c_char_p()
            L.fspann_comm_info(self._h, None, None, C.byref(lib))
            self.library = (lib.value or b"").decode()
            self.ok = True
        except Exception as e:  # noqa: BLE001 - a Python-side failure here is not agreed on: report it loudly
            self.close()
            raise RuntimeError(f"fspann LibComm bootstrap failed on rank {rank}: {e}") from e

    def allgather_topk(self, local: "TopkBuffer", out: "GatheredTopk"):
        """Enqueued on the context's stream (behind the Refine that filled `local`)."""
        from . import _native as N
        N.check(self._L.fspann_allgather_topk_dev(self._h, local.q, local.k, local.raw.data_ptr(), out.raw.data_ptr()))
        return out

    def close(self):
        if self._h is not None:
            self._L.fspann_comm_destroy(self._h)
            self._h = None
        self.ok = False


class DeviceEvent:
    """HIP event with a DEVICE-scope release (hipEventReleaseToDevice | hipEventDisableTiming).
    A default event performs a system-scope fence when it is recorded — L2 writeback + invalidate — which costs the
    recording stream ~10 us per step here and cools the caches for the kernels that follow.  Producer and consumer of the
    top-k buffer are streams of the SAME GPU, so device scope is what the hand-off needs."""

    _lib = None

    def __init__(self):
        import ctypes as C
        if DeviceEvent._lib is None:
            lib = C.CDLL("libamdhip64.so")
            lib.hipEventCreateWithFlags.argtypes = [C.POINTER(C.c_void_p), C.c_uint]
            lib.hipEventRecord.argtypes = [C.c_void_p, C.c_void_p]
            lib.hipStreamWaitEvent.argtypes = [C.c_void_p, C.c_void_p, C.c_uint]
            lib.hipEventDestroy.argtypes = [C.c_void_p]
            DeviceEvent._lib = lib
        self.h = C.c_void_p()
        rc = 1
        # hipEventDisableTiming (0x2) with a device-scope release (0x40000000); older runtimes: no system fence (0x20000000)
        for flags in (0x2 | 0x40000000, 0x2 | 0x20000000, 0x2):
            rc = DeviceEvent._lib.hipEventCreateWithFlags(C.byref(self.h), flags)
            if rc == 0:
                self.flags = flags
                break
        if rc != 0:
            raise RuntimeError(f"hipEventCreateWithFlags -> {rc}")

    def record(self, stream):
        """stream: torch.cuda.Stream (or ExternalStream)."""
        rc = DeviceEvent._lib.hipEventRecord(self.h, stream.cuda_stream)
        if rc != 0:
            raise RuntimeError(f"hipEventRecord -> {rc}")

    def wait(self, stream):
        """Make `stream` wait for this event."""
        rc = DeviceEvent._lib.hipStreamWaitEvent(stream.cuda_stream, self.h, 0)
        if rc != 0:
            raise RuntimeError(f"hipStreamWaitEvent -> {rc}")
