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


class DirectRccl:
    """The same all-gather issued straight through librccl (ncclAllGather on a given HIP stream): a few microseconds of
    host time per call instead of the ~80 us torch.distributed spends in Python/ProcessGroup bookkeeping, which at
    ~100 us per step is what limits a rank.  The communicator is bootstrapped over the existing torch.distributed group
    (rank 0's ncclUniqueId is broadcast), lives next to torch's own, and is verified once against
    `all_gather_into_tensor`; any failure leaves `ok = False` and the caller stays on torch.distributed.
    """

    def __init__(self, world: int, rank: int, device):
        import ctypes as C
        import os
        import torch
        import torch.distributed as dist
        self.ok = False
        self._C = C
        try:
            # every step below is taken by all ranks or by none (a rank that dropped out alone would leave the others
            # waiting in a collective): local failures are agreed on through the existing group first
            def all_agree(flag: bool) -> bool:
                f = torch.tensor([1 if flag else 0], dtype=torch.int32, device=device)
                dist.all_reduce(f, op=dist.ReduceOp.MIN)
                return int(f.item()) == 1

            class UniqueId(C.Structure):
                _fields_ = [("internal", C.c_char * 128)]

            loaded = False
            try:
                if os.environ.get("FSPANN_DIRECT_RCCL", "1") != "0":
                    path = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
                    self.lib = C.CDLL(path)          # the instance torch already loaded
                    self.lib.ncclGetUniqueId.argtypes = [C.POINTER(UniqueId)]
                    self.lib.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, UniqueId, C.c_int]
                    self.lib.ncclAllGather.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_void_p]
                    self.lib.ncclCommDestroy.argtypes = [C.c_void_p]
                    loaded = True
            except OSError:
                loaded = False
            if not all_agree(loaded):
                return
            uid = UniqueId()
            status = 1
            if rank == 0 and self.lib.ncclGetUniqueId(C.byref(uid)) != 0:
                status = 0
            payload = (bytes(bytearray(uid)) if rank == 0 else bytes(128)) + bytes([status])
            t = torch.frombuffer(bytearray(payload), dtype=torch.uint8).to(device)
            dist.broadcast(t, src=0)
            raw = bytes(t.cpu().numpy().tobytes())
            if raw[128] != 1:
                return
            C.memmove(C.byref(uid), raw[:128], 128)
            self.comm = C.c_void_p()
            with torch.cuda.device(device):
                rc = self.lib.ncclCommInitRank(C.byref(self.comm), world, uid, rank)
            if rc != 0:
                raise RuntimeError(f"ncclCommInitRank -> {rc}")
            self.world, self.rank, self.device = world, rank, device
            # one verified exchange before anything relies on it
            probe = torch.arange(rank * 1000, rank * 1000 + 256, dtype=torch.int32, device=device).view(torch.uint8)
            got = torch.zeros(world * probe.numel(), dtype=torch.uint8, device=device)
            ref = torch.zeros_like(got)
            s = torch.cuda.current_stream(device)
            self.allgather_bytes(probe, got, s)
            s.synchronize()
            dist.all_gather_into_tensor(ref, probe)
            torch.cuda.synchronize(device)
            self.ok = bool(torch.equal(got, ref))
        except Exception as e:  # noqa: BLE001 - any failure means "use torch.distributed"
            import sys
            print(f"[fspann] direct RCCL disabled: {e}", file=sys.stderr)
            self.ok = False

    def allgather_bytes(self, send, recv, stream):
        """ncclAllGather(send -> recv) of uint8 tensors on `stream` (a torch.cuda.Stream)."""
        rc = self.lib.ncclAllGather(send.data_ptr(), recv.data_ptr(), send.numel(), 0, self.comm, stream.cuda_stream)  # 0 = ncclInt8
        if rc != 0:
            raise RuntimeError(f"ncclAllGather -> {rc}")

    def allgather_topk(self, local: TopkBuffer, out: GatheredTopk, stream):
        self.allgather_bytes(local.raw, out.raw, stream)
        return out

    def close(self):
        if getattr(self, "comm", None) is not None and self.comm.value:
            try:
                self.lib.ncclCommDestroy(self.comm)
            except Exception:  # noqa: BLE001
                pass
            self.comm = None


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
