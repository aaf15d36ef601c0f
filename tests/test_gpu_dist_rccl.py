"""GPU: the merge collective of the C library (fspann_comm_* / fspann_allgather_topk_dev, include/fspann.h) on a
world-size-1 RCCL communicator: bootstrap of the unique id over torch.distributed (dist.LibComm), the verified probe
exchange, a packed TopkBuffer round trip on the context's stream, and the raw C entry points without torch."""
import ctypes as C
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _ctx(pkg):
    cfg = pkg.PaperRuntimeConfig(tables=2, divisions=1, m=8, lambda_=2, dim=8, refinement_limit=64)
    return pkg.FspannContext(cfg, 0)


def test_library_allgather_world1(pkg):
    import torch
    import torch.distributed as dist
    from fspann_amd import dist as fdist
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        with _ctx(pkg) as ctx:
            comm = fdist.LibComm(ctx, 1, 0, dev)
            assert comm.ok, "the library collective should come up (the caller falls back to torch.distributed otherwise)"
            assert "rccl" in comm.library
            q, k = 33, 10
            local = fdist.TopkBuffer(q, k, dev)
            assert local.nbytes == pkg._native.lib().fspann_topk_bytes(q, k)
            assert local.id_bytes == pkg._native.lib().fspann_topk_dist_offset(q, k)
            rng = np.random.default_rng(3)
            local.ids.copy_(torch.from_numpy(rng.integers(-1, 10**6, (q, k)).astype(np.int32)))
            local.dist.copy_(torch.from_numpy(rng.standard_normal((q, k))))
            out = fdist.GatheredTopk(1, q, k, dev)
            torch.cuda.synchronize()
            comm.allgather_topk(local, out)
            ctx.sync()
            ids, dd = out.split()
            assert torch.equal(ids, local.ids) and torch.equal(dd, local.dist)
            ev = fdist.DeviceEvent()
            ev.record(torch.cuda.ExternalStream(ctx.stream, device=dev))
            ev.wait(torch.cuda.current_stream(dev))
            torch.cuda.synchronize()
            comm.close()
    finally:
        dist.destroy_process_group()


def test_comm_c_abi_without_torch_distributed(pkg):
    """What a JVM does: unique id from rank 0, create, all-gather, destroy — plain C calls, device memory from the library."""
    L = pkg._native.lib()
    assert L.fspann_comm_available() == 1
    with _ctx(pkg) as ctx:
        uid = (C.c_char * 128)()
        pkg._native.check(L.fspann_comm_unique_id(uid))
        h = C.c_void_p()
        pkg._native.check(L.fspann_comm_create(ctx.handle, uid, 1, 0, C.byref(h)))
        w, r = C.c_int(0), C.c_int(-1)
        pkg._native.check(L.fspann_comm_info(h, C.byref(w), C.byref(r), None))
        assert (w.value, r.value) == (1, 0)
        nq, k = 7, 4
        nb = L.fspann_topk_bytes(nq, k)
        src = np.arange(nb, dtype=np.uint8)
        a, b = C.c_void_p(), C.c_void_p()
        pkg._native.check(L.fspann_dev_alloc(ctx.handle, nb, C.byref(a)))
        pkg._native.check(L.fspann_dev_alloc(ctx.handle, nb, C.byref(b)))
        pkg._native.check(L.fspann_h2d(ctx.handle, a, src.ctypes.data_as(C.c_void_p), nb))
        pkg._native.check(L.fspann_allgather_topk_dev(h, nq, k, a, b))
        got = np.zeros(nb, np.uint8)
        pkg._native.check(L.fspann_d2h(ctx.handle, got.ctypes.data_as(C.c_void_p), b, nb))
        assert np.array_equal(got, src)
        with pytest.raises(pkg.FspannArgumentError):
            pkg._native.check(L.fspann_comm_create(ctx.handle, uid, 2, 5, C.byref(C.c_void_p())))
        with pytest.raises(pkg.FspannNullError):
            pkg._native.check(L.fspann_allgather_topk_dev(h, nq, k, None, b))
        L.fspann_dev_free(ctx.handle, a)
        L.fspann_dev_free(ctx.handle, b)
        L.fspann_comm_destroy(h)


def test_context_destroyed_before_its_communicator(pkg):
    """A communicator launches on its context's stream: destroying the context first must not leave it a dangling pointer —
    the context lives until the last fspann_comm_destroy (the all-gather still works in between)."""
    L = pkg._native.lib()
    cfg = pkg.PaperRuntimeConfig(tables=1, divisions=1, m=8, lambda_=2, dim=8, refinement_limit=16)
    ctx = pkg.FspannContext(cfg, 0)
    uid = (C.c_char * 128)()
    pkg._native.check(L.fspann_comm_unique_id(uid))
    h = C.c_void_p()
    pkg._native.check(L.fspann_comm_create(ctx.handle, uid, 1, 0, C.byref(h)))
    raw = ctx.handle
    nq, k = 5, 3
    nb = L.fspann_topk_bytes(nq, k)
    a, b = C.c_void_p(), C.c_void_p()
    pkg._native.check(L.fspann_dev_alloc(raw, nb, C.byref(a)))
    pkg._native.check(L.fspann_dev_alloc(raw, nb, C.byref(b)))
    src = np.arange(nb, dtype=np.uint8)
    pkg._native.check(L.fspann_h2d(raw, a, src.ctypes.data_as(C.c_void_p), nb))
    ctx.close()                                               # fspann_ctx_destroy: deferred, the communicator holds the context
    pkg._native.check(L.fspann_allgather_topk_dev(h, nq, k, a, b))
    got = np.zeros(nb, np.uint8)
    pkg._native.check(L.fspann_d2h(raw, got.ctypes.data_as(C.c_void_p), b, nb))
    assert np.array_equal(got, src)
    L.fspann_dev_free(raw, a)
    L.fspann_dev_free(raw, b)
    assert L.fspann_comm_destroy(h) == 0                      # the last holder finishes the destroy
