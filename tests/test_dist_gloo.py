"""CPU, world_size = 2, gloo: the N > 1 data path of bench.py — query sharding + the single
all-gather top-k merge (fspann-query-system_amd/dist.py)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import __graft_entry__ as graft


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, nq, k, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        graft.load_package()
        from fspann_amd import dist as fd
        rng = np.random.default_rng(123)           # same on every rank: the "global" answer
        ids_all = rng.integers(0, 1_000_000, (nq, k)).astype(np.int32)
        dist_all = np.sort(rng.random((nq, k)), axis=1)
        short = rng.integers(0, k + 1, nq)          # some queries return fewer than k
        for i in range(nq):
            ids_all[i, short[i]:] = -1
            dist_all[i, short[i]:] = np.inf
        qloc = (nq + world - 1) // world            # equal shard size, last one padded
        lo, hi = rank * qloc, min((rank + 1) * qloc, nq)
        my_ids = np.full((qloc, k), -1, np.int32)
        my_dist = np.full((qloc, k), np.inf)
        my_ids[: hi - lo] = ids_all[lo:hi]
        my_dist[: hi - lo] = dist_all[lo:hi]
        local = fd.TopkBuffer(qloc, k)
        local.ids.copy_(torch.from_numpy(my_ids))          # (on the GPU path Refine writes these views directly)
        local.dist.copy_(torch.from_numpy(my_dist))
        out = fd.allgather_topk(local, fd.GatheredTopk(world, qloc, k))
        g_ids, g_dist = out.split()
        ok = np.array_equal(g_ids.numpy()[:nq], ids_all) and np.array_equal(g_dist.numpy()[:nq], dist_all)
        ok = ok and g_ids.shape == (world * qloc, k)
        # shard_bounds covers [0, nq) without gaps or overlap
        spans = [fd.shard_bounds(nq, world, r) for r in range(world)]
        ok = ok and spans[0][0] == 0 and spans[-1][1] == nq and all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
        ret[rank] = bool(ok)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("nq,k", [(10, 3), (257, 10)])
def test_allgather_topk_gloo_world2(nq, k):
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), nq, k, ret), nprocs=world, join=True)
    assert dict(ret) == {0: True, 1: True}


def test_shard_bounds():
    graft.load_package()
    from fspann_amd import dist as fd
    assert [fd.shard_bounds(10, 4, r) for r in range(4)] == [(0, 3), (3, 6), (6, 8), (8, 10)]
    assert fd.shard_bounds(0, 2, 1) == (0, 0)
    with pytest.raises(ValueError):
        fd.shard_bounds(10, 2, 2)


def _agree_worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        graft.load_package()
        from fspann_amd import dist as fd

        class NoGpuCtx:                      # no device here: fspann_comm_create fails at once on every rank (null context)
            handle = None

            def sync(self):
                pass

        comm = fd.LibComm(NoGpuCtx(), world, rank, "cpu")
        # every rank left the bootstrap the same way: the next collective of the group still pairs up
        t = torch.tensor([rank + 1])
        dist.all_reduce(t)
        ret[rank] = (comm.ok, int(t.item()))
    finally:
        dist.destroy_process_group()


def test_libcomm_bootstrap_failure_is_agreed_on_by_all_ranks():
    """A rank that cannot create its communicator must not leave its peers inside a collective: the bootstrap agrees on
    every step over the existing group, all ranks fall back together (DirectRccl skipped the reference gather)."""
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_agree_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    assert dict(ret) == {0: (False, 3), 1: (False, 3)}
