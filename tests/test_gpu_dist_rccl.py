"""GPU: the direct librccl all-gather (dist.DirectRccl) on a world-size-1 RCCL group: bootstrap over torch.distributed,
self-check against all_gather_into_tensor, TopkBuffer round trip on a non-default stream."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_direct_rccl_allgather_world1(pkg):
    import torch
    import torch.distributed as dist
    from fspann_amd import dist as fdist
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        r = fdist.DirectRccl(1, 0, dev)
        assert r.ok, "direct RCCL path should come up (falls back to torch.distributed otherwise)"
        q, k = 33, 10
        local = fdist.TopkBuffer(q, k, dev)
        rng = np.random.default_rng(3)
        local.ids.copy_(torch.from_numpy(rng.integers(-1, 10**6, (q, k)).astype(np.int32)))
        local.dist.copy_(torch.from_numpy(rng.standard_normal((q, k))))
        out = fdist.GatheredTopk(1, q, k, dev)
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        r.allgather_topk(local, out, side)
        side.synchronize()
        ids, dd = out.split()
        assert torch.equal(ids, local.ids) and torch.equal(dd, local.dist)
        ev = fdist.DeviceEvent()
        ev.record(side)
        ev.wait(torch.cuda.current_stream(dev))
        torch.cuda.synchronize()
        r.close()
    finally:
        dist.destroy_process_group()
