"""GPU: fspann_search_store_dev (one call: encode -> route(limit = B) -> refine from the store) == the three calls == oracle."""
import numpy as np
import pytest

from conftest import make_scene

pytestmark = pytest.mark.gpu


def test_search_store_dev_equals_stages_and_oracle(pkg, oracle):
    import torch
    sc = make_scene(oracle, n=20000, d=32, T=8, D=1, m=12, lam=2, B=128, seed=41)
    p = sc["params"]
    nq, B, k = 37, p["B"], 10
    Q = sc["rng"].standard_normal((nq, p["d"])).astype(np.float32)
    o = sc["oracle"]
    ref = o.search(Q.astype(np.float64), k, codes=o.encode(Q.astype(np.float64)))
    cfg = pkg.PaperRuntimeConfig(tables=p["T"], divisions=p["D"], m=p["m"], lambda_=p["lam"], dim=p["d"], refinement_limit=B)
    with pkg.FspannContext(cfg, 0) as ctx:
        ctx.set_gfunctions(sc["alpha"], sc["r"], sc["omega"])
        ctx.set_id_meta(p["n"])
        ctx.build_index(sc["X"])
        dev = torch.device("cuda", 0)
        qd = torch.from_numpy(Q).to(dev)
        out_ids = torch.full((nq, k), -7, dtype=torch.int32, device=dev)
        out_dist = torch.zeros((nq, k), dtype=torch.float64, device=dev)
        out_cnt = torch.zeros(nq, dtype=torch.int32, device=dev)
        scored = torch.zeros(nq, dtype=torch.int32, device=dev)
        sel = torch.full((nq, B), -1, dtype=torch.int32, device=dev)
        selc = torch.zeros(nq, dtype=torch.int32, device=dev)
        F32 = pkg._native.F32
        with pytest.raises(pkg.FspannStateError):          # no store yet
            ctx.search_store_dev(nq, qd.data_ptr(), F32, -1, B, k, out_ids.data_ptr(), out_dist.data_ptr(), out_cnt.data_ptr())
        ctx.store_set(sc["X"])
        ctx.search_store_dev(nq, qd.data_ptr(), F32, -1, B, k, out_ids.data_ptr(), out_dist.data_ptr(), out_cnt.data_ptr(),
                             scored.data_ptr(), sel.data_ptr(), selc.data_ptr())
        ctx.sync()
        # the same through the separate stages (full select with counters)
        codes = ctx.encode(Q)
        codes = codes["codes"] if isinstance(codes, dict) else codes
        routed = ctx.route(codes, limit=B)
        staged = ctx.refine_store(Q, routed["ids"][:, :B], routed["count"], k)
        # and without the optional outputs
        out2 = torch.full((nq, k), -7, dtype=torch.int32, device=dev)
        ctx.search_store_dev(nq, qd.data_ptr(), F32, -1, B, k, out2.data_ptr(), out_dist.data_ptr(), out_cnt.data_ptr())
        ctx.sync()
    got_ids, got_cnt = out_ids.cpu().numpy(), out_cnt.cpu().numpy()
    assert np.array_equal(selc.cpu().numpy(), routed["count"])
    for i in range(nq):
        assert np.array_equal(sel.cpu().numpy()[i, :routed["count"][i]], routed["ids"][i, :routed["count"][i]])
    assert np.array_equal(got_cnt, staged["count"]) and np.array_equal(got_ids, staged["ids"])
    assert np.array_equal(out2.cpu().numpy(), got_ids)
    assert np.array_equal(scored.cpu().numpy(), staged["scored"])
    assert np.array_equal(got_ids, ref["ids"]) and np.array_equal(out_dist.cpu().numpy(), ref["dist"])
