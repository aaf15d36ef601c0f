"""Fail loudly where java.util.HashMap would treeify a bin.

The Route order key (HashMap bin, then first insertion) is the JVM's iteration order only while every bin is a plain
chain.  A put that finds 8 nodes in its bin (table >= 64) makes the JVM treeify it (HashMap.putVal, TREEIFY_THRESHOLD);
from then on the order is not modelled — by the oracle (`unmodelled`) or by the product.  The product must DETECT that
exactly and refuse (count = -1, FSPANN_E_STATE) instead of returning a list the JVM would not produce
(PIS:619,690-693; PIS:413 + idx/GreedyPartitioner.java:45-51 for the build).  The oracle's literal HashMap model says,
per query, whether a treeifyBin() on a table >= 64 was met: the two must agree query by query, across capacity stages.
"""
import numpy as np
import pytest

from conftest import make_scene

pytestmark = pytest.mark.gpu


def _ctx(pkg, sc, jh, hard_cap=None, B=None):
    p = sc["params"]
    cfg = pkg.PaperRuntimeConfig(tables=p["T"], divisions=p["D"], m=p["m"], lambda_=p["lam"], dim=p["d"],
                                 refinement_limit=B or p["B"], max_global_candidates=hard_cap or p["hard_cap"])
    ctx = pkg.FspannContext(cfg, 0)
    ctx.set_gfunctions(sc["alpha"], sc["r"], sc["omega"])
    ctx.set_id_meta(p["n"], jh)
    return ctx


def _import(ctx, o):
    for td in range(o.TD):
        ctx.set_index(td, **o.get_index(td))
    ctx.finalize()


def test_build_index_refuses_treeified_staging_map(pkg, oracle):
    n = 6000
    sc = make_scene(oracle, n=n, d=8, T=2, D=1, m=8, lam=2, B=64, seed=3)
    jh = (np.arange(n) % 7).astype(np.int32)             # 857 ids per bin
    with _ctx(pkg, sc, jh) as ctx:
        with pytest.raises(pkg.FspannStateError, match="treeified"):
            ctx.build_index(sc["X"])
        with pytest.raises(pkg.FspannStateError, match="not finalized"):
            ctx.route(np.zeros((1, 2, 1), np.uint64))
    # exactly at the threshold: 8 ids in one bin is still a chain, the 9th treeifies
    for k, ok in ((8, True), (9, False)):
        jh = np.arange(n).astype(np.int32) * 65536          # spread(h) & (cap-1): bin = h >> 16 ... distinct bins
        jh[:k] = 5 * 65536
        with _ctx(pkg, sc, jh) as ctx:
            if ok:
                ctx.build_index(sc["X"])
            else:
                with pytest.raises(pkg.FspannStateError, match="treeified"):
                    ctx.build_index(sc["X"])


@pytest.mark.parametrize("hard_cap,B", [(20000, 64), (700, 64), (100, 40), (40, 33)])
def test_route_flags_agree_with_literal_hashmap(pkg, oracle, hard_cap, B):
    """Random scenes whose hashCodes crowd a few bins.  HARD_CAP = 20000: one capacity stage, counters folded over the
    hash table; 700: one stage or two; 100 / 40: the map resizes once or twice while it fills (PIS:612-619)."""
    flagged_total = clean_total = 0
    for seed in range(6):
        rng = np.random.default_rng(900 + seed)
        n = int(rng.integers(1500, 6000))
        sc = make_scene(oracle, n=n, d=12, T=4, D=2, m=8, lam=2, B=B, hard_cap=hard_cap, seed=40 + seed)
        o = sc["oracle"]
        V = [60, 400, 2000, 10**6, 150, 10**6][seed]       # crowded bins ... well-spread hashCodes
        jh = (rng.integers(0, V, n) * int(rng.choice([1, 64, 4096, 65536 + 17]))).astype(np.int32)
        o.set_id_meta(n, jh)
        o.build_index(sc["X64"])              # partitions of the literal model (their tie order may be unmodelled: imported as data)
        Q = rng.standard_normal((32, 12))
        codes = o.encode(Q)
        for probes in (-1, 10):
            want = o.route_treeified(codes, probe_override=probes)
            ids, score, count, raw = o.route(codes, probe_override=probes)
            with _ctx(pkg, sc, jh, hard_cap=hard_cap, B=B) as ctx:
                _import(ctx, o)
                if want.any():
                    with pytest.raises(pkg.FspannStateError, match="treeified"):
                        ctx.route(codes, probe_override=probes)
                    assert ctx.unmodelled_queries() == int(want.sum())
                res = ctx.route(codes, probe_override=probes, allow_unmodelled=True)
                assert ctx.unmodelled_queries() == int(want.sum())
                assert ctx.unmodelled_queries() == 0          # reset by the previous call
            got = res["count"] < 0
            assert np.array_equal(got, want), (seed, probes, np.flatnonzero(got != want))
            for i in np.flatnonzero(~want):                   # everything else is still the reference's list
                assert res["count"][i] == count[i]
                assert np.array_equal(res["ids"][i, :count[i]], ids[i, :count[i]]), (seed, probes, i)
                assert np.array_equal(res["score"][i, :count[i]], score[i, :count[i]])
            flagged_total += int(want.sum())
            clean_total += int((~want).sum())
    assert flagged_total > 0 and clean_total > 0, (flagged_total, clean_total)


def test_bounded_select_hands_treeified_groups_to_the_full_select(pkg, oracle):
    """All ids in five bins: every (score, bin) group of the bounded select holds >= 9 entries -> handed back -> flagged."""
    n = 30000
    sc = make_scene(oracle, n=n, d=16, T=8, D=1, m=12, lam=2, B=256, seed=24)
    o = sc["oracle"]
    jh = (np.arange(n) % 5).astype(np.int32)
    codes = o.encode(sc["rng"].standard_normal((24, 16)))
    with _ctx(pkg, sc, jh) as ctx:
        _import(ctx, o)                                        # partitions cut with the stock hashCodes
        ctx.set_route_mode(2)
        lazy = ctx.route(codes, limit=256, counters=False, allow_unmodelled=True)
        info = ctx.last_route_info()
        assert info["lazy"] and info["overflowed"] == 24
        assert (lazy["count"] == -1).all()
        assert ctx.unmodelled_queries() == 24


def test_search_call_returns_nothing_for_a_flagged_query(pkg, oracle):
    import torch
    n, d, B, K = 8000, 16, 64, 5
    sc = make_scene(oracle, n=n, d=d, T=4, D=1, m=10, lam=2, B=B, seed=77)
    o = sc["oracle"]
    Q = sc["rng"].standard_normal((8, d)).astype(np.float32)
    codes = o.encode(Q.astype(np.float64))
    ids, _, count, _ = o.route(codes)
    jh = oracle.decimal_hashes(n).copy()
    jh[ids[0, :12]] = 123456                                 # twelve of query 0's candidates share one hashCode
    o.set_id_meta(n, jh)
    want = o.route_treeified(codes)
    assert want[0] and not want.all()
    dev = torch.device("cuda", 0)
    with _ctx(pkg, sc, jh) as ctx:
        _import(ctx, o)
        ctx.store_set(sc["X"])
        ctx.set_route_mode(1)                                # the full select: exact detection
        qd = torch.from_numpy(Q).to(dev)
        oi = torch.zeros((8, K), dtype=torch.int32, device=dev)
        od = torch.zeros((8, K), dtype=torch.float64, device=dev)
        oc = torch.zeros(8, dtype=torch.int32, device=dev)
        scn = torch.zeros(8, dtype=torch.int32, device=dev)
        sel = torch.zeros((8, B), dtype=torch.int32, device=dev)
        selc = torch.zeros(8, dtype=torch.int32, device=dev)
        torch.cuda.synchronize()
        ctx.search_store_dev(8, qd.data_ptr(), pkg._native.F32, -1, B, K, oi.data_ptr(), od.data_ptr(), oc.data_ptr(), scn.data_ptr(),
                             sel.data_ptr(), selc.data_ptr())
        ctx.sync()
        assert ctx.unmodelled_queries() == int(want.sum())
        selc, oc = selc.cpu().numpy(), oc.cpu().numpy()
    assert np.array_equal(selc < 0, want)
    assert (oc[want] == 0).all() and (oc[~want] == K).all()
