"""Queries (and Setups) whose java.util.HashMap would treeify a bin are ANSWERED, not refused.

The Route kernels derive the iteration order of HashMap<String,Long> bestScore (PIS:619,690-693) in closed form — bin at the
final table length, then first insertion — which is the JVM's order only while every bin is a plain chain.  A put that finds 8
nodes in its bin (table >= 64) makes the JVM treeify it; from then on the bin iterates through TreeNode's `next` list.  The
full select DETECTS that exactly (count = -1, per capacity stage of the map) and the library finishes such a query on the host
with a literal model of the JDK's tree bins (host/java_hashmap.hpp + host/route_replay.hpp): `fspann_route` by itself,
`fspann_route_resolve_dev` / `fspann_search_store_finish_dev` behind the asynchronous entry points.  Compared here, query by
query, with the oracle's independently written model (oracle/fspann_oracle.cpp JHashMap): detection flags AND the lists.
What stays refused: a tree bin that has to order different ids with EQUAL String.hashCode when the ids are not decimal
ordinals (String.compareTo of Strings the library never sees).  PIS:413 + idx/GreedyPartitioner.java:45-51 for the build.
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


def _spread_inv(s):
    """String.hashCode values whose HashMap.hash() spread is `s` (h ^ h >>> 16 is an involution)."""
    s = np.asarray(s).astype(np.uint32)
    return (s ^ (s >> 16)).view(np.int32)


def _crowded_distinct_hashes(rng, n, nbins, cap):
    """n DIFFERENT hashCodes that fall into `nbins` bins of a table of length `cap` (and keep colliding partly as it doubles)."""
    bins = rng.choice(cap, nbins, replace=False)
    return _spread_inv(bins[rng.integers(0, nbins, n)] + cap * rng.permutation(n).astype(np.int64) % (1 << 31))


def test_build_index_orders_a_treeified_staging_map_like_the_jdk(pkg, oracle):
    """HashMap<String,BitSet>(staged.size()) with crowded bins: the GPU cut takes the map's iteration order from the host model
    (and the host cut, FSPANN_GPU_CUT=0, too): every table equals the oracle's, which iterates its own literal tree bins."""
    n = 6000
    sc = make_scene(oracle, n=n, d=8, T=2, D=1, m=8, lam=2, B=64, seed=3)
    o = sc["oracle"]
    rng = np.random.default_rng(1)
    jh = _crowded_distinct_hashes(rng, n, 9, 8192)           # ~670 ids per bin: trees, splits and untreeified halves on the way
    o.set_id_meta(n, jh)
    o.build_index(sc["X64"])
    assert o.treeified and not o.unmodelled
    for cut in ("1", "0"):
        import os
        os.environ["FSPANN_GPU_CUT"] = cut
        try:
            with _ctx(pkg, sc, jh) as ctx:
                ctx.build_index(sc["X"])
                for td in range(o.TD):
                    a, b = ctx.get_index(td), o.get_index(td)
                    for key in ("min_key", "max_key", "rep", "id_off", "ids"):
                        assert np.array_equal(a[key], b[key]), (cut, td, key)
        finally:
            del os.environ["FSPANN_GPU_CUT"]
    # exactly at the threshold: 8 ids in one bin is still a chain, the 9th treeifies — both build, both equal the oracle's
    for k in (8, 9):
        jh = np.arange(n).astype(np.int32) * 65536           # spread = (i << 16) ^ i: bin i of the 8192-table, all different
        j = np.arange(k)
        jh[:k] = (j << 16) | (5 ^ j)                         # k different hashCodes whose spread ends in ...0101: ONE bin (5)
        o.set_id_meta(n, jh)
        o.build_index(sc["X64"])
        assert o.treeified == (k == 9) and not o.unmodelled
        with _ctx(pkg, sc, jh) as ctx:
            ctx.build_index(sc["X"])
            for td in range(o.TD):
                assert np.array_equal(ctx.get_index(td)["ids"], o.get_index(td)["ids"]), (k, td)
    # EQUAL hashCodes of non-decimal ids inside a tree bin: String.compareTo is unknown -> refused, loudly
    jh = (np.arange(n) % 7).astype(np.int32)
    with _ctx(pkg, sc, jh) as ctx:
        with pytest.raises(pkg.FspannStateError, match="EQUAL String.hashCode"):
            ctx.build_index(sc["X"])
        with pytest.raises(pkg.FspannStateError, match="not finalized"):
            ctx.route(np.zeros((1, 2, 1), np.uint64))


@pytest.mark.parametrize("hard_cap,B", [(20000, 64), (700, 64), (100, 40), (40, 33)])
def test_treeified_queries_get_the_jdk_order(pkg, oracle, hard_cap, B):
    """Random scenes whose (distinct) hashCodes crowd a few bins.  HARD_CAP = 20000: one capacity stage; 700: one or two;
    100 / 40: the map resizes while it fills (PIS:612-619): trees split, halves untreeify.  The GPU's detection must agree with the
    literal model query by query, and fspann_route's answer — host replay for the flagged ones — must be the oracle's list."""
    flagged_total = clean_total = 0
    for seed in range(6):
        rng = np.random.default_rng(900 + seed)
        n = int(rng.integers(1500, 6000))
        sc = make_scene(oracle, n=n, d=12, T=4, D=2, m=8, lam=2, B=B, hard_cap=hard_cap, seed=40 + seed)
        o = sc["oracle"]
        cap0 = oracle.table_size_for(min(max(hard_cap, B), 1 << 16))
        nb = [3, 20, 200, cap0, 8, cap0][seed]                 # crowded bins ... well-spread hashCodes
        jh = _crowded_distinct_hashes(rng, n, min(nb, cap0), cap0)
        o.set_id_meta(n, jh)
        o.build_index(sc["X64"])
        Q = rng.standard_normal((32, 12))
        codes = o.encode(Q)
        for probes in (-1, 10):
            want = o.route_treeified(codes, probe_override=probes)
            ids, score, count, raw = o.route(codes, probe_override=probes)
            assert not o.unmodelled
            with _ctx(pkg, sc, jh, hard_cap=hard_cap, B=B) as ctx:
                _import(ctx, o)
                res = ctx.route(codes, probe_override=probes)            # finishes flagged queries by itself
                assert ctx.unmodelled_queries() == 0
                flags = ctx.route_flags(codes, probe_override=probes)     # what the kernels flagged before the host model ran
            assert np.array_equal(flags, want), (seed, probes, np.flatnonzero(flags != want))
            for i in range(len(Q)):
                assert res["count"][i] == count[i], (seed, probes, i, bool(want[i]))
                assert np.array_equal(res["ids"][i, :count[i]], ids[i, :count[i]]), (seed, probes, i, bool(want[i]))
                assert np.array_equal(res["score"][i, :count[i]], score[i, :count[i]])
                assert res["kept"][i] == count[i] and res["raw_seen"][i] == raw[i]
            flagged_total += int(want.sum())
            clean_total += int((~want).sum())
    assert flagged_total > 0 and clean_total > 0, (flagged_total, clean_total)


def test_equal_hashcodes_of_opaque_ids_stay_refused(pkg, oracle):
    """Non-decimal ids with EQUAL hashCodes crowding a tree bin: the JVM orders them by String.compareTo, which nobody outside
    the JVM can evaluate here -> those queries keep count = -1 and fspann_route fails loudly; the others are answered."""
    rng = np.random.default_rng(5)
    n = 4000
    sc = make_scene(oracle, n=n, d=12, T=4, D=2, m=8, lam=2, B=64, seed=12)
    o = sc["oracle"]
    jh = (rng.integers(0, 60, n)).astype(np.int32)
    o.set_id_meta(n, jh)
    o.build_index(sc["X64"])
    codes = o.encode(rng.standard_normal((16, 12)))
    want = o.route_treeified(codes)
    assert want.any() and o.unmodelled
    with _ctx(pkg, sc, jh) as ctx:
        _import(ctx, o)
        with pytest.raises(pkg.FspannStateError, match="String.compareTo"):
            ctx.route(codes)
        assert ctx.unmodelled_queries() == int(want.sum())      # (and reset)
        res = ctx.route(codes, allow_unmodelled=True)
        assert np.array_equal(res["count"] < 0, want)
        assert ctx.unmodelled_queries() == int(want.sum())


def test_bounded_select_hand_over_ends_in_the_host_model(pkg, oracle):
    """Every id in five bins (distinct hashCodes): every (score, bin) group of the bounded select holds >= 9 entries -> handed back
    -> flagged by the full select -> finished by the host model: the first 256 entries are the oracle's."""
    n = 30000
    sc = make_scene(oracle, n=n, d=16, T=8, D=1, m=12, lam=2, B=256, seed=24)
    o = sc["oracle"]
    jh = _crowded_distinct_hashes(np.random.default_rng(2), n, 5, 32768)
    o.set_id_meta(n, jh)
    o.build_index(sc["X64"])
    codes = o.encode(sc["rng"].standard_normal((24, 16)))
    ids, score, count, _ = o.route(codes)
    assert o.route_treeified(codes).all() and not o.unmodelled
    with _ctx(pkg, sc, jh) as ctx:
        _import(ctx, o)
        ctx.set_route_mode(2)
        lazy = ctx.route(codes, limit=256, counters=False)
        info = ctx.last_route_info()
        assert info["lazy"] and info["overflowed"] == 24
        assert ctx.unmodelled_queries() == 0
    assert (lazy["count"] == np.minimum(count, 256)).all()
    for i in range(24):
        assert np.array_equal(lazy["ids"][i, :lazy["count"][i]], ids[i, :lazy["count"][i]]), i


def test_search_call_is_completed_for_flagged_queries(pkg, oracle):
    """fspann_search_store_dev leaves a flagged query empty (count -1, nothing scored); fspann_search_store_finish_dev finishes its
    Route on the host and scores the batch again: every query then equals oracle.search."""
    import torch
    n, d, B, K = 8000, 16, 64, 5
    sc = make_scene(oracle, n=n, d=d, T=4, D=1, m=10, lam=2, B=B, seed=77)
    o = sc["oracle"]
    Q = sc["rng"].standard_normal((8, d)).astype(np.float32)
    codes = o.encode(Q.astype(np.float64))
    ids, _, count, _ = o.route(codes)
    jh = oracle.decimal_hashes(n).copy()
    jh[ids[0, :12]] = _spread_inv(777 + 32768 * np.arange(1, 13))        # twelve of query 0's candidates in ONE bin, different hashCodes
    o.set_id_meta(n, jh)
    o.build_index(sc["X64"])
    codes = o.encode(Q.astype(np.float64))
    want = o.route_treeified(codes)
    assert want.any() and not want.all()
    ref = o.search(Q.astype(np.float64), K)
    assert not o.unmodelled
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
        args = (8, qd.data_ptr(), pkg._native.F32, -1, B, K, oi.data_ptr(), od.data_ptr(), oc.data_ptr(), scn.data_ptr(), sel.data_ptr(), selc.data_ptr())
        ctx.search_store_dev(*args)
        ctx.sync()
        assert ctx.unmodelled_queries(reset=False) == int(want.sum())
        assert np.array_equal(selc.cpu().numpy() < 0, want) and (oc.cpu().numpy()[want] == 0).all()
        done = ctx.search_store_finish_dev(*args)
        ctx.sync()
        assert done == int(want.sum()) and ctx.unmodelled_queries() == 0
        assert np.array_equal(selc.cpu().numpy(), ref["sel_count"])
        assert np.array_equal(np.where(np.arange(B)[None] < ref["sel_count"][:, None], sel.cpu().numpy(), -1), ref["sel"][:, :B])
        assert np.array_equal(oi.cpu().numpy(), ref["ids"]) and np.array_equal(od.cpu().numpy(), ref["dist"])
        assert np.array_equal(oc.cpu().numpy(), ref["count"])


def test_host_pipeline_answers_flagged_queries(pkg, oracle):
    """The native host pipeline (Route | AES-GCM open | Refine) with hashCodes crowded into five bins: every query's map treeifies.
    Stage A resolves the flagged queries with the host model before their candidate ids go to the decrypt threads — the results
    equal oracle.search, nothing is reported unmodelled, no query comes back empty."""
    from fspann_amd import hostpipe
    n, d, B, K = 20000, 16, 64, 5
    sc = make_scene(oracle, n=n, d=d, T=6, D=1, m=12, lam=2, B=B, seed=91)
    o = sc["oracle"]
    jh = _crowded_distinct_hashes(np.random.default_rng(5), n, 5, oracle.table_size_for(20000))
    o.set_id_meta(n, jh)
    o.build_index(sc["X64"])
    o.set_store(sc["X64"])
    batches = [sc["rng"].standard_normal((nq, d)).astype(np.float32) for nq in (40, 9, 40)]
    assert o.route_treeified(o.encode(batches[0].astype(np.float64))).all()
    with _ctx(pkg, sc, jh) as ctx, hostpipe.PointStore(n, d) as ps:
        _import(ctx, o)
        ps.encrypt(sc["X"], threads=8)
        with hostpipe.Pipeline(ctx, ps, 40, B, K, host_threads=8) as pl:
            out = []
            for qb in batches:
                pl.submit(qb)
            while pl.in_flight:
                out.append(pl.collect())
        assert ctx.unmodelled_queries() == 0
    for qb, res in zip(batches, out):
        ref = o.search(qb.astype(np.float64), K)
        assert (res["count"] == ref["count"]).all() and (res["count"] > 0).all()
        assert np.array_equal(res["ids"], ref["ids"]) and np.array_equal(res["dist"], ref["dist"])
    assert not o.unmodelled
