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


@pytest.mark.parametrize("mode", [0, 2, 1])
def test_search_call_is_completed_for_flagged_queries(pkg, oracle, mode):
    """fspann_search_store_dev leaves a flagged query empty (count -1, nothing scored); fspann_search_store_finish_dev finishes its
    Route on the host and scores the batch again: every query then equals oracle.search.  In auto mode (0) and with the bounded
    select forced (2) the flag comes from the bounded select's own exact check over bin16 (opaque ids) -> hand-over -> full
    select; mode 1 is the full select alone."""
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
        ctx.set_route_mode(mode)
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


def _crowd(oracle, o, sc, codes, q, k, rng, positions=None):
    """hashCodes = decimal ones, except k of query q's candidates — spread over its WHOLE list, every score level, far beyond the
    first B — which get k different hashCodes falling into ONE bin of the 32768-table.  Returns (jh, the crowded ids)."""
    n = sc["params"]["n"]
    ids, _, count, _ = o.route(codes)
    pos = np.unique(np.linspace(0, count[q] - 1, k).astype(int)) if positions is None else np.asarray(positions)
    crowd = ids[q, pos]
    jh = oracle.decimal_hashes(n).copy()
    h = jh.view(np.uint32)
    free = np.setdiff1d(np.arange(32768), (h ^ (h >> 16)) & 32767)       # a bin no decimal hashCode of this id space falls into
    b = int(free[rng.integers(0, len(free))])
    jh[crowd] = _spread_inv(b + 32768 * (1 + rng.permutation(len(crowd)).astype(np.int64)))
    return jh, crowd


def _max_bin_load(jh, lists, counts):
    """Per query: the largest number of (distinct, live — the lists hold nothing else) ids sharing one bin of the 32768-table."""
    h = jh.view(np.uint32)
    sp = (h ^ (h >> 16)) & 32767
    return np.array([np.unique(sp[l[:c]], return_counts=True)[1].max() for l, c in zip(lists, counts)])


@pytest.mark.parametrize("mode", [0, 2])
def test_bounded_select_sees_a_bin_filled_by_ids_it_never_loads(pkg, oracle, mode):
    """The hole VERDICT r03 named: opaque ids (caller-supplied String.hashCode) whose codes crowd ONE bin across DIFFERENT scores
    and partitions the bounded select never loads.  The JVM treeifies that bin and reorders it; the bounded select's exact check
    (bin16: the bins of ALL ids of the probed partitions) must flag exactly those queries in auto mode and in mode 2 — and
    fspann_route's answer (full select -> host model for them) is the oracle's list for EVERY query.  Eight in a bin is a chain
    (not flagged), nine a tree; a deleted id is never put (PIS:739) and does not count."""
    B = 64
    total_flagged = total_clean = 0
    for seed, k, ndel in ((0, 12, 0), (1, 10, 0), (2, 8, 0), (3, 10, 2), (4, 16, 0), (5, 40, 0)):
        rng = np.random.default_rng(4100 + seed)
        n = 20000
        sc = make_scene(oracle, n=n, d=16, T=8, D=1, m=12, lam=2, B=B, seed=300 + seed)
        o = sc["oracle"]
        Q = rng.standard_normal((12, 16))
        codes = o.encode(Q)
        jh, crowd = _crowd(oracle, o, sc, codes, 3, k, rng)
        deleted = None
        if ndel:
            deleted = np.zeros(n, np.uint8)
            deleted[crowd[[1, len(crowd) // 2]]] = 1         # two of the ten are deleted: eight puts, no tree
        o.set_id_meta(n, jh, deleted)
        o.build_index(sc["X64"])                             # (the staging map's order depends on the hashCodes)
        ids, score, count, _ = o.route(codes)
        want = o.route_treeified(codes)
        assert not o.unmodelled
        # what this scene is for: the crowded ids of query 3 sit on several score levels, most of them behind the first B entries
        # (re-cutting the partitions with the new hashCodes can move a crowded id out of reach: what counts is what is in the list)
        where = np.flatnonzero(np.isin(ids[3, :count[3]], crowd))
        load = _max_bin_load(jh, ids, count)
        assert np.array_equal(want, load >= 9), (seed, load)            # the oracle's own flag, re-derived from first principles
        assert load[3] == len(where) and len(np.unique(score[3, where])) >= 3 and (where < B).sum() < 9, (seed, where)
        with _ctx(pkg, sc, jh, B=B) as ctx:
            if deleted is not None:
                ctx.set_id_meta(n, jh, deleted)
            _import(ctx, o)
            ctx.set_route_mode(mode)
            res = ctx.route(codes, limit=B, counters=False)
            info = ctx.last_route_info()
            assert info["lazy"], "the bounded select must be the one that ran"
            assert info["overflowed"] >= int(want.sum())     # every query the JVM would treeify was handed over
            assert ctx.unmodelled_queries() == 0             # ... and finished by the host model inside fspann_route
            # the bounded select alone (device entry point, nothing resolved): flagged queries == the oracle's
            flags = ctx.route_flags_bounded(codes, limit=B)
        assert np.array_equal(flags, want), (seed, np.flatnonzero(flags != want))
        for i in range(len(Q)):
            c = min(count[i], B)
            assert res["count"][i] == c, (seed, i)
            assert np.array_equal(res["ids"][i, :c], ids[i, :c]), (seed, i, bool(want[i]))
            assert np.array_equal(res["score"][i, :c], score[i, :c]), (seed, i)
        total_flagged += int(want.sum())
        total_clean += int((~want).sum())
    assert total_flagged >= 3 and total_clean > 0


def test_decimal_ids_can_ask_for_the_exact_check(pkg, oracle, monkeypatch):
    """FSPANN_ROUTE_BINCHECK=1: decimal ordinals run the same exact check (bin16 is built for them too); lists unchanged."""
    monkeypatch.setenv("FSPANN_ROUTE_BINCHECK", "1")
    sc = make_scene(oracle, n=30000, d=16, T=8, D=2, m=12, lam=2, B=256, seed=5)
    o = sc["oracle"]
    codes = o.encode(sc["rng"].standard_normal((40, 16)))
    ids, score, count, _ = o.route(codes)
    with _ctx(pkg, sc, None, B=256) as ctx:
        _import(ctx, o)
        res = ctx.route(codes, limit=256, counters=False)
        assert ctx.last_route_info()["lazy"]
    for i in range(40):
        c = min(count[i], 256)
        assert res["count"][i] == c and np.array_equal(res["ids"][i, :c], ids[i, :c]) and np.array_equal(res["score"][i, :c], score[i, :c])


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
