"""GPU parity, Route edge cases: every rarely-taken path of the route kernels against the oracle.

  * HARD_CAP triggering mid-table / mid-probe (PIS:612-615,624,628,657-659)
  * deleted ids (PIS:736-742)
  * heavy repeats across tables (clustered data: > kDupListMax repeats -> table-phase path)
  * tiny tables (fewer partitions than probes), one-partition tables
  * many probes (retry P = 10, fallback P = 20), W = 2..3 code words
  * per-query scratch that does not fit LDS (global arena path)
  * arbitrary String ids (java_hash given explicitly)
"""
import numpy as np
import pytest

from conftest import make_scene

pytestmark = pytest.mark.gpu

LAZY_RUNS = []   # last_route_info() of every bounded-select attempt made by check_route


def ctx_for(pkg, sc, java_hash=None, **over):
    p = dict(sc["params"])
    p.update(over)
    cfg = pkg.PaperRuntimeConfig(tables=p["T"], divisions=p["D"], m=p["m"], lambda_=p["lam"], dim=p["d"],
                                 refinement_limit=p["B"], max_global_candidates=p["hard_cap"],
                                 probe_override=p["probe_override"], hamming_prefilter_threshold=p["tau"])
    ctx = pkg.FspannContext(cfg, 0)
    ctx.set_gfunctions(sc["alpha"], sc["r"], sc["omega"])
    ctx.set_id_meta(p["n"], java_hash, sc["deleted"])
    return ctx


def check_route(pkg, sc, nq=16, probes=-1, limits=(None,), java_hash=None, import_index=False):
    o = sc["oracle"]
    p = sc["params"]
    Q = sc["rng"].standard_normal((nq, p["d"])).astype(np.float32).astype(np.float64)
    if p.get("clustered"):
        Q = 5 + 0.1 * Q
    codes = o.encode(Q)
    ids, score, count, raw = o.route(codes, probe_override=probes)
    assert not o.unmodelled, "oracle HashMap treeified: order not pinned for this scene"
    with ctx_for(pkg, sc, java_hash=java_hash) as ctx:
        if import_index:
            for td in range(o.TD):
                ctx.set_index(td, **o.get_index(td))
            ctx.finalize()
        else:
            ctx.build_index(sc["X"])
        for lim in limits:
            res = ctx.route(codes, probe_override=probes, limit=lim if lim else 2**31 - 1)
            assert np.array_equal(res["kept"], count)
            assert np.array_equal(res["raw_seen"], raw)
            for i in range(nq):
                n = count[i] if not lim else min(lim, count[i])
                assert res["count"][i] == n
                assert np.array_equal(res["ids"][i, :n], ids[i, :n]), (lim, i)
                assert np.array_equal(res["score"][i, :n], score[i, :n]), (lim, i)
            if lim and lim <= 1024:
                # the bounded select (no counters requested): same first `lim` entries wherever it is legal
                ctx.set_route_mode(2)
                res2 = ctx.route(codes, probe_override=probes, limit=lim, counters=False)
                ctx.set_route_mode(0)
                LAZY_RUNS.append(ctx.last_route_info())
                assert np.array_equal(res2["count"], res["count"]), lim
                for i in range(nq):
                    n = res["count"][i]
                    assert np.array_equal(res2["ids"][i, :n], ids[i, :n]), ("bounded", lim, i)
                    assert np.array_equal(res2["score"][i, :n], score[i, :n]), ("bounded", lim, i)
    return count, raw


def test_hard_cap_triggers(pkg, oracle):
    # T*D*P*S = 6*1*5*64 = 1920 tuples against HARD_CAP = 700: stops in the middle of a table
    sc = make_scene(oracle, n=6000, d=16, T=6, D=1, m=10, lam=2, B=100, hard_cap=700, seed=5)
    count, _ = check_route(pkg, sc, limits=(None, 100, 700))
    assert count.max() >= 700 and count.max() <= 700 + 63
    # cap a few ids above a multiple of the block size / exactly at the first probe
    for cap in (64, 65, 129, 320, 321):
        sc = make_scene(oracle, n=3000, d=12, T=4, D=2, m=8, lam=2, B=40, hard_cap=cap, seed=cap)
        check_route(pkg, sc, limits=(None, 40))


def test_hard_cap_with_probe_overrides(pkg, oracle):
    sc = make_scene(oracle, n=8000, d=16, T=4, D=2, m=12, lam=2, B=256, hard_cap=1500, seed=9)
    for probes in (1, 3, 10, 20):
        check_route(pkg, sc, probes=probes, limits=(None, 256))


def test_deleted_ids(pkg, oracle):
    sc = make_scene(oracle, n=5000, d=16, T=4, D=2, m=10, lam=2, B=128, deleted_frac=0.3, seed=11)
    check_route(pkg, sc, limits=(None, 128))
    sc = make_scene(oracle, n=2000, d=8, T=3, D=1, m=8, lam=2, B=64, deleted_frac=0.97, seed=12)
    check_route(pkg, sc, limits=(None, 64))


def test_heavy_repeats_clustered(pkg, oracle):
    # tight cluster: all tables probe nearly the same ids -> thousands of repeated occurrences
    sc = make_scene(oracle, n=1024, d=8, T=4, D=4, m=4, lam=3, B=128, clustered=True, seed=13)
    sc["params"]["clustered"] = True
    count, raw = check_route(pkg, sc, limits=(None, 128, 10))
    assert (raw > count).any()


def test_heavy_repeats_same_gfunction(pkg, oracle):
    # identical tables: every id repeats T*D times with equal scores (no strict improvement)
    sc = make_scene(oracle, n=3000, d=10, T=3, D=2, m=8, lam=2, B=64, seed=14)
    a, r, w = sc["alpha"], sc["r"], sc["omega"]
    a[:] = a[0]
    r[:] = r[0]
    w[:] = w[0]
    o = sc["oracle"]
    o.set_gfunctions(a, r, w)
    o.build_index(sc["X64"])
    count, raw = check_route(pkg, sc, limits=(None, 64))
    assert np.array_equal(count, raw)  # repeats never improve


def test_repeat_list_capacity_boundary(pkg, oracle):
    # 12 tables x 5 probes x 64 = 3 840 tuples over 11-20 k ids: 500-1 000 repeats per query, on both sides of the repeat list's capacity
    # (kDupListMax = 768: the list walked in LDS below it, table by table above it) — rawSeen counts every strict improvement either way
    seen = []
    for n, seed in ((11000, 61), (13000, 62), (15000, 63), (17000, 64), (20000, 65)):
        sc = make_scene(oracle, n=n, d=12, T=4, D=3, m=10, lam=2, B=512, seed=seed)
        count, raw = check_route(pkg, sc, nq=24, limits=(None, 512))
        seen += list(3840 - count)
    seen = np.array(seen)
    assert (seen <= 768).sum() >= 8 and (seen > 768).sum() >= 8, (seen.min(), seen.max())


def test_tiny_tables(pkg, oracle):
    for n in (1, 5, 64, 65, 130):
        sc = make_scene(oracle, n=n, d=4, T=2, D=2, m=4, lam=2, B=33, seed=20 + n)
        check_route(pkg, sc, nq=6, limits=(None, 33, 1))
        check_route(pkg, sc, nq=6, probes=10, limits=(None,))


def test_wide_codes(pkg, oracle):
    sc = make_scene(oracle, n=4000, d=20, T=2, D=2, m=40, lam=3, B=100, seed=31)   # 120 bits -> W = 2
    check_route(pkg, sc, limits=(None, 100))
    sc = make_scene(oracle, n=3000, d=20, T=2, D=1, m=80, lam=2, B=100, seed=32)   # 160 bits -> W = 3
    check_route(pkg, sc, limits=(None, 100))
    sc = make_scene(oracle, n=3000, d=12, T=2, D=2, m=31, lam=2, B=50, seed=33)    # 62 bits
    check_route(pkg, sc, limits=(None, 50))
    sc = make_scene(oracle, n=3000, d=12, T=2, D=2, m=32, lam=2, B=50, seed=34)    # 64 bits: bit 63 is not in the key
    check_route(pkg, sc, limits=(None, 50))


def test_global_arena_path(pkg, oracle):
    # 10 tables x 4 divisions x 20 probes x 64 = 51200 tuple slots: per-query scratch exceeds LDS
    sc = make_scene(oracle, n=20000, d=16, T=10, D=4, m=12, lam=2, B=512, hard_cap=60000, seed=41)
    count, _ = check_route(pkg, sc, nq=6, probes=20, limits=(None, 512))
    assert count.max() > 8192
    # same with a HARD_CAP cut inside
    sc = make_scene(oracle, n=20000, d=16, T=10, D=4, m=12, lam=2, B=512, hard_cap=9000, seed=42)
    check_route(pkg, sc, nq=6, probes=20, limits=(None, 512))


def test_long_lists_sorted_in_lds_and_global(pkg, oracle):
    # full lists of ~3-4k entries (LDS bitonic) and ~20k entries (global bitonic)
    sc = make_scene(oracle, n=20000, d=16, T=8, D=1, m=12, lam=2, B=4000, seed=51)
    check_route(pkg, sc, nq=8, probes=10, limits=(None, 4000, 1500))
    sc = make_scene(oracle, n=60000, d=16, T=8, D=4, m=14, lam=2, B=20000, hard_cap=30000, seed=52)
    check_route(pkg, sc, nq=4, probes=10, limits=(None, 20000))


def test_string_ids_via_java_hash(pkg, oracle):
    """Arbitrary String ids: the adapter hands String.hashCode per handle (fspann_set_id_meta)."""
    n = 4000
    sc = make_scene(oracle, n=n, d=12, T=3, D=2, m=10, lam=2, B=64, seed=61)
    names = ["vec-%05x-%d" % (i * 2654435761 % (1 << 20), i) for i in range(n)]
    jh = np.array([oracle.string_hash(s) for s in names], dtype=np.int32)
    o = sc["oracle"]
    o.set_id_meta(n, jh, None)
    o.build_index(sc["X64"])
    check_route(pkg, sc, limits=(None, 64), java_hash=jh)


def test_imported_index_equals_native_build(pkg, oracle):
    sc = make_scene(oracle, n=5000, d=16, T=3, D=3, m=10, lam=2, B=99, seed=71)
    check_route(pkg, sc, limits=(None, 99), import_index=True)


def test_batch_larger_than_grid(pkg, oracle):
    sc = make_scene(oracle, n=3000, d=8, T=2, D=2, m=8, lam=2, B=32, seed=81)
    check_route(pkg, sc, nq=1500, limits=(32,))


def _rehash(sc, jh):
    o = sc["oracle"]
    o.set_id_meta(sc["params"]["n"], jh, None)
    o.build_index(sc["X64"])       # GreedyPartitioner's input order is a HashMap iteration: it depends on the hashes


def test_bounded_select_bucket_collisions(pkg, oracle):
    """Survivors that share (score, HashMap bucket) are ordered by the reference's first insertion, recomputed from
    the inverse id map.  (<= 6 ids per hash value: the reference HashMap does not treeify, the order stays modelled.)"""
    n = 30000
    jh = (np.arange(n) % 5000).astype(np.int32)
    for seed, clustered in ((21, False), (22, True)):
        sc = make_scene(oracle, n=n, d=16, T=8, D=1, m=12, lam=2, B=256, seed=seed, clustered=clustered)
        sc["params"]["clustered"] = clustered
        _rehash(sc, jh)
        before = len(LAZY_RUNS)
        check_route(pkg, sc, nq=24, limits=(256, 100, 17), java_hash=jh)
        assert all(r["lazy"] for r in LAZY_RUNS[before:])


def test_bounded_select_many_collisions_hands_back(pkg, oracle):
    """More entries sharing (score, HashMap bin) than the bounded select settles itself (kLzCollMax) -> the full select
    redoes the query.  The hashCodes are laid out per query so that every bin holds 3 of its candidates: far from the 9
    that would treeify a bin (tests/test_gpu_treeify.py covers that side), so the oracle's order is the reference's."""
    n = 20000
    sc = make_scene(oracle, n=n, d=16, T=1, D=1, m=12, lam=2, B=300, seed=24)
    o = sc["oracle"]
    Q = sc["rng"].standard_normal((3, 16))
    codes = o.encode(Q)
    base = oracle.decimal_hashes(n).astype(np.int64)
    for qi in range(3):
        ids0, _, cnt0, _ = o.route(codes[qi:qi + 1])
        jh = (base * 65536 + 40000).astype(np.int64)              # everything else: bins far away from the crafted ones
        cands = ids0[0, :cnt0[0]]
        jh[cands] = (np.arange(len(cands)) // 3) * 65536          # spread(h) & (cap-1) = h >> 16: bin j // 3 (the ~320 entries the select holds at limit 300 all collide: > kLzCollMax)
        jh = (jh & 0xFFFFFFFF).astype(np.uint32).view(np.int32)
        o.set_id_meta(n, jh)
        ids, score, count, raw = o.route(codes[qi:qi + 1])         # same partitions, new HashMap order
        assert not o.route_treeified(codes[qi:qi + 1]).any()
        with ctx_for(pkg, sc, java_hash=jh) as ctx:
            for td in range(o.TD):
                ctx.set_index(td, **o.get_index(td))
            ctx.finalize()
            full = ctx.route(codes[qi:qi + 1], limit=300)
            ctx.set_route_mode(2)
            lazy = ctx.route(codes[qi:qi + 1], limit=300, counters=False)
            info = ctx.last_route_info()
        assert info["lazy"] and info["overflowed"] == 1
        c = min(300, count[0])
        assert full["count"][0] == c and lazy["count"][0] == c
        assert np.array_equal(full["ids"][0, :c], ids[0, :c]) and np.array_equal(lazy["ids"][0, :c], ids[0, :c])
        assert np.array_equal(lazy["score"][0, :c], score[0, :c])
    o.set_id_meta(n)


@pytest.mark.parametrize("block_size", [16, 100, 128])
def test_bounded_select_other_block_sizes(pkg, oracle, block_size):
    """The reference's block size is the constant 64 (PIS:92); the library takes others.  Partitions that are not one
    wave wide go through the whole-partition path of the bounded select: compared with the full select."""
    sc = make_scene(oracle, n=20000, d=16, T=6, D=1, m=12, lam=2, B=200, seed=31 + block_size)
    p = sc["params"]
    codes = sc["oracle"].encode(sc["rng"].standard_normal((20, 16)))
    cfg = pkg.PaperRuntimeConfig(tables=p["T"], divisions=p["D"], m=p["m"], lambda_=p["lam"], dim=p["d"], refinement_limit=p["B"],
                                 block_size=block_size)
    with pkg.FspannContext(cfg, 0) as ctx:
        ctx.set_gfunctions(sc["alpha"], sc["r"], sc["omega"])
        ctx.set_id_meta(p["n"])
        ctx.build_index(sc["X"])
        for lim in (200, 33):
            full = ctx.route(codes, limit=lim)
            ctx.set_route_mode(2)
            lazy = ctx.route(codes, limit=lim, counters=False)
            info = ctx.last_route_info()
            ctx.set_route_mode(0)
            assert info["lazy"] and info["overflowed"] == 0
            assert np.array_equal(lazy["count"], full["count"])
            assert np.array_equal(lazy["ids"], full["ids"]) and np.array_equal(lazy["score"], full["score"])


def test_bounded_select_hands_back_large_queries(pkg, oracle, monkeypatch):
    """Queries whose entries exceed what the bounded select may hold are redone by the full select."""
    sc = make_scene(oracle, n=40000, d=16, T=10, D=1, m=12, lam=2, B=256, seed=23)
    monkeypatch.setenv("FSPANN_ROUTE_LAZY_CAP", "258")     # limit + 2: a cut that overshoots by a few ids no longer fits
    before = len(LAZY_RUNS)
    check_route(pkg, sc, nq=40, limits=(256,))
    assert LAZY_RUNS[before]["lazy"] and 0 < LAZY_RUNS[before]["overflowed"] <= 40
    monkeypatch.setenv("FSPANN_ROUTE_LAZY_CAP", "8")
    check_route(pkg, sc, nq=40, limits=(256, 5))
    assert LAZY_RUNS[-1]["overflowed"] > 0


@pytest.mark.parametrize("case", ["plain", "runs_of_equal_keys", "long_runs", "tiny", "far_queries", "clustered", "deleted", "two_divisions"])
def test_bounded_select_shape_specialised_build(pkg, oracle, case):
    """16 tables x 5 probes, one code word, blocks of 64, limit <= 256: the bounded select's build with the shape as compile-time
    constants (route_lazy.hip.h: kTD / kP).  The rarely-taken paths of the probe and the walk at exactly that shape: runs of equal
    keys spanning partitions (a < b: the replayed binary search picks the centre), runs longer than the window, tables with
    fewer partitions than the window, query keys below / above every partition and in gaps between partitions, deleted ids, two
    divisions per table — all against the oracle, full select and bounded select."""
    kw = dict(n=40000, d=16, T=16, D=1, m=12, lam=2, B=256, seed=900)
    scale = 1.0
    if case == "runs_of_equal_keys":
        kw.update(m=4, seed=901)             # 8-bit codes: ~156 points per key -> runs over 2-3 partitions
    elif case == "long_runs":
        kw.update(m=3, n=60000, seed=902)    # 6-bit codes: ~940 points per key -> runs of ~15 partitions, beyond the window
    elif case == "tiny":
        kw.update(n=300, B=64, seed=903)     # five partitions per table: fewer than the 11-record window
    elif case == "far_queries":
        kw.update(seed=904)
        scale = 40.0                         # saturated projections: keys at and beyond the tables' ends
    elif case == "clustered":
        kw.update(clustered=True, seed=905)
    elif case == "deleted":
        kw.update(deleted_frac=0.25, seed=906)
    elif case == "two_divisions":
        kw.update(T=8, D=2, seed=907)
    sc = make_scene(oracle, **kw)
    if scale != 1.0:
        rng0 = sc["rng"]

        class _Scaled:
            def standard_normal(self, shape):
                return rng0.standard_normal(shape) * scale
        sc["rng"] = _Scaled()
    sc["params"]["clustered"] = bool(kw.get("clustered"))
    lim = min(256, kw["B"])
    check_route(pkg, sc, nq=48, limits=(lim, 17))
    if case == "tiny":                       # one and two partitions per table
        for n in (40, 100):
            sc = make_scene(oracle, **dict(kw, n=n, seed=910 + n))
            check_route(pkg, sc, nq=16, limits=(64,))


def test_zz_bounded_select_was_exercised():
    """The scenes above must have driven the bounded select itself, and its hand-back to the full select."""
    assert sum(1 for r in LAZY_RUNS if r["lazy"]) >= 10
    assert any(r["lazy"] and r["overflowed"] == 0 for r in LAZY_RUNS)


def test_hamming_prefilter_threshold_is_order_equivalent(pkg, oracle):
    """runtime.hammingPrefilterThreshold tau > 0 (QSI:169-206): every candidate with score <= tau first, then the rest until B.
    The list is already stable-sorted by score, so stage A.5 keeps the same first B whatever tau is — asserted against the
    oracle's literal two-pass selection for tau in {4, 7} (the values of the reference's *_sub1.json profiles)."""
    for tau in (4, 7):
        sc = make_scene(oracle, n=20000, d=16, T=6, D=2, m=12, lam=2, B=150, tau=tau, seed=90 + tau)
        o = sc["oracle"]
        Q = sc["rng"].standard_normal((32, 16)).astype(np.float32)
        ref = o.search(Q.astype(np.float64), 10)
        codes = o.encode(Q.astype(np.float64))
        with ctx_for(pkg, sc) as ctx:
            ctx.build_index(sc["X"])
            rt = ctx.route(codes, limit=150)
            ctx.set_route_mode(2)
            rt2 = ctx.route(codes, limit=150, counters=False)
        assert np.array_equal(rt["count"], ref["sel_count"]) and np.array_equal(rt2["count"], ref["sel_count"])
        sel = np.where(np.arange(150)[None] < rt["count"][:, None], rt["ids"][:, :150], -1)
        assert np.array_equal(sel, ref["sel"][:, :150])
        assert np.array_equal(rt2["ids"], rt["ids"])
        assert (rt["score"][np.arange(32), 0] <= tau).any()        # the threshold actually splits some lists


def test_bounded_select_size_classes(pkg, oracle):
    """limit <= 256 -> 512-entry class, <= 512 -> 1024, <= 1024 -> 2048 (BASELINE config #4's B): each against the oracle's list, with
    levels wide enough that the class's entry budget is really used (16 tables x 5 probes x 64 ids per query)."""
    sc = make_scene(oracle, n=60000, d=24, T=16, D=1, m=14, lam=2, B=1024, seed=77)
    before = len(LAZY_RUNS)
    check_route(pkg, sc, nq=24, limits=(200, 256, 257, 512, 513, 700, 1000, 1024))
    runs = LAZY_RUNS[before:]
    assert len(runs) == 8 and all(r["lazy"] for r in runs), runs
    sc = make_scene(oracle, n=30000, d=16, T=12, D=2, m=12, lam=2, B=900, seed=78, clustered=True)
    sc["params"]["clustered"] = True
    check_route(pkg, sc, nq=16, probes=10, limits=(900, 1024))
