"""Boundary completeness for a real JVM (through the C ABI, ctypes):

  * fspann_build_begin / _append / _finish: IndexService.insert is one vector at a time (common/.../IndexService.java:19,
    PIS:265-347) and a direct ByteBuffer holds at most 2 GB — Setup accepts the rows in pieces (fp32 or fp64, any chunking) and
    produces the index fspann_build_index produces (== the oracle's own);
  * fspann_set_deleted: PIS:739 asks metadata.isDeleted(id) at query time — a delete shows in the next query, without un-freezing
    the context, on the owner and on its clones alike;
  * a context shared by several threads: calls are serialised inside the library, results stay correct.
"""
import threading

import numpy as np
import pytest

from conftest import make_scene

pytestmark = pytest.mark.gpu
K = 5          # 10 K <= B everywhere below: the oracle's adaptive retry (QSI:327-337) stays out of the comparison


def _ctx(pkg, sc, deleted=None):
    p = sc["params"]
    cfg = pkg.PaperRuntimeConfig(tables=p["T"], divisions=p["D"], m=p["m"], lambda_=p["lam"], dim=p["d"], refinement_limit=p["B"],
                                 max_global_candidates=p["hard_cap"])
    ctx = pkg.FspannContext(cfg, 0)
    ctx.set_gfunctions(sc["alpha"], sc["r"], sc["omega"])
    ctx.set_id_meta(p["n"], None, deleted)
    return ctx


def _tables_equal(ctx, o):
    for td in range(o.TD):
        a, b = ctx.get_index(td), o.get_index(td)
        for key in ("min_key", "max_key", "rep", "id_off", "ids"):
            assert np.array_equal(a[key], b[key]), (td, key)


def test_incremental_build_equals_one_shot_build(pkg, oracle):
    sc = make_scene(oracle, n=30000, d=24, T=5, D=2, m=11, lam=2, B=96, seed=61)
    o, X = sc["oracle"], sc["X"]
    n = len(X)
    with _ctx(pkg, sc) as ctx:
        with pytest.raises(pkg.FspannStateError, match="no build in progress"):
            ctx.build_append(X[:10])
        ctx.build_begin(n // 3)                                          # a hint: more rows than hinted are accepted (the buffer grows)
        with pytest.raises(pkg.FspannStateError, match="no rows appended"):
            ctx.build_finish()
        ctx.build_begin(n // 3)                                          # starting over is allowed
        cuts = [0, 1, 7, 4096, 4097, 12000, 12001, 29999, n]             # ragged pieces, fp32 and fp64 alternating, one beyond 4096 rows (MFMA path)
        for i, (a, b) in enumerate(zip(cuts[:-1], cuts[1:])):
            ctx.build_append(X[a:b] if i % 2 == 0 else X[a:b].astype(np.float64))
        ctx.build_finish()
        _tables_equal(ctx, o)
        codes = ctx.encode(sc["rng"].standard_normal((40, 24)).astype(np.float32))
        assert ctx.route(codes, limit=96)["count"].min() > 0            # frozen and serving
        with pytest.raises(pkg.FspannStateError, match="no build in progress"):
            ctx.build_finish()
    with _ctx(pkg, sc) as ctx:                                            # a NaN row fails the append and abandons the build
        ctx.build_begin(n)
        bad = X[:100].copy()
        bad[37, 3] = np.nan
        with pytest.raises(pkg.FspannArgumentError, match="NaN/Inf .handle 37"):
            ctx.build_append(bad)
        with pytest.raises(pkg.FspannStateError, match="no build in progress"):
            ctx.build_append(X[:10])
    order = np.random.default_rng(3).permutation(n).astype(np.int32)      # a caller-given staging order
    o.build_index(sc["X64"], order=order)
    with _ctx(pkg, sc) as ctx:
        ctx.build_begin(n)
        for a in range(0, n, 7001):
            ctx.build_append(X[a:a + 7001])
        ctx.build_finish(order)
        _tables_equal(ctx, o)


def _check_against(o, ctx, codes, B, note):
    ids, score, count, raw = o.route(codes)
    for cx in ctx if isinstance(ctx, (list, tuple)) else [ctx]:
        for counters in (True, False):                                     # full select / bounded select
            r = cx.route(codes, limit=B, counters=counters)
            c = np.minimum(count, B)
            assert np.array_equal(r["count"], c), note
            for i in range(len(codes)):
                assert np.array_equal(r["ids"][i, :c[i]], ids[i, :c[i]]), (note, i, counters)
            if counters:
                assert np.array_equal(r["kept"], count) and np.array_equal(r["raw_seen"], raw), note


def test_live_deletion_shows_in_the_next_query_on_owner_and_clones(pkg, oracle):
    sc = make_scene(oracle, n=40000, d=16, T=8, D=1, m=12, lam=2, B=128, seed=71)
    o, p = sc["oracle"], sc["params"]
    n, B = p["n"], p["B"]
    rng = np.random.default_rng(9)
    codes = o.encode(rng.standard_normal((48, 16)))
    with _ctx(pkg, sc) as ctx:
        ctx.build_index(sc["X"])
        ctx.set_route_mode(2)
        clone = ctx.clone()
        clone.set_route_mode(2)
        try:
            _check_against(o, [ctx, clone], codes, B, "nothing deleted")
            # delete ids that ARE in the answers (so the lists must change), through the CLONE, while both stay frozen
            ids0, _, cnt0, _ = o.route(codes)
            victims = np.unique(np.concatenate([ids0[i, :min(cnt0[i], 20)] for i in range(0, 48, 3)])).astype(np.int32)
            deleted = np.zeros(n, np.uint8)
            deleted[victims] = 1
            clone.set_deleted(victims, True)
            o.set_id_meta(n, None, deleted)
            _check_against(o, [ctx, clone], codes, B, "after the first delete")
            assert not np.isin(ctx.route(codes, limit=B)["ids"], victims).any()
            # more deletes through the OWNER (a second clone made in between sees all of them), then undelete a few
            more = rng.choice(n, 3000, replace=False).astype(np.int32)
            clone2 = ctx.clone()
            clone2.set_route_mode(2)
            try:
                ctx.set_deleted(more, True)
                deleted[more] = 1
                back = victims[::2]
                ctx.set_deleted(back, False)
                deleted[back] = 0
                o.set_id_meta(n, None, deleted)
                _check_against(o, [ctx, clone, clone2], codes, B, "after more deletes and some undeletes")
            finally:
                clone2.close()
            with pytest.raises(pkg.FspannArgumentError, match="out of range"):
                ctx.set_deleted(np.array([n], np.int32))
            with pytest.raises(pkg.FspannStateError, match="shared with"):      # the index itself stays read-only while clones live
                ctx.set_id_meta(n, None, deleted)
        finally:
            clone.close()
        # the index file carries the live flags
        import os, tempfile
        path = os.path.join(tempfile.mkdtemp(), "ix.bin")
        ctx.save_index(path)
    with _ctx(pkg, sc) as c2:
        c2.load_index(path)
        _check_against(o, c2, codes, B, "reloaded")


def test_undelete_before_any_delete_is_a_no_op(pkg, oracle):
    sc = make_scene(oracle, n=5000, d=8, T=3, D=1, m=8, lam=2, B=64, seed=72)
    o = sc["oracle"]
    codes = o.encode(sc["rng"].standard_normal((8, 8)))
    with _ctx(pkg, sc) as ctx:
        ctx.build_index(sc["X"])
        ctx.set_deleted(np.arange(100, dtype=np.int32), False)
        _check_against(o, ctx, codes, 64, "undelete of nothing")


def test_one_context_shared_by_threads_is_serialised_inside_the_library(pkg, oracle):
    """Eight threads drive ONE context (host-pointer entry points: they use the context's staging areas) — without the
    per-context lock their copies and launches interleave and the results are garbage."""
    sc = make_scene(oracle, n=20000, d=16, T=6, D=1, m=12, lam=2, B=64, seed=73)
    o = sc["oracle"]
    rng = np.random.default_rng(4)
    Qs = [rng.standard_normal((int(rng.integers(5, 60)), 16)).astype(np.float32) for _ in range(8)]
    refs = [o.search(q.astype(np.float64), K) for q in Qs]
    errors = []
    with _ctx(pkg, sc) as ctx:
        ctx.build_index(sc["X"])
        ctx.store_set(sc["X"])

        def work(i):
            try:
                for _ in range(12):
                    codes = ctx.encode(Qs[i])
                    rt = ctx.route(codes, limit=64, counters=False)
                    out = ctx.refine_store(Qs[i], rt["ids"][:, :64], rt["count"], K)
                    if not (np.array_equal(out["ids"], refs[i]["ids"]) and np.array_equal(out["dist"], refs[i]["dist"])):
                        errors.append(i)
            except Exception as e:   # noqa: BLE001
                errors.append((i, repr(e)))
        th = [threading.Thread(target=work, args=(i,)) for i in range(8)]
        for t in th:
            t.start()
        for t in th:
            t.join()
    assert not errors, errors[:4]


def test_clones_created_and_destroyed_from_many_threads(pkg, oracle):
    sc = make_scene(oracle, n=4000, d=8, T=3, D=1, m=8, lam=2, B=64, seed=74)
    o = sc["oracle"]
    codes = o.encode(sc["rng"].standard_normal((6, 8)))
    ids, _, count, _ = o.route(codes)
    errors = []
    ctx = _ctx(pkg, sc)
    ctx.build_index(sc["X"])

    def work():
        try:
            for _ in range(10):
                c = ctx.clone()
                r = c.route(codes, limit=64, counters=False)
                if not np.array_equal(r["count"], np.minimum(count, 64)):
                    errors.append("count")
                c.close()
        except Exception as e:   # noqa: BLE001
            errors.append(repr(e))
    th = [threading.Thread(target=work) for _ in range(6)]
    for t in th:
        t.start()
    ctx_closed_early = threading.Event()
    for t in th:
        t.join()
    ctx.close()                      # owner last here; owner-first is covered by test_gpu_abi_guards
    assert not errors and not ctx_closed_early.is_set(), errors[:4]
