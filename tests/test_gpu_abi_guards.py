"""The C ABI validates what it is handed before a kernel can index with it, and never lets a C++ exception out
(include/fspann.h conventions).  Error classes follow the reference's exceptions (IllegalArgument / IllegalState)."""
import os
import struct

import numpy as np
import pytest

from conftest import make_scene

pytestmark = pytest.mark.gpu


def _scene(oracle):
    return make_scene(oracle, n=3000, d=8, T=2, D=2, m=8, lam=2, B=64, seed=5)


def _ctx(pkg, sc):
    p = sc["params"]
    cfg = pkg.PaperRuntimeConfig(tables=p["T"], divisions=p["D"], m=p["m"], lambda_=p["lam"], dim=p["d"], refinement_limit=p["B"])
    ctx = pkg.FspannContext(cfg, 0)
    ctx.set_gfunctions(sc["alpha"], sc["r"], sc["omega"])
    return ctx


def test_tables_before_id_meta_with_a_bad_handle(pkg, oracle):
    """The adapter's documented order sets the tables first (GpuRouteRefine.exportTable ... finish()): handles are
    validated by finalize, whatever the order, and Route stays off until a finalize has succeeded."""
    sc = _scene(oracle)
    o, n = sc["oracle"], sc["params"]["n"]
    codes = o.encode(sc["rng"].standard_normal((4, 8)))
    with _ctx(pkg, sc) as ctx:
        for td in range(o.TD):
            t = o.get_index(td)
            if td == 1:
                t["ids"] = t["ids"].copy()
                t["ids"][17] = n + 5                        # a handle the id metadata will not cover
            ctx.set_index(td, **t)
        ctx.set_id_meta(n)
        with pytest.raises(pkg.FspannArgumentError, match="out of range"):
            ctx.finalize()
        with pytest.raises(pkg.FspannStateError, match="not finalized"):
            ctx.route(codes)
        ctx.set_index(1, **o.get_index(1))                  # repaired
        ctx.finalize()
        assert ctx.route(codes)["count"].min() > 0
        # shrinking the id universe afterwards un-freezes the context; the next finalize re-validates every table
        ctx.set_id_meta(n - 100)
        with pytest.raises(pkg.FspannStateError, match="not finalized"):
            ctx.route(codes)
        with pytest.raises(pkg.FspannArgumentError, match="out of range"):
            ctx.finalize()
        # an id twice in one table: a division's HashMap cannot hold that
        ctx.set_id_meta(n)
        t = o.get_index(0)
        t["ids"] = t["ids"].copy()
        t["ids"][3] = t["ids"][900]
        ctx.set_index(0, **t)
        with pytest.raises(pkg.FspannArgumentError, match="twice"):
            ctx.finalize()


def test_build_index_order_is_a_permutation_of_the_rows(pkg, oracle):
    sc = _scene(oracle)
    n = sc["params"]["n"]
    with _ctx(pkg, sc) as ctx:
        ctx.set_id_meta(n + 50)                             # more handles than rows is allowed ...
        order = np.arange(n, dtype=np.int32)
        order[10] = n + 3                                   # ... but a staged handle without a row is not
        with pytest.raises(pkg.FspannArgumentError, match="not a handle"):
            ctx.build_index(sc["X"], order=order)
        order = np.arange(n, dtype=np.int32)
        order[10] = order[11]
        with pytest.raises(pkg.FspannArgumentError, match="twice"):
            ctx.build_index(sc["X"], order=order)
        ctx.build_index(sc["X"], order=np.arange(n, dtype=np.int32)[::-1].copy())


def test_corrupt_index_files_fail_cleanly(pkg, oracle, tmp_path):
    sc = _scene(oracle)
    n = sc["params"]["n"]
    good = str(tmp_path / "good.fsx")
    with _ctx(pkg, sc) as ctx:
        ctx.set_id_meta(n)
        ctx.build_index(sc["X"])
        ctx.save_index(good)
        codes = sc["oracle"].encode(sc["rng"].standard_normal((4, 8)))
        want = ctx.route(codes)
    blob = open(good, "rb").read()
    hdr = 8 + 4 + 24                                        # magic, version, cfg
    cases = {
        "empty": b"",
        "magic_only": blob[:8],
        "cut_in_header": blob[:hdr + 3],
        "cut_in_gfunctions": blob[:hdr + 9 + 1000],
        "cut_in_table": blob[:len(blob) - 1234],
        "huge_n_ids": blob[:hdr] + struct.pack("<q", (1 << 31) - 1) + blob[hdr + 8:],
        "negative_n_ids": blob[:hdr] + struct.pack("<q", -4) + blob[hdr + 8:],
    }
    # a table header claiming 2^40 partitions / ids: must be refused before anything is sized from it
    P, d = sc["params"]["T"] * sc["params"]["D"] * sc["params"]["m"], sc["params"]["d"]
    t0 = hdr + 9 + (P * d + 2 * P) * 8 + n * 5
    cases["huge_n_parts"] = blob[:t0] + struct.pack("<q", 1 << 40) + blob[t0 + 8:]
    cases["huge_table_ids"] = blob[:t0 + 8] + struct.pack("<q", 1 << 40) + blob[t0 + 16:]
    cases["negative_n_parts"] = blob[:t0] + struct.pack("<q", -1) + blob[t0 + 8:]
    for name, data in cases.items():
        path = str(tmp_path / (name + ".fsx"))
        open(path, "wb").write(data)
        with _ctx(pkg, sc) as ctx:
            ctx.set_id_meta(n)
            ctx.build_index(sc["X"])                        # a frozen context ...
            with pytest.raises((pkg.FspannArgumentError, pkg.FspannStateError)):
                ctx.load_index(path)
            with pytest.raises(pkg.FspannStateError, match="not finalized"):   # ... is no longer frozen after a failed load
                ctx.route(codes)
            ctx.load_index(good)                            # and recovers with a good file
            got = ctx.route(codes)
            assert np.array_equal(got["ids"], want["ids"]) and np.array_equal(got["count"], want["count"])
    with _ctx(pkg, sc) as ctx:
        with pytest.raises(pkg.FspannArgumentError):
            ctx.load_index(os.path.join(str(tmp_path), "does_not_exist.fsx"))


def test_clone_shares_the_index_and_keeps_it_read_only(pkg, oracle):
    """fspann_ctx_clone: same results from the clone, shared state read-only on both sides while the clone lives, the owner may go first."""
    from conftest import make_scene
    sc = make_scene(oracle, n=9000, d=16, T=5, D=1, m=10, lam=2, B=128, seed=31)     # B >= 10 K: QSI's adaptive retry stays off
    p = sc["params"]
    cfg = pkg.PaperRuntimeConfig(tables=p["T"], divisions=p["D"], m=p["m"], lambda_=p["lam"], dim=p["d"], refinement_limit=p["B"],
                                 max_global_candidates=p["hard_cap"])
    Q = sc["rng"].standard_normal((40, p["d"])).astype(np.float32)
    ctx = pkg.FspannContext(cfg, 0)
    ctx.set_gfunctions(sc["alpha"], sc["r"], sc["omega"])
    ctx.set_id_meta(p["n"])
    with pytest.raises(pkg.FspannStateError, match="not finalized"):
        ctx.clone()
    ctx.build_index(sc["X"])
    ctx.store_set(sc["X"])
    ref = sc["oracle"].search(Q.astype(np.float64), 10)
    cl = ctx.clone()
    cl2 = cl.clone()                                           # a clone of a clone shares the same owner
    for c_ in (ctx, cl, cl2):
        codes = c_.encode(Q)
        rt = c_.route(codes, limit=p["B"], counters=False)
        rs = c_.refine_store(Q, rt["ids"][:, :p["B"]], rt["count"], 10)
        assert np.array_equal(rt["count"], ref["sel_count"])
        assert np.array_equal(rs["ids"], ref["ids"]) and np.array_equal(rs["dist"], ref["dist"])
    assert all(np.array_equal(cl.get_index(0)[k_], v_) for k_, v_ in ctx.get_index(0).items())
    for c_ in (ctx, cl):                                       # read-only on both sides
        with pytest.raises(pkg.FspannStateError, match="clone"):
            c_.set_id_meta(p["n"])
        with pytest.raises(pkg.FspannStateError, match="clone"):
            c_.build_index(sc["X"])
    ctx.close()                                                # the owner goes first: its arrays live on for the clones
    codes = cl.encode(Q)
    rt = cl.route(codes, limit=p["B"], counters=False)
    rs = cl2.refine_store(Q, rt["ids"][:, :p["B"]], rt["count"], 10)
    assert np.array_equal(rs["ids"], ref["ids"]) and np.array_equal(rs["dist"], ref["dist"])
    cl.close()
    cl2.close()
