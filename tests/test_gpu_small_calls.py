"""GPU parity of the operator surface's OWN call pattern: QueryService.search is one token per call (ForwardSecureANNSystem.java:636-748),
so fspann_encode / fspann_route / fspann_refine see nq = 1.  Calls of a handful of queries take the zero-copy path (the kernels read
their arguments from, and write their results into, the context's mapped pinned block), larger ones one copy each way, and
FSPANN_ZERO_COPY=0 sends everything through copies: all three must give what one batched call gives — which the other suites pin to
the oracle.  Candidate rows may come from the context's pinned block (fspann_host_buffer) or from ordinary memory."""
import numpy as np
import pytest

from conftest import make_scene

pytestmark = pytest.mark.gpu


def _ctx(pkg, sc):
    p = sc["params"]
    cfg = pkg.PaperRuntimeConfig(tables=p["T"], divisions=p["D"], m=p["m"], lambda_=p["lam"], dim=p["d"], refinement_limit=p["B"],
                                 max_global_candidates=p["hard_cap"], probe_override=p["probe_override"])
    ctx = pkg.FspannContext(cfg, 0)
    ctx.set_gfunctions(sc["alpha"], sc["r"], sc["omega"])
    ctx.set_id_meta(p["n"])
    ctx.build_index(sc["X"])
    return ctx


@pytest.mark.parametrize("zero_copy", ["1", "0"])
def test_one_token_per_call_equals_the_batched_call(pkg, oracle, monkeypatch, zero_copy):
    monkeypatch.setenv("FSPANN_ZERO_COPY", zero_copy)
    sc = make_scene(oracle, n=30000, d=24, T=4, D=4, m=12, lam=2, B=192, seed=71)
    p, o = sc["params"], sc["oracle"]
    nq, B, K = 40, p["B"], 10
    Q = sc["rng"].standard_normal((nq, p["d"])).astype(np.float32).astype(np.float64)
    ref = o.search(Q, K)
    assert not o.unmodelled
    with _ctx(pkg, sc) as ctx:
        codes_all = ctx.encode(Q)
        assert np.array_equal(codes_all, o.encode(Q))
        rt_all = ctx.route(codes_all, limit=B, counters=False)
        i = 0
        for step in (1, 1, 2, 3, 4, 5, 8, 16):            # handfuls (zero copy), then calls beyond it (one copy each way)
            sl = slice(i, i + step)
            i += step
            codes = ctx.encode(Q[sl])
            assert np.array_equal(codes, codes_all[sl]), step
            full = ctx.route(codes)                         # whole lists with the counters
            ids_o, score_o, count_o, raw_o = o.route(codes)
            assert np.array_equal(full["count"], count_o) and np.array_equal(full["raw_seen"], raw_o)
            rt = ctx.route(codes, limit=B, counters=False)
            assert np.array_equal(rt["count"], rt_all["count"][sl])
            for j in range(step):
                c = rt["count"][j]
                assert np.array_equal(rt["ids"][j, :c], rt_all["ids"][sl][j, :c]) and np.array_equal(rt["ids"][j, :c], ids_o[j, :c])
            rows = np.zeros((step, B, p["d"]), np.float64)
            for j in range(step):
                rows[j, :rt["count"][j]] = sc["X64"][rt["ids"][j, :rt["count"][j]]]
            out = ctx.refine(Q[sl], rows, rt["ids"][:, :B], rt["count"], K)
            assert np.array_equal(out["ids"], ref["ids"][sl]) and np.array_equal(out["dist"], ref["dist"][sl]), step
            assert np.array_equal(out["count"], ref["count"][sl])
            # the same rows packed into the context's pinned block
            hb = ctx.host_buffer((step, B, p["d"]), np.float64)
            hb[...] = rows
            out2 = ctx.refine(Q[sl], hb, rt["ids"][:, :B], rt["count"], K)
            assert np.array_equal(out2["ids"], out["ids"]) and np.array_equal(out2["dist"], out["dist"])
        assert i == nq - 0 or i <= nq


def test_host_buffer_grows_and_survives(pkg, oracle):
    sc = make_scene(oracle, n=2000, d=8, T=2, D=2, m=6, lam=2, B=64, seed=72)
    with _ctx(pkg, sc) as ctx:
        a = ctx.host_buffer((4, 8), np.float64)
        a[...] = 1.5
        b = ctx.host_buffer((2, 8), np.float64)             # smaller: the same block
        assert b[0, 0] == 1.5
        c = ctx.host_buffer((1 << 16, 64), np.float32)      # 16 MB: a new block
        c[...] = 2.0
        assert float(c.sum()) == 2.0 * c.size
        d = ctx.host_buffer((3, 5), np.int32)
        d[...] = 7
        assert int(d.sum()) == 105


def test_nan_query_is_refused_in_a_single_call(pkg, oracle):
    sc = make_scene(oracle, n=2000, d=8, T=2, D=2, m=6, lam=2, B=64, seed=73)
    with _ctx(pkg, sc) as ctx:
        q = np.zeros((1, 8), np.float64)
        q[0, 3] = np.nan
        with pytest.raises(ValueError):
            ctx.encode(q)                                   # Coding.java:360 -> IllegalArgumentException
        assert ctx.encode(np.ones((1, 8), np.float64)).shape == (1, 4, 1)


def test_single_call_whose_bounded_select_hands_the_query_over(pkg, oracle, monkeypatch):
    # FSPANN_ROUTE_LAZY_CAP = 64: no query fits the bounded select's size class, every one is handed to the full select — in a
    # zero-copy call that launch is issued only after the host has seen the PENDING count
    monkeypatch.setenv("FSPANN_ROUTE_LAZY_CAP", "64")
    sc = make_scene(oracle, n=30000, d=24, T=4, D=4, m=12, lam=2, B=192, seed=74)
    p, o = sc["params"], sc["oracle"]
    Q = sc["rng"].standard_normal((9, p["d"])).astype(np.float32).astype(np.float64)
    codes = o.encode(Q)
    ids_o, score_o, count_o, raw_o = o.route(codes)
    with _ctx(pkg, sc) as ctx:
        i = 0
        for step in (1, 2, 1, 5):
            sl = slice(i, i + step)
            i += step
            rt = ctx.route(codes[sl], limit=p["B"], counters=False)
            info = ctx.last_route_info()
            assert info["lazy"] and info["overflowed"] == step, info
            for j in range(step):
                c = min(p["B"], count_o[sl][j])
                assert rt["count"][j] == c and np.array_equal(rt["ids"][j, :c], ids_o[sl][j, :c])
