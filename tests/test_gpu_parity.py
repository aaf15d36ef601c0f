"""GPU parity: every stage of the HIP path against the CPU oracle, through the C ABI.

Bar (BASELINE.json north_star): bit-exact routed candidate id sets; L2 distances
within 1e-5 relative — here they are asserted BIT-EXACT because inputs are
fp32-representable and the kernel sums in the reference's order in fp64.
"""
import numpy as np
import pytest

from conftest import make_scene

pytestmark = pytest.mark.gpu

SCENES = [
    # n, d, T, D, m, lam, B
    dict(n=3000, d=16, T=3, D=2, m=8, lam=2, B=64),        # small
    dict(n=10000, d=128, T=8, D=1, m=8, lam=2, B=64),      # BASELINE config #1
    dict(n=5000, d=24, T=2, D=4, m=24, lam=3, B=200),      # b = 72 bits -> W = 2 words
    dict(n=1024, d=8, T=2, D=4, m=4, lam=3, B=128, clustered=True),   # BaseUnifiedIT-shaped: heavy key ties
    dict(n=2500, d=6, T=3, D=4, m=6, lam=3, B=40),         # MultiTableSystemIntegrationTest-shaped params
    dict(n=4000, d=33, T=2, D=2, m=10, lam=2, B=300),      # odd dim (scalar load path), B > 256 -> chunk merge
]


def ctx_for(pkg, sc, **over):
    p = dict(sc["params"])
    p.update(over)
    cfg = pkg.PaperRuntimeConfig(tables=p["T"], divisions=p["D"], m=p["m"], lambda_=p["lam"], dim=p["d"],
                                 refinement_limit=p["B"], max_global_candidates=p["hard_cap"],
                                 probe_override=p["probe_override"], hamming_prefilter_threshold=p["tau"])
    ctx = pkg.FspannContext(cfg, 0)
    ctx.set_gfunctions(sc["alpha"], sc["r"], sc["omega"])
    ctx.set_id_meta(p["n"], None, sc["deleted"])
    return ctx


@pytest.fixture(scope="module", params=range(len(SCENES)))
def scene(request, oracle):
    return make_scene(oracle, seed=100 + request.param, **SCENES[request.param])


def test_encode_bit_exact(pkg, scene):
    Q = scene["rng"].standard_normal((37, scene["params"]["d"])).astype(np.float32)
    with ctx_for(pkg, scene) as ctx:
        codes, hs = ctx.encode(Q, want_hashes=True)
        codes64 = ctx.encode(Q.astype(np.float64))
    o = scene["oracle"]
    assert np.array_equal(hs, o.hashes(Q.astype(np.float64)))
    assert np.array_equal(codes, o.encode(Q.astype(np.float64)))
    assert np.array_equal(codes64, codes)


def test_build_index_matches_oracle(pkg, scene):
    o = scene["oracle"]
    with ctx_for(pkg, scene) as ctx:
        ctx.build_index(scene["X"])
        for td in range(o.TD):
            a, b = ctx.get_index(td), o.get_index(td)
            for k in ("min_key", "max_key", "rep", "id_off", "ids"):
                assert np.array_equal(a[k], b[k]), (td, k)
    assert not o.unmodelled


@pytest.mark.parametrize("probes", [-1, 2, 10])
def test_route_full_list(pkg, scene, probes):
    o = scene["oracle"]
    Q = scene["rng"].standard_normal((24, scene["params"]["d"])).astype(np.float32).astype(np.float64)
    codes = o.encode(Q)
    ids, score, count, raw = o.route(codes, probe_override=probes)
    with ctx_for(pkg, scene) as ctx:
        ctx.build_index(scene["X"])
        res = ctx.route(codes, probe_override=probes)
    assert not o.unmodelled
    assert np.array_equal(res["count"], count)
    assert np.array_equal(res["kept"], count)
    assert np.array_equal(res["raw_seen"], raw)
    for i in range(len(Q)):
        assert np.array_equal(res["ids"][i, :count[i]], ids[i, :count[i]]), i
        assert np.array_equal(res["score"][i, :count[i]], score[i, :count[i]]), i


def test_route_select_first_B(pkg, scene):
    o = scene["oracle"]
    B = scene["params"]["B"]
    Q = scene["rng"].standard_normal((24, scene["params"]["d"])).astype(np.float32).astype(np.float64)
    codes = o.encode(Q)
    ids, score, count, raw = o.route(codes)
    with ctx_for(pkg, scene) as ctx:
        ctx.build_index(scene["X"])
        res = ctx.route(codes, limit=B)
    for i in range(len(Q)):
        n = min(B, count[i])
        assert res["count"][i] == n
        assert res["kept"][i] == count[i]
        assert np.array_equal(res["ids"][i, :n], ids[i, :n]), i


def test_refine_bit_exact(pkg, scene):
    p = scene["params"]
    rng = scene["rng"]
    nq, B, d, K = 9, p["B"], p["d"], 10
    Q = rng.standard_normal((nq, d)).astype(np.float32)
    cand_ids = np.stack([rng.choice(p["n"], B, replace=False) for _ in range(nq)]).astype(np.int32)
    cand = scene["X"][cand_ids]
    cand_count = rng.integers(0, B + 1, nq).astype(np.int32)
    cand_count[0] = B
    cand_count[1] = 0
    import oracle.oracle as O
    ref_ids, ref_dist, ref_cnt = O.refine(Q, cand, cand_ids, cand_count, K)
    with ctx_for(pkg, scene) as ctx:
        res = ctx.refine(Q, cand, cand_ids, cand_count, K)
        res64 = ctx.refine(Q.astype(np.float64), cand.astype(np.float64), cand_ids, cand_count, K)
    for r in (res, res64):
        assert np.array_equal(r["count"], ref_cnt)
        assert np.array_equal(r["ids"], ref_ids)
        assert np.array_equal(r["dist"], ref_dist)   # bit-exact fp64 (inf padding included)
        assert np.array_equal(r["scored"], cand_count)


def test_search_end_to_end(pkg, scene):
    """encode -> route(limit=B) -> host gather -> refine == oracle's QSI.search (no retry: K*10 <= B)."""
    o = scene["oracle"]
    p = scene["params"]
    B = p["B"]
    K = max(1, min(10, B // 10))
    Q = scene["rng"].standard_normal((20, p["d"])).astype(np.float32)
    ref = o.search(Q.astype(np.float64), K)
    assert not ref["metrics"][:, 4].any()
    with ctx_for(pkg, scene) as ctx:
        ctx.build_index(scene["X"])
        codes = ctx.encode(Q)
        rt = ctx.route(codes, limit=B)
        cand = np.zeros((len(Q), B, p["d"]), np.float32)
        for i in range(len(Q)):
            c = rt["count"][i]
            cand[i, :c] = scene["X"][rt["ids"][i, :c]]
        res = ctx.refine(Q, cand, rt["ids"][:, :B], rt["count"], K)
    assert np.array_equal(rt["count"], ref["sel_count"])
    for i in range(len(Q)):
        assert np.array_equal(rt["ids"][i, :rt["count"][i]], ref["sel"][i, :rt["count"][i]])
    assert np.array_equal(res["ids"], ref["ids"])
    assert np.array_equal(res["dist"], ref["dist"])
    assert np.array_equal(rt["raw_seen"], ref["metrics"][:, 0])
    assert np.array_equal(rt["kept"], ref["metrics"][:, 1])
