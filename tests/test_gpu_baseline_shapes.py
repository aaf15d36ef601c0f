"""GPU parity at the dimensions / table counts of BASELINE.json configs 3 and 4 (reduced N so the oracle
finishes in seconds): d = 960 and d = 768, 16 x 32-bit and 32 x 64-bit tables, B = 512 / 1024."""
import numpy as np
import pytest

from conftest import make_scene

pytestmark = pytest.mark.gpu

SHAPES = [
    dict(n=20000, d=960, T=16, D=1, m=16, lam=2, B=512, nq=24),    # config #3 shape (GIST-like)
    dict(n=20000, d=768, T=32, D=1, m=32, lam=2, B=1024, nq=16),   # config #4 shape: 64-bit codes, key = first 63 bits
]


@pytest.mark.parametrize("shape", SHAPES, ids=["cfg3_d960", "cfg4_d768_b64"])
def test_full_path_at_baseline_shapes(pkg, oracle, shape):
    nq = shape.pop("nq")
    try:
        sc = make_scene(oracle, seed=7, **shape)
    finally:
        shape["nq"] = nq
    p, o = sc["params"], sc["oracle"]
    K = 10
    Q = sc["rng"].standard_normal((nq, p["d"])).astype(np.float32)
    ref = o.search(Q.astype(np.float64), K)
    assert not o.unmodelled and not ref["metrics"][:, 4].any()
    cfg = pkg.PaperRuntimeConfig(tables=p["T"], divisions=p["D"], m=p["m"], lambda_=p["lam"], dim=p["d"], refinement_limit=p["B"])
    with pkg.FspannContext(cfg, 0) as ctx:
        ctx.set_gfunctions(sc["alpha"], sc["r"], sc["omega"])
        ctx.set_id_meta(p["n"])
        ctx.build_index(sc["X"])                 # MFMA pre-filter path (n >= 4096) + exact re-check
        for td in (0, p["T"] - 1):
            a, b = ctx.get_index(td), o.get_index(td)
            assert all(np.array_equal(a[k], b[k]) for k in a)
        codes = ctx.encode(Q)
        assert np.array_equal(codes, o.encode(Q.astype(np.float64)))
        B = p["B"]
        rt = ctx.route(codes, limit=B)
        assert np.array_equal(rt["count"], ref["sel_count"])
        cand = np.zeros((nq, B, p["d"]), np.float32)
        for i in range(nq):
            c = rt["count"][i]
            assert np.array_equal(rt["ids"][i, :c], ref["sel"][i, :c])
            cand[i, :c] = sc["X"][rt["ids"][i, :c]]
        res = ctx.refine(Q, cand, rt["ids"][:, :B], rt["count"], K)     # B > 256: chunked scan + merge kernel
    assert np.array_equal(res["ids"], ref["ids"])
    assert np.array_equal(res["dist"], ref["dist"])
    assert np.array_equal(rt["kept"], ref["metrics"][:, 1]) and np.array_equal(rt["raw_seen"], ref["metrics"][:, 0])
