"""GPU: the MFMA fp32 GEMM encode path (pre-filter + exact fp64 re-check) is bit-identical to the exact
fp64 kernel and to the oracle, including inputs constructed to sit on bucket boundaries."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _setup(pkg, oracle, T, D, m, lam, d, seed, scale=1.0):
    rng = np.random.default_rng(seed)
    S = (rng.standard_normal((1000, d)) * scale).astype(np.float32).astype(np.float64)
    alpha, r, w = oracle.registry_init(S, m, 13, T, D)
    o = oracle.Oracle(T, D, m, lam, d)
    o.set_gfunctions(alpha, r, w)
    ctx = pkg.FspannContext(pkg.PaperRuntimeConfig(tables=T, divisions=D, m=m, lambda_=lam, dim=d), 0)
    ctx.set_gfunctions(alpha, r, w)
    return rng, o, ctx, (alpha, r, w)


@pytest.mark.parametrize("T,D,m,lam,d,nq", [(16, 1, 16, 2, 128, 1024), (3, 2, 24, 2, 100, 333), (2, 2, 7, 3, 33, 70),
                                            (8, 1, 8, 2, 960, 200), (1, 1, 1, 1, 1, 5), (2, 1, 100, 2, 64, 65)])
def test_mfma_equals_exact(pkg, oracle, T, D, m, lam, d, nq):
    rng, o, ctx, _ = _setup(pkg, oracle, T, D, m, lam, d, seed=T * 7 + d)
    with ctx:
        for dt in (np.float32, np.float64):
            Q = (rng.standard_normal((nq, d)) * 2).astype(dt)
            ref_h, ref_c = o.hashes(Q.astype(np.float64)), o.encode(Q.astype(np.float64))
            ctx.set_encode_mode(1)
            c1, h1 = ctx.encode(Q, want_hashes=True)
            assert ctx.last_encode_rechecked() == 0
            ctx.set_encode_mode(2)
            c2, h2 = ctx.encode(Q, want_hashes=True)
            c3 = ctx.encode(Q)                                   # without the hashes output buffer
            n_fix = ctx.last_encode_rechecked()
            assert np.array_equal(h1, ref_h) and np.array_equal(c1, ref_c)
            assert np.array_equal(h2, ref_h) and np.array_equal(c2, ref_c) and np.array_equal(c3, ref_c)
            assert n_fix < 0.01 * h2.size + 8, n_fix            # the guard band is narrow on ordinary data


def test_mfma_boundary_cases_are_rechecked(pkg, oracle):
    """Vectors scaled so that (alpha.v + r)/omega lands within 1e-12..1e-6 of an integer."""
    T, D, m, lam, d = 4, 1, 16, 2, 64
    rng, o, ctx, (alpha, r, w) = _setup(pkg, oracle, T, D, m, lam, d, seed=99)
    with ctx:
        rows = []
        for i in range(400):
            v = rng.standard_normal(d)
            p = rng.integers(0, T * D * m)
            a = alpha.reshape(-1, d)[p]
            y = float(np.dot(v, a))
            kbucket = rng.integers(-3, 4)
            eps = rng.choice([0.0, 1e-12, -1e-12, 1e-9, -1e-9, 1e-7, -1e-7, 3e-6, -3e-6])
            target = (kbucket + eps) * w.reshape(-1)[p] - r.reshape(-1)[p]
            rows.append(v * (target / y) if abs(y) > 1e-3 else v)
        Q = np.array(rows)
        ref_h, ref_c = o.hashes(Q), o.encode(Q)
        ctx.set_encode_mode(2)
        c2, h2 = ctx.encode(Q, want_hashes=True)
        assert ctx.last_encode_rechecked() >= 300      # the constructed pairs (and a few natural ones) hit the band
        assert np.array_equal(h2, ref_h) and np.array_equal(c2, ref_c)
        c32 = ctx.encode(Q.astype(np.float32))         # fp32 inputs: reference = widened fp32
        assert np.array_equal(c32, o.encode(Q.astype(np.float32).astype(np.float64)))


def test_mfma_degenerate_omega_overflows_to_exact(pkg, oracle):
    """omega so small that nearly every pair is inside the guard band: the re-check list overflows and the
    guarded exact kernel recomputes the batch."""
    T, D, m, lam, d = 2, 1, 16, 2, 32
    rng, o, ctx, (alpha, r, w) = _setup(pkg, oracle, T, D, m, lam, d, seed=5)
    w2 = np.full_like(w, 1e-6)
    r2 = r * 0 + 3e-7
    o.set_gfunctions(alpha, r2, w2)
    with ctx:
        ctx.set_gfunctions(alpha, r2, w2)
        Q = (rng.standard_normal((40000, d)) * 50).astype(np.float32)
        ctx.set_encode_mode(2)
        c2, h2 = ctx.encode(Q, want_hashes=True)
        assert ctx.last_encode_rechecked() > 40000 * T * D * m // 16
        assert np.array_equal(h2, o.hashes(Q.astype(np.float64)))
        assert np.array_equal(c2, o.encode(Q.astype(np.float64)))


def test_mfma_nonfinite_and_huge(pkg, oracle):
    T, D, m, lam, d = 2, 1, 8, 2, 16
    rng, o, ctx, _ = _setup(pkg, oracle, T, D, m, lam, d, seed=6)
    with ctx:
        ctx.set_encode_mode(2)
        Q = rng.standard_normal((9, d))
        Q[2] *= 1e200          # overflows fp32 -> re-checked exactly, int32 saturation
        Q[3] *= 1e-200         # underflows fp32
        Q[4] = 0.0
        c2, h2 = ctx.encode(Q, want_hashes=True)
        assert np.array_equal(h2, o.hashes(Q)) and np.array_equal(c2, o.encode(Q))
        Q[5, 3] = np.nan
        with pytest.raises(pkg.FspannArgumentError, match="NaN/Inf"):
            ctx.encode(Q)


def test_auto_mode_index_build_uses_mfma_and_matches(pkg, oracle):
    T, D, m, lam, d, n = 4, 1, 16, 2, 256, 200000      # 200 000 x 64 x 256 = 3.3e9 multiply-adds: above the auto rule's 5e8
    rng, o, ctx, _ = _setup(pkg, oracle, T, D, m, lam, d, seed=8)
    X = rng.standard_normal((n, d)).astype(np.float32)
    o.set_id_meta(n)
    o.build_index(X.astype(np.float64))
    with ctx:
        ctx.set_id_meta(n)
        ctx.build_index(X)                      # auto mode: rows x projections x dim >= 5e8 -> MFMA path
        assert ctx.last_encode_rechecked() > 0   # (only the MFMA path re-checks pairs)
        for td in range(T * D):
            a, b = ctx.get_index(td), o.get_index(td)
            for k in a:
                assert np.array_equal(a[k], b[k]), (td, k)
