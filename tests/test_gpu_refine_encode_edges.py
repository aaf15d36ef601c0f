"""GPU parity, Refine and Encode edge cases against the oracle."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _ctx(pkg, d, B=64, T=1, D=1, m=8, lam=2):
    return pkg.FspannContext(pkg.PaperRuntimeConfig(tables=T, divisions=D, m=m, lambda_=lam, dim=d, refinement_limit=B), 0)


def _check_refine(pkg, oracle, q, cand, ids, cnt, k, dtype=np.float32):
    q = q.astype(dtype)
    cand = cand.astype(dtype)
    ref_ids, ref_dist, ref_cnt = oracle.refine(q.astype(np.float64), cand.astype(np.float64), ids, cnt, k)
    with _ctx(pkg, cand.shape[2], B=cand.shape[1]) as ctx:
        res = ctx.refine(q, cand, ids, cnt, k)
    assert np.array_equal(res["count"], ref_cnt)
    assert np.array_equal(res["ids"], ref_ids)
    assert np.array_equal(res["dist"], ref_dist)
    return res


@pytest.mark.parametrize("B,k,d", [(256, 10, 128), (256, 1, 128), (256, 100, 64), (300, 10, 32), (1000, 50, 16),
                                   (6000, 100, 24), (64, 10, 960), (17, 32, 7), (257, 33, 12)])
def test_refine_shapes(pkg, oracle, B, k, d):
    rng = np.random.default_rng(B * 31 + k)
    nq = 5
    q = rng.standard_normal((nq, d))
    cand = rng.standard_normal((nq, B, d))
    ids = rng.integers(0, 10**6, (nq, B)).astype(np.int32)
    cnt = np.array([B, 0, 1, min(B, k - 1) if k > 1 else 1, B // 2], np.int32)
    for dt in (np.float32, np.float64):
        _check_refine(pkg, oracle, q, cand, ids, cnt, k, dt)


def test_refine_ties_are_stable(pkg, oracle):
    """Equal distances keep candidate order (List.sort is stable, QSI:298)."""
    rng = np.random.default_rng(1)
    nq, B, d, k = 3, 512, 16, 20
    base = rng.standard_normal((nq, 8, d))
    cand = base[:, rng.integers(0, 8, B)][np.arange(nq)[:, None], np.arange(B)[None] % 1 + np.zeros((nq, B), int)]
    cand = np.stack([base[i][rng.integers(0, 8, B)] for i in range(nq)])      # only 8 distinct rows per query
    q = rng.standard_normal((nq, d))
    ids = np.tile(np.arange(B, dtype=np.int32), (nq, 1))
    cnt = np.full(nq, B, np.int32)
    res = _check_refine(pkg, oracle, q, cand, ids, cnt, k)
    for i in range(nq):
        dist = res["dist"][i]
        same = np.flatnonzero(np.diff(dist) == 0)
        assert len(same) > 0
        assert all(res["ids"][i][j] < res["ids"][i][j + 1] for j in same)      # ties in ascending position


def test_refine_topk_epilogue_corner_cases(pkg, oracle):
    """The scan's top-K epilogue (per-wave bisected cut, dense survivor list, pair-parallel exact rank): every candidate equal
    (all 256 survive the cut: one lane per survivor), a zero distance (top word 0), distances spread over the whole fp64
    exponent range inside one chunk with a small k (the bisect walks every bit), fewer valid rows than k in three of four
    waves, and k = 32 / 33 on either side of the filter's limit."""
    rng = np.random.default_rng(11)
    B, d = 256, 8
    q = rng.standard_normal((6, d))
    cand = rng.standard_normal((6, B, d))
    cand[0, :, :] = cand[0, 0, :]                                   # all equal
    cand[1, 17, :] = q[1]                                           # one exact hit: distance 0
    mags = 10.0 ** rng.uniform(-150, 150, B)                        # squares span 1e-300 .. 1e300
    cand[2] = q[2] + np.outer(mags, np.eye(d)[0])
    cand[3, :, 0] = np.where(np.arange(B) % 64 < 3, cand[3, :, 0], np.nan)     # three valid rows per wave
    cand[4, 64:, 1] = np.inf                                        # only the first wave holds valid rows
    # 256 DIFFERENT distances inside one 2^-10 bucket of the bisected top word: all 256 survive the cut (T = 256, one lane per
    # survivor, no partial counts to meet) and the order is decided by the low bits alone
    cand[5] = q[5] + np.outer(1.0 + np.arange(B)[::-1] * 1e-7, np.eye(d)[1])
    ids = np.tile(np.arange(B, dtype=np.int32), (6, 1))
    cnt = np.full(6, B, np.int32)
    for k in (1, 5, 10, 32, 33):
        _check_refine(pkg, oracle, q, cand, ids, cnt, k, np.float64)
    cand32 = np.clip(cand, -1e30, 1e30)
    cand32[3, :, 0] = np.where(np.arange(B) % 64 < 3, rng.standard_normal(B), np.nan)
    cand32[4, 64:, 1] = np.inf
    for k in (1, 10, 32):
        _check_refine(pkg, oracle, q, cand32, ids, cnt, k, np.float32)


def test_refine_merge_of_many_partial_lists(pkg, oracle):
    """B > 256: per-chunk top-k lists merged by refine_merge_kernel — with the keys staged in LDS (<= 72 KB: 32 lists of 100)
    and from global memory beyond that (118 lists of 100), ties across chunks included."""
    rng = np.random.default_rng(12)
    for B, k, d in ((8000, 100, 8), (30000, 100, 4), (30000, 7, 4)):
        nq = 3
        q = rng.standard_normal((nq, d))
        base = rng.standard_normal((nq, 40, d))
        cand = np.stack([base[i][rng.integers(0, 40, B)] for i in range(nq)])      # 40 distinct rows: ties in every chunk
        cand[1] = rng.standard_normal((B, d))
        ids = np.tile(np.arange(B, dtype=np.int32), (nq, 1))
        cnt = np.array([B, B - 300, 5], np.int32)
        _check_refine(pkg, oracle, q, cand, ids, cnt, k)


@pytest.mark.parametrize("nq,B,k", [(3, 6000, 100), (1, 22000, 100), (40, 2048, 64), (700, 1024, 33), (5, 3000, 128), (2, 777, 40)])
def test_refine_running_topk_over_runs_of_chunks(pkg, oracle, nq, B, k):
    """Long lists, 32 < k <= 128: a workgroup walks a run of consecutive chunks of one query and keeps the best k in LDS
    (refine_topk_running).  Few queries are cut into several runs (their lists merged by refine_merge_kernel), many queries get
    one run each (no merge kernel).  Partial counts (trailing chunks empty or short), few distinct rows (ties across
    chunks keep candidate order, QSI:298), NaN / Inf rows (skipped, QSI:407-413), counts below k — all vs the oracle."""
    rng = np.random.default_rng(nq * 131 + B + k)
    d = 16
    q = rng.standard_normal((nq, d))
    cand = rng.standard_normal((nq, B, d))
    few = rng.standard_normal((nq, 12, d))
    for i in range(0, nq, 2):                                   # every other query: only 12 distinct rows -> ties everywhere
        cand[i] = few[i][rng.integers(0, 12, B)]
    bad = rng.random((nq, B)) < 0.01
    cand[bad, 3] = np.nan
    cand[rng.random((nq, B)) < 0.005, 0] = np.inf
    ids = rng.integers(0, 10**6, (nq, B)).astype(np.int32)
    cnt = rng.integers(0, B + 1, nq).astype(np.int32)
    cnt[0] = B
    if nq > 1:
        cnt[1] = k - 1                                          # fewer rows than k
    if nq > 2:
        cnt[2] = 257                                            # one full chunk and one row
    for dt in (np.float32, np.float64):
        _check_refine(pkg, oracle, q, cand, ids, cnt, k, dt)


def test_refine_merge_with_partial_counts(pkg, oracle):
    """k <= 32 over several chunks = one list per chunk + refine_merge_kernel; cand_count < B leaves trailing lists empty or short:
    the cut is then vouched for by the lists that hold enough keys (it used to be dropped altogether) — same results."""
    rng = np.random.default_rng(5)
    nq, B, d, k = 12, 2048, 16, 20
    q = rng.standard_normal((nq, d))
    cand = rng.standard_normal((nq, B, d))
    ids = rng.integers(0, 10**6, (nq, B)).astype(np.int32)
    cnt = np.array([B, 257, 300, 512, 513, 1000, 19, 20, 21, 256, 1025, 0], np.int32)
    _check_refine(pkg, oracle, q, cand, ids, cnt, k)
    _check_refine(pkg, oracle, q, cand, ids, cnt, 7)


def test_refine_nonfinite(pkg, oracle):
    rng = np.random.default_rng(2)
    nq, B, d, k = 4, 256, 32, 10
    q = rng.standard_normal((nq, d)).astype(np.float32)
    cand = rng.standard_normal((nq, B, d)).astype(np.float32)
    cand[0, 3, 5] = np.nan          # candidate skipped (QSI:253-260)
    cand[0, 7, 0] = np.inf
    cand[1, :, 2] = -np.inf         # every candidate invalid -> empty result
    q[2, 4] = np.nan                # invalid query -> empty result (QSI:137-140)
    cand[3, 9, :] = 3.0e38          # finite inputs, squared sum stays finite in fp64 -> kept
    ids = np.tile(np.arange(B, dtype=np.int32), (nq, 1))
    cnt = np.full(nq, B, np.int32)
    ref_ids, ref_dist, ref_cnt = oracle.refine(q.astype(np.float64), cand.astype(np.float64), ids, cnt, k)
    ref_cnt[2] = 0                  # the oracle helper does not model the query check; QSI returns empty
    ref_ids[2] = -1
    ref_dist[2] = np.inf
    with _ctx(pkg, d, B) as ctx:
        res = ctx.refine(q, cand, ids, cnt, k)
    assert np.array_equal(res["count"], ref_cnt) and list(res["count"]) == [10, 0, 0, 10]
    assert np.array_equal(res["ids"], ref_ids)
    assert np.array_equal(res["dist"], ref_dist)
    assert list(res["scored"]) == [B - 2, 0, 0, B]


def test_refine_sqrt_is_correctly_rounded(pkg):
    """Math.sqrt is IEEE-exact; check the device sqrt on awkward arguments through d = 1 distances."""
    rng = np.random.default_rng(3)
    B = 256
    vals = np.concatenate([rng.random(100) * 1e-300, rng.random(100) * 1e300, [0.0, 1.0, 2.0, 3.0, 1e-320, 4.9e-324],
                           rng.random(50)]).astype(np.float64)[:B]
    cand = np.sqrt(vals).reshape(1, B, 1)             # distance = |0 - x| = sqrt(x^2) ...
    q = np.zeros((1, 1))
    with _ctx(pkg, 1, B) as ctx:
        res = ctx.refine(q, cand, np.arange(B, dtype=np.int32)[None], np.array([B], np.int32), B)
    got = dict(zip(res["ids"][0], res["dist"][0]))
    for j in range(B):
        x = cand[0, j, 0]
        assert got[j] == np.sqrt(x * x)


@pytest.mark.parametrize("T,D,m,lam,d", [(1, 1, 1, 1, 1), (3, 2, 24, 2, 128), (2, 3, 7, 5, 300), (10, 20, 16, 2, 16),
                                         (2, 2, 100, 3, 33), (1, 2, 256, 2, 8), (2, 1, 16, 32, 12)])
def test_encode_shapes(pkg, oracle, T, D, m, lam, d):
    rng = np.random.default_rng(T * 100 + m)
    S = rng.standard_normal((200, d))
    alpha, r, w = oracle.registry_init(S, m, 7, T, D)
    o = oracle.Oracle(T, D, m, lam, d)
    o.set_gfunctions(alpha, r, w)
    Q = (rng.standard_normal((19, d)) * 4).astype(np.float32)
    with _ctx(pkg, d, T=T, D=D, m=m, lam=lam) as ctx:
        ctx.set_gfunctions(alpha, r, w)
        codes, hs = ctx.encode(Q, want_hashes=True)
        codes64 = ctx.encode(Q.astype(np.float64))
    assert np.array_equal(hs, o.hashes(Q.astype(np.float64)))
    assert np.array_equal(codes, o.encode(Q.astype(np.float64)))
    assert np.array_equal(codes64, codes)


def test_encode_saturation_and_negative_hashes(pkg, oracle):
    T, D, m, lam, d = 1, 2, 12, 4, 6
    rng = np.random.default_rng(5)
    alpha, r, w = oracle.registry_init(rng.standard_normal((50, d)) * 1e-3, m, 3, T, D)   # tiny omega
    o = oracle.Oracle(T, D, m, lam, d)
    o.set_gfunctions(alpha, r, w)
    Q = np.stack([np.full(d, 1e300), np.full(d, -1e300), np.full(d, 1e12), -np.arange(d) * 7.0, np.zeros(d),
                  np.full(d, 2147483647.0 * w.max()), rng.standard_normal(d) * 1e6])
    H = o.hashes(Q)
    assert (H == 2**31 - 1).any() and (H == -2**31).any() and (H < 0).any()               # (int) cast saturates
    with _ctx(pkg, d, T=T, D=D, m=m, lam=lam) as ctx:
        ctx.set_gfunctions(alpha, r, w)
        codes, hs = ctx.encode(Q, want_hashes=True)
    assert np.array_equal(hs, H)
    assert np.array_equal(codes, o.encode(Q))


def test_encode_rejects_nan_inf(pkg, oracle):
    T, D, m, lam, d = 2, 1, 8, 2, 10
    alpha, r, w = oracle.registry_init(np.random.default_rng(6).standard_normal((50, d)), m, 3, T, D)
    with _ctx(pkg, d, T=T, D=D, m=m, lam=lam) as ctx:
        with pytest.raises(pkg.FspannStateError):       # GFunctionRegistry not initialized
            ctx.encode(np.zeros((1, d)))
        ctx.set_gfunctions(alpha, r, w)
        Q = np.zeros((5, d))
        Q[3, 2] = np.nan
        with pytest.raises(pkg.FspannArgumentError, match="NaN/Inf"):
            ctx.encode(Q)
        Q[3, 2] = -np.inf
        with pytest.raises(ValueError):
            ctx.encode(Q.astype(np.float32))
        with pytest.raises(pkg.FspannArgumentError):
            ctx.encode(np.zeros(d + 1))


def test_encode_bulk_path(pkg, oracle):
    """nq >= 8192 takes the 8-queries-per-block variant (index coding)."""
    T, D, m, lam, d = 2, 2, 16, 2, 24
    rng = np.random.default_rng(8)
    X = rng.standard_normal((9000, d)).astype(np.float32)
    alpha, r, w = oracle.registry_init(X[:500].astype(np.float64), m, 13, T, D)
    o = oracle.Oracle(T, D, m, lam, d)
    o.set_gfunctions(alpha, r, w)
    with _ctx(pkg, d, T=T, D=D, m=m, lam=lam) as ctx:
        ctx.set_gfunctions(alpha, r, w)
        assert np.array_equal(ctx.encode(X), o.encode(X.astype(np.float64)))


def test_native_registry_initialize_matches_oracle(pkg, oracle):
    T, D, m, lam, d = 3, 2, 10, 2, 20
    S = np.random.default_rng(9).standard_normal((1000, d)).astype(np.float32).astype(np.float64)
    a0, r0, w0 = oracle.registry_init(S, m, 13, T, D)
    with _ctx(pkg, d, T=T, D=D, m=m, lam=lam) as ctx:
        ctx.registry_initialize(S, 13)
        a, r, w = ctx.get_gfunctions()
    assert np.array_equal(w, w0) and np.array_equal(r, r0)      # exact fp64 projections on the GPU
    assert np.array_equal(a, a0)                                # same libm on this box; else compare to 1e-15


@pytest.mark.parametrize("B,k,d,dt", [(256, 10, 128, np.float32), (600, 40, 30, np.float32), (64, 10, 960, np.float32),
                                      (256, 10, 128, np.float64), (300, 7, 33, np.float64)])
def test_refine_from_resident_store(pkg, oracle, B, k, d, dt):
    """fspann_refine_store: rows addressed by id in the resident store == the dense [nq][B][d] refine == oracle.
    Ids outside the store are points that failed to load (QSI:252-256): skipped, not scored."""
    rng = np.random.default_rng(B + k + d)
    n, nq = 5000, 6
    store = rng.standard_normal((n, d)).astype(dt)
    store[17, d // 2] = np.inf                      # a corrupt stored row is skipped like any non-finite candidate
    q = rng.standard_normal((nq, d)).astype(dt)
    ids = rng.integers(0, n, (nq, B)).astype(np.int32)
    ids[0, 5] = 17
    ids[1, 0] = -1
    ids[1, B - 1] = n                               # one past the end
    ids[2, 3] = 2**31 - 1
    cnt = np.array([B, B, B, 0, 1, B // 2 + 1], np.int32)
    bad = (ids < 0) | (ids >= n)
    dense = store[np.where(bad, 0, ids)].copy()
    dense[bad] = np.nan                             # the oracle skips non-finite rows: same effect as "not loaded"
    ref_ids, ref_dist, ref_cnt = oracle.refine(q.astype(np.float64), dense.astype(np.float64), ids, cnt, k)
    with _ctx(pkg, d, B=B) as ctx:
        with pytest.raises(pkg.FspannStateError):
            ctx.refine_store(q, ids, cnt, k)        # no store yet
        ctx.store_set(store)
        res = ctx.refine_store(q, ids, cnt, k)
        via_dense = ctx.refine(q, dense, ids, cnt, k)
    for key in ("count", "ids", "dist", "scored"):
        assert np.array_equal(res[key], via_dense[key]), key
    assert np.array_equal(res["count"], ref_cnt)
    assert np.array_equal(res["ids"], ref_ids)
    assert np.array_equal(res["dist"], ref_dist)
