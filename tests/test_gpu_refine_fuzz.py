"""GPU parity, randomized Refine: dense [nq][B][d] candidates and store-resident rows (by id), fp32 / fp64 stores and
queries, ragged counts, ids outside the store, non-finite rows and queries, B across the 256-row chunk boundary."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


import os

N_SEEDS = int(os.environ.get("FSPANN_FUZZ_SEEDS", "20"))


@pytest.mark.parametrize("seed", range(N_SEEDS))
def test_random_refine(pkg, oracle, seed):
    rng = np.random.default_rng(500 + seed)
    B = int(rng.choice([1, 7, 64, 255, 256, 257, 512, 700, 1100]))
    k = int(rng.choice([1, 3, 10, 32, 33, 64]))
    d = int(rng.choice([1, 3, 16, 31, 32, 96, 128, 200]))
    nq = int(rng.integers(1, 9))
    n = 3000
    sdt = np.float32 if rng.random() < 0.6 else np.float64
    qdt = np.float32 if rng.random() < 0.6 else np.float64
    store = rng.standard_normal((n, d)).astype(sdt)
    for r in rng.integers(0, n, 5):                      # a few corrupt stored rows
        store[r, rng.integers(0, d)] = rng.choice([np.nan, np.inf, -np.inf])
    q = rng.standard_normal((nq, d)).astype(qdt)
    if rng.random() < 0.3:
        q[rng.integers(0, nq), rng.integers(0, d)] = np.nan      # QSI:137-140: that query returns nothing
    ids = rng.integers(0, n, (nq, B)).astype(np.int32)
    bad = rng.random((nq, B)) < 0.05
    ids[bad] = rng.choice([-1, n, n + 17, 2**31 - 1], size=int(bad.sum()))
    cnt = rng.integers(0, B + 1, nq).astype(np.int32)
    cnt[rng.integers(0, nq)] = B
    invalid = (ids < 0) | (ids >= n)
    dense = store[np.where(invalid, 0, ids)].astype(np.float64)
    dense[invalid] = np.nan                               # "point failed to load": skipped, like a non-finite row
    ref_ids, ref_dist, ref_cnt = oracle.refine(q.astype(np.float64), dense, ids, cnt, k)
    qbad = ~np.isfinite(q).all(axis=1)                    # the query-level check sits in QSI.search, above oracle.refine
    ref_cnt[qbad] = 0
    ref_ids[qbad] = -1
    ref_dist[qbad] = np.inf
    cfg = pkg.PaperRuntimeConfig(tables=1, divisions=1, m=8, lambda_=2, dim=d, refinement_limit=B)
    with pkg.FspannContext(cfg, 0) as ctx:
        ctx.store_set(store)
        got = ctx.refine_store(q, ids, cnt, k)
        via_dense = ctx.refine(q.astype(sdt), dense.astype(sdt), ids, cnt, k) if sdt == qdt else None
    assert np.array_equal(got["count"], ref_cnt), (B, k, d)
    assert np.array_equal(got["ids"], ref_ids), (B, k, d)
    assert np.array_equal(got["dist"], ref_dist), (B, k, d)
    if via_dense is not None:
        for key in ("count", "ids", "dist", "scored"):
            assert np.array_equal(via_dense[key], got[key]), key
