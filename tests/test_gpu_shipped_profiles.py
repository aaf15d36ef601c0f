"""GPU parity at the profiles the reference SHIPS and publishes numbers for (config/src/main/resources/config_sift1m.json:44-128,
logs/New Results:27-57) — the operating range of a real deployment, not the north-star's B = 256:

  SIFT_P4_FAST    m 20, lambda 2, divisions 8, tables 5 (40 (t,d) tables x 40 bits), probeOverride 4, refinementLimit 8 000,
                  maxGlobalCandidates 10 000 -> HARD_CAP 10 000 < 40*4*64 = 10 240 tuples: the cap CAN cut the traversal
  SIFT_P10_HIGH   m 26, divisions 8, tables 7 (56 tables x 52 bits), probeOverride 10, refinementLimit 22 000,
                  maxGlobalCandidates 28 000 -> HARD_CAP 28 000 < 35 840 tuples (the HARD_CAP-ordered path), and bestScore = new
                  HashMap(28 000) resizes 32 768 -> 65 536 on its 24 577th id; with ~25 k ids in 32 768 bins a bin treeifies in
                  ~0.3 % of the queries (finished by the host model)

at N = 200 000 clustered vectors, k = 100 (eval.kVariants' maximum: the token's topK) and k = 10: the oracle builds its own index,
every table is compared, then the full list (ids, scores, lastCandKept, rawSeen), the search call (full select + chunked scan +
merge kernel, k = 100 > the per-wave filter's 32) and the staged dense path against oracle.search.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

PROFILES = {
    "SIFT_P4_FAST": dict(T=5, D=8, m=20, lam=2, probes=4, B=8000, hard_cap=10000),
    "SIFT_P10_HIGH": dict(T=7, D=8, m=26, lam=2, probes=10, B=22000, hard_cap=28000),
}


def clustered(rng, n, d, nc=1024, sigma=0.15):
    C = rng.standard_normal((nc, d), dtype=np.float32)
    return C, (C[rng.integers(0, nc, n)] + np.float32(sigma) * rng.standard_normal((n, d), dtype=np.float32))


def siftlike(rng, n, d, r=16, noise=6.0):
    """bench.py's SIFT-like generator (integers 0..255 of intrinsic dimension r): candidate lists that overlap little — a few hundred
    repeats per query instead of thousands, scores crowded on a few levels: other paths of the full select than the blobs take."""
    U = (rng.standard_normal((r, d)) / np.sqrt(r)).astype(np.float32)
    def draw(cnt):
        y = rng.standard_normal((cnt, r), dtype=np.float32) @ U
        return np.clip(np.rint(np.float32(64.0) + np.float32(48.0) * y + np.float32(noise) * rng.standard_normal((cnt, d), dtype=np.float32)), 0, 255).astype(np.float32)
    return draw


@pytest.mark.parametrize("data", ["clustered", "siftlike"])
@pytest.mark.parametrize("name", sorted(PROFILES))
def test_shipped_profile_against_the_oracle(pkg, oracle, name, data):
    import torch
    pr = PROFILES[name]
    n, d, nq = 200_000, 128, 48
    T, D, m, lam, P, B, HC = (pr[k] for k in ("T", "D", "m", "lam", "probes", "B", "hard_cap"))
    rng = np.random.default_rng(11)
    if data == "clustered":
        C, X = clustered(rng, n, d)
        Q = (C[rng.integers(0, len(C), nq)] + np.float32(0.15) * rng.standard_normal((nq, d), dtype=np.float32))
    else:
        draw = siftlike(rng, n, d)
        X, Q = draw(n), draw(nq)
    X64 = X.astype(np.float64)
    alpha, r, w = oracle.registry_init(X64[:1000], m, 13, T, D)
    o = oracle.Oracle(T, D, m, lam, d, max_global_candidates=HC, refinement_limit=B, probe_override=P)
    o.set_gfunctions(alpha, r, w)
    o.set_id_meta(n)
    o.set_store(X64)
    o.build_index(X64)
    assert not o.unmodelled
    cfg = pkg.PaperRuntimeConfig(tables=T, divisions=D, m=m, lambda_=lam, dim=d, refinement_limit=B, max_global_candidates=HC, probe_override=P)
    dev = torch.device("cuda", 0)
    with pkg.FspannContext(cfg, 0) as ctx:
        ctx.set_gfunctions(alpha, r, w)
        ctx.set_id_meta(n)
        ctx.build_index(X)
        for td in (0, 7, T * D - 1):
            a, b = ctx.get_index(td), o.get_index(td)
            assert all(np.array_equal(a[k], b[k]) for k in a), td
        ctx.store_set(X)
        Q64 = Q.astype(np.float64)
        codes = ctx.encode(Q)
        assert np.array_equal(codes, o.encode(Q64))
        # ---- the whole list: lookupCandidatesWithScores (PIS:592-715), HARD_CAP rule and map resize included
        ids, score, count, raw = o.route(codes)
        full = ctx.route(codes)
        assert np.array_equal(full["count"], count) and np.array_equal(full["kept"], count) and np.array_equal(full["raw_seen"], raw)
        for i in range(nq):
            assert np.array_equal(full["ids"][i, :count[i]], ids[i, :count[i]]), (name, i)
            assert np.array_equal(full["score"][i, :count[i]], score[i, :count[i]])
        assert count.max() <= HC - 1 + 64 and count.min() > B // 4          # the profile's operating range, not a toy
        # ---- lookupCandidateIds: the same list truncated at HARD_CAP (PIS:558-565)
        capped = ctx.route(codes, limit=HC)
        assert np.array_equal(capped["count"], np.minimum(count, HC))
        for K in (100, 10):
            ref = o.search(Q64, K)
            assert not o.unmodelled and not ref["metrics"][:, 4].any()       # B >= 10 K: no adaptive retry
            qd = torch.from_numpy(Q).to(dev)
            oi = torch.full((nq, K), -7, dtype=torch.int32, device=dev)
            od = torch.zeros((nq, K), dtype=torch.float64, device=dev)
            oc = torch.zeros(nq, dtype=torch.int32, device=dev)
            scn = torch.zeros(nq, dtype=torch.int32, device=dev)
            sel = torch.full((nq, B), -1, dtype=torch.int32, device=dev)
            selc = torch.zeros(nq, dtype=torch.int32, device=dev)
            torch.cuda.synchronize()
            args = (nq, qd.data_ptr(), pkg._native.F32, -1, B, K, oi.data_ptr(), od.data_ptr(), oc.data_ptr(), scn.data_ptr(), sel.data_ptr(), selc.data_ptr())
            ctx.search_store_dev(*args)          # encode -> full select (limit = B) -> chunked scan over 32 / 86 chunks + merge
            ctx.search_store_finish_dev(*args)   # (finishes queries whose map treeified a bin: the host model)
            ctx.sync()
            assert ctx.unmodelled_queries() == 0
            sc_h = selc.cpu().numpy()
            assert np.array_equal(sc_h, ref["sel_count"])
            assert np.array_equal(np.where(np.arange(B)[None] < sc_h[:, None], sel.cpu().numpy(), -1), ref["sel"][:, :B])
            assert np.array_equal(oi.cpu().numpy(), ref["ids"]) and np.array_equal(od.cpu().numpy(), ref["dist"]), (name, K)
            assert np.array_equal(oc.cpu().numpy(), ref["count"]) and np.array_equal(scn.cpu().numpy(), ref["metrics"][:, 2])
        # ---- the staged boundary (what a JVM drives): F_q -> host rows -> fspann_refine (dense block, chunked scan + merge)
        sub = slice(0, 12)
        rt = ctx.route(codes[sub], limit=B)
        selh = np.where(np.arange(B)[None] < rt["count"][:, None], rt["ids"][:, :B], 0)
        out = ctx.refine(Q[sub], X[selh], rt["ids"][:, :B], rt["count"], 100)
        ref = o.search(Q64[sub], 100)
        assert np.array_equal(out["ids"], ref["ids"]) and np.array_equal(out["dist"], ref["dist"])
