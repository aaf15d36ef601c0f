"""GPU: exact brute-force ground truth and the evaluation metrics against the restatement of
GroundtruthPrecompute.run (api/.../GroundtruthPrecompute.java:142-189,218-272) and ForwardSecureANNSystem.computeMetricsAtK
(FSA:770-835): same ids in the same order (ties by lower id), bit-identical squared distances, recall and ratio."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _gt(pkg, X, Q, k):
    import torch
    dev = torch.device("cuda", 0)
    cfg = pkg.PaperRuntimeConfig(tables=1, divisions=1, m=4, lambda_=2, dim=X.shape[1])
    with pkg.FspannContext(cfg, 0) as ctx:
        xd, qd = torch.from_numpy(X).to(dev), torch.from_numpy(Q).to(dev)
        ids = torch.zeros((len(Q), k), dtype=torch.int32, device=dev)
        d2 = torch.zeros((len(Q), k), dtype=torch.float64, device=dev)
        torch.cuda.synchronize()
        ctx.groundtruth_dev(len(X), xd.data_ptr(), len(Q), qd.data_ptr(), X.shape[1], k, ids.data_ptr(), d2.data_ptr())
        ctx.sync()
        return ids.cpu().numpy(), d2.cpu().numpy()


@pytest.mark.parametrize("n,d,nq,k", [(5000, 128, 37, 10), (300, 7, 5, 100), (70000, 32, 20, 1), (40, 16, 3, 64)])
def test_groundtruth_matches_reference_semantics(pkg, oracle, n, d, nq, k):
    rng = np.random.default_rng(n + k)
    X = rng.standard_normal((n, d)).astype(np.float32)
    Q = rng.standard_normal((nq, d)).astype(np.float32)
    ids, d2 = _gt(pkg, X, Q, k)
    ref_ids, ref_d2 = oracle.groundtruth(X, Q, k)
    assert np.array_equal(ids, ref_ids)
    assert np.array_equal(d2, ref_d2)                       # same arithmetic: float subtraction, fp64 squares in order
    if k > n:
        assert (ids[:, n:] == -1).all()


def test_groundtruth_ties_go_to_the_lower_id(pkg, oracle):
    """SIFT-like integer data with duplicated vectors: many exactly equal distances."""
    rng = np.random.default_rng(7)
    X = rng.integers(0, 4, (3000, 8)).astype(np.float32)
    X[1000:2000] = X[:1000]                                  # every vector of the first block twice
    Q = rng.integers(0, 4, (25, 8)).astype(np.float32)
    ids, d2 = _gt(pkg, X, Q, 50)
    ref_ids, ref_d2 = oracle.groundtruth(X, Q, 50)
    assert np.array_equal(ids, ref_ids) and np.array_equal(d2, ref_d2)
    for i in range(len(Q)):                                  # ascending (distance, id)
        key = list(zip(d2[i], ids[i]))
        assert key == sorted(key)


def test_metrics_match_compute_metrics_at_k(pkg, oracle):
    import torch
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(3)
    n, d, nq, k = 4000, 24, 64, 10
    X = rng.standard_normal((n, d)).astype(np.float32)
    Q = rng.standard_normal((nq, d)).astype(np.float32)
    Q[5] = X[17]                                             # distance 0 to its nearest neighbour: ratio is NaN there (dGt <= 0 is skipped)
    gt, _ = oracle.groundtruth(X, Q, 20)
    ann = gt[:, :12].copy()
    for i in range(nq):                                      # an approximate answer: some true neighbours replaced
        m = rng.random(12) < 0.4
        ann[i, m] = rng.integers(0, n, int(m.sum()))
    cnt = np.full(nq, 12, np.int32)
    cnt[3], cnt[9] = 7, 0                                    # fewer than k results: ratio NaN, recall over what exists
    ann[11, 2] = -1                                          # an unparsable id
    cfg = pkg.PaperRuntimeConfig(tables=1, divisions=1, m=4, lambda_=2, dim=d)
    with pkg.FspannContext(cfg, 0) as ctx:
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
        xd, qd, ad, cd, gd = t(X), t(Q), t(ann), t(cnt), t(gt)
        rec = torch.zeros(nq, dtype=torch.float64, device=dev)
        rat = torch.zeros(nq, dtype=torch.float64, device=dev)
        torch.cuda.synchronize()
        ctx.eval_metrics_dev(n, xd.data_ptr(), nq, qd.data_ptr(), d, k, ad.data_ptr(), 12, cd.data_ptr(), gd.data_ptr(), 20, rec.data_ptr(), rat.data_ptr())
        ctx.sync()
        rec, rat = rec.cpu().numpy(), rat.cpu().numpy()
    ref_rec, ref_rat = oracle.metrics(X, Q, k, ann, cnt, gt)
    assert np.array_equal(rec, ref_rec)
    assert np.array_equal(np.isnan(rat), np.isnan(ref_rat)) and np.isnan(rat[[3, 5, 9, 11]]).all()
    ok = ~np.isnan(rat)
    assert np.array_equal(rat[ok], ref_rat[ok]) and (rat[ok] >= 1.0 - 1e-12).all()
