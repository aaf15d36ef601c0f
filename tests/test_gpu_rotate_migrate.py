"""BASELINE config #5 at test scale: live Rotate + Migrate during a query stream.

Host AES-256-GCM re-encryption (OpenSSL, tests/aesgcm_host.py) runs in a background thread while the
GPU serves Route/Refine; the routed candidate sets and the results must be bit-identical before,
during and after — the paper's forward-security invariant (routing state never depends on keys;
key/.../KeyRotationServiceImpl.java:215-305 only touches metadata + ciphertexts).  Mirrors
it/adversarial/ForwardSecurityGameTest.java:174-327 (old key cannot decrypt after re-encryption;
ciphertext changes) and the @Disabled G6 routing-invariance check (:132-152)."""
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_aes_gcm_roundtrip_and_aad_binding(pkg):
    from aesgcm_host import AesGcmHost, AuthError
    h = AesGcmHost(master=b"\x01" * 32)
    v = np.arange(16) * 0.25
    ep = h.encrypt("42", v)
    assert len(ep.iv) == 12 and len(ep.ciphertext) == 8 * 16 + 16
    assert np.array_equal(h.decryptFromPoint(ep, h.getVersion(ep.version).key), v)
    with pytest.raises(AuthError):                      # wrong key
        h.decryptFromPoint(ep, h.getVersion(2).key)
    ep2 = h.ops.EncryptedPoint("43", ep.version, ep.iv, ep.ciphertext, ep.dim)
    with pytest.raises(AuthError):                      # AAD binds the id
        h.decryptFromPoint(ep2, h.getVersion(ep.version).key)
    assert h.encrypt("42", v).ciphertext != ep.ciphertext   # fresh IV


def test_rotate_and_migrate_during_query_stream(pkg):
    from aesgcm_host import AesGcmHost, AuthError
    from fspann_amd import operators as ops
    n, d, K = 6000, 24, 10
    rng = np.random.default_rng(17)
    centers = rng.standard_normal((40, d)) * 4
    X = (centers[rng.integers(0, 40, n)] + 0.3 * rng.standard_normal((n, d))).astype(np.float32).astype(np.float64)
    Q = (centers[rng.integers(0, 40, 24)] + 0.3 * rng.standard_normal((24, d))).astype(np.float32).astype(np.float64)
    cfg = ops.SystemConfig(m=12, lambda_=2, divisions=2, tables=4, seed=13, refinementLimit=128, kVariants=(K,))
    host = AesGcmHost()
    ops.GFunctionRegistry.reset()
    index = ops.PartitionedIndexService(host, cfg, host, host)
    try:
        for i in range(n):
            index.insert(str(i), X[i])
        index.finalizeForSearch()
        tf = ops.QueryTokenFactory(host, host, cfg)
        qs = ops.QueryServiceImpl(index, host, host, tf, cfg)
        tokens = [tf.create(q, K) for q in Q]

        def run_all():
            out = []
            for t in tokens:
                res = qs.search(t)
                routed = tuple(index.lookupCandidateIds(t)[:cfg.refinementLimit])
                out.append((routed, [(r.id, r.distance) for r in res]))
            return out

        base = run_all()
        assert all(len(r[1]) == K for r in base)
        v1_cipher = {i: host.points[str(i)].ciphertext for i in (0, 1, 2)}

        # ---- Rotate, then Migrate everything in the background while queries keep running -------------
        stop = threading.Event()
        state = {}

        def migrate():
            v2 = host.rotateKeyOnly()
            state["n"] = host.reencrypt([str(i) for i in range(n)], v2)
            stop.set()

        th = threading.Thread(target=migrate)
        th.start()
        rounds = 0
        while not stop.is_set() or rounds == 0:
            assert run_all() == base, "routing/results changed during rotation"
            rounds += 1
        th.join()
        assert state["n"] == n and rounds >= 1
        # ---- Retire the old key: old ciphertexts are gone, old key is useless, results unchanged -------
        host.retire(1)
        assert all(host.points[str(i)].version == 2 for i in range(n))
        assert all(host.points[str(i)].ciphertext != v1_cipher[i] for i in v1_cipher)
        with pytest.raises(AuthError):
            host.decryptFromPoint(host.points["0"], AesGcmHost(master=host.master)._key(1))
        # tokens were encrypted under v1: the query decrypt falls back like QSI:124-129 -> recreate under v2
        tokens = [tf.create(q, K) for q in Q]
        after = run_all()
        assert after == base
    finally:
        if index.ctx is not None:
            index.ctx.close()
        ops.GFunctionRegistry.reset()
