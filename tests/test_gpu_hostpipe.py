"""GPU: the native host candidate pipeline — Route on the GPU, AES-256-GCM open of F_q's records on host
threads into pinned staging, H2D, Refine — against the oracle's QueryServiceImpl.search (QSI:101-352), and BASELINE
config #5: the same query stream while another thread rotates the key and migrates records
(keymanagement/KeyRotationServiceImpl.java:215-305) must return bit-identical results (routing state never changes)."""
import threading
import time

import numpy as np
import pytest

from conftest import make_scene

pytestmark = pytest.mark.gpu
K = 10


def _ctx(pkg, sc):
    p = sc["params"]
    cfg = pkg.PaperRuntimeConfig(tables=p["T"], divisions=p["D"], m=p["m"], lambda_=p["lam"], dim=p["d"], refinement_limit=p["B"])
    ctx = pkg.FspannContext(cfg, 0)
    ctx.set_gfunctions(sc["alpha"], sc["r"], sc["omega"])
    ctx.set_id_meta(p["n"])
    ctx.build_index(sc["X"])
    return ctx


def _run(pl, batches):
    out = []
    for qb in batches:                       # keep the pipeline full: submit ahead, collect in order
        pl.submit(qb)
        if pl.in_flight == 4:
            out.append(pl.collect())
    while pl.in_flight:
        out.append(pl.collect())
    assert [o["ticket"] for o in out] == sorted(o["ticket"] for o in out)
    return out


def test_pipeline_matches_oracle_with_failed_loads(pkg, oracle):
    from fspann_amd import hostpipe
    sc = make_scene(oracle, n=30000, d=64, T=8, D=1, m=12, lam=2, B=256, seed=11)
    p, o = sc["params"], sc["oracle"]
    rng = sc["rng"]
    valid = np.ones(p["n"], np.uint8)
    gone = rng.choice(p["n"], 3000, replace=False)
    valid[gone] = 0
    o.set_store(sc["X64"], valid)                                    # loadPointIfActive() == null / decrypt failure for those
    batches = [rng.standard_normal((nq, 64)).astype(np.float32) for nq in (200, 128, 7, 200, 200, 33)]
    with _ctx(pkg, sc) as ctx, hostpipe.PointStore(p["n"], 64) as ps:
        ps.encrypt(sc["X"], threads=8)
        for h in gone[:1500]:
            ps.delete(int(h))                                        # no record
        for h in gone[1500:]:
            ver, iv, ct = ps.get_record(int(h))
            ps.put_record(int(h), ver, iv, ct[:-1] + bytes([ct[-1] ^ 0x80]))   # tag mismatch
        with hostpipe.Pipeline(ctx, ps, 200, p["B"], K, host_threads=8) as pl:
            out = _run(pl, batches)
            st = pl.stats()
        assert st["batches"] == len(batches) and st["decrypt_ms"] > 0
    for qb, res in zip(batches, out):
        ref = o.search(qb.astype(np.float64), K)
        assert np.array_equal(res["ids"], ref["ids"]) and np.array_equal(res["dist"], ref["dist"]) and np.array_equal(res["count"], ref["count"])
    assert not o.unmodelled


@pytest.mark.fullsize
def test_config5_live_rotate_migrate_at_sift1m_shape(pkg, oracle):
    """BASELINE config #5 at its stated size: 1 M x 128, 16 x 32 bits, B = 256, 1024-query batches, host AES-GCM re-encrypt
    running concurrently with GPU Route / Refine."""
    from fspann_amd import hostpipe
    n, d, T, m, B, Q = 1_000_000, 128, 16, 16, 256, 1024
    rng = np.random.default_rng(5)
    X = rng.standard_normal((n, d), dtype=np.float32)
    batches = [rng.standard_normal((Q, d), dtype=np.float32) for _ in range(6)]
    cfg = pkg.PaperRuntimeConfig(tables=T, divisions=1, m=m, lambda_=2, dim=d, refinement_limit=B)
    with pkg.FspannContext(cfg, 0) as ctx, hostpipe.PointStore(n, d) as ps:
        ctx.registry_initialize(X[:1000].astype(np.float64))
        ctx.set_id_meta(n)
        ctx.build_index(X)
        ps.encrypt(X)
        codes = ctx.encode(batches[0])
        routed_before = ctx.route(codes, limit=B, counters=False)
        with hostpipe.Pipeline(ctx, ps, Q, B, K) as pl:
            quiet = _run(pl, batches)                                 # reference run: nothing else touches the store
            # spot-check the quiet run against the oracle's Refine on the routed rows (plaintext)
            sel = routed_before["ids"][:64, :B]
            oi, od, oc = oracle.refine(batches[0][:64].astype(np.float64), X[sel].astype(np.float64), sel, routed_before["count"][:64], K)
            assert np.array_equal(quiet[0]["ids"][:64], oi) and np.array_equal(quiet[0]["dist"][:64], od)
            # live Rotate + Migrate: rotateKeyOnly, then reencryptTouched over every id, in slices, while queries stream
            moved, stop = [0], [False]
            assert ps.rotate() == 2

            def migrate():
                order = np.random.default_rng(9).permutation(n).astype(np.int32)
                for s in range(0, n, 20000):
                    if stop[0]:
                        return
                    moved[0] += ps.reencrypt(order[s:s + 20000], threads=8)

            th = threading.Thread(target=migrate)
            th.start()
            live = []
            t0 = time.time()
            while th.is_alive() and time.time() - t0 < 25:           # keep querying for as long as the migration runs
                live.append(_run(pl, batches))
            stop[0] = True
            th.join()
            live.append(_run(pl, batches))                            # and once more after it has finished
        assert moved[0] > 100000, moved
        for run in live:
            for a, b in zip(run, quiet):
                assert np.array_equal(a["ids"], b["ids"]) and np.array_equal(a["dist"], b["dist"]) and np.array_equal(a["count"], b["count"])
        routed_after = ctx.route(codes, limit=B, counters=False)      # the routed candidate sets never changed
        assert np.array_equal(routed_after["ids"], routed_before["ids"]) and np.array_equal(routed_after["count"], routed_before["count"])
        assert ps.stats()["failed"] == 0
        versions = {ps.get_record(int(h))[0] for h in rng.integers(0, n, 200)}
        assert versions <= {1, 2} and 2 in versions
