#!/usr/bin/env python3
"""Generates tests/golden/*.npz from the CPU oracle (oracle/fspann_oracle.cpp).

PROVENANCE: the reference cannot run in the build container (no JVM) and ships no golden
vectors for this path, so these fixtures are produced by the C++ restatement of the
reference ("parity unpinned", see the oracle header).  They pin (a) the oracle against
regressions and (b) the HIP path against the oracle without needing the oracle at run time.
A 30-line Java dumper run on a JVM must reproduce them before Java<->native parity is claimed.

Scenes follow the shapes of the reference's own integration tests:
  quickcheck   index/src/test/java/com/fspann/index/CodingQuickCheck.java:10-37
  it_multi     it/.../MultiTableSystemIntegrationTest.java:40-51,101-127   (20 pts, d=6, T=3, D=4, m=6, lambda=3)
  it_smoke     it/.../ForwardSecureANNQuerySmokeIT.java:60-96              (1000 pts (i*1e-3, i*1e-3), d=2)
  it_unified   it/.../BaseUnifiedIT.java:47-59,113-130                    (1024 clustered pts, d=8, T=2, D=4, m=4, lambda=3)
  baseline_cfg1  BASELINE.json configs[0]: 10k x 128, 8 tables x 16 bits, B=64, 100 queries, k=10

Usage: python tests/golden/make_golden.py   (writes next to this file)
"""
import hashlib
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402


def splitmix_u8(n, d, seed):
    """Deterministic SIFT-like data: integers 0..255 from a SplitMix64 stream (pure uint64 numpy)."""
    with np.errstate(over="ignore"):
        idx = np.arange(1, n * d + 1, dtype=np.uint64)
        z = np.uint64(seed) + idx * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return (z >> np.uint64(56)).astype(np.float64).reshape(n, d)


def digest(*arrays):
    h = hashlib.sha256()
    for a in arrays:
        h.update(np.ascontiguousarray(a).tobytes())
    return np.frombuffer(h.digest(), dtype=np.uint8).copy()


def scene(name, X, Q, T, D, m, lam, B, K, seed, hard_cap=20000, store_index=True, route_q=None):
    n, d = X.shape
    alpha, r, w = O.registry_init(X[: min(n, 1000)], m, seed, T, D)
    o = O.Oracle(T, D, m, lam, d, max_global_candidates=hard_cap, refinement_limit=B)
    o.set_gfunctions(alpha, r, w)
    o.set_id_meta(n)
    o.set_store(X)
    o.build_index(X)
    assert not o.unmodelled
    out = dict(T=T, D=D, m=m, lam=lam, B=B, K=K, seed=seed, hard_cap=hard_cap, n=n, d=d,
               Q=Q, alpha=alpha, r=r, omega=w)
    codes = o.encode(Q)
    out["codes"] = codes
    out["hashes"] = o.hashes(Q)
    idx_digest = []
    for td in range(T * D):
        ix = o.get_index(td)
        idx_digest.append(digest(ix["min_key"], ix["max_key"], ix["rep"], ix["id_off"], ix["ids"]))
        if store_index:
            for k, v in ix.items():
                out[f"index{td}_{k}"] = v
    out["index_digest"] = np.stack(idx_digest)
    for probes in (5, 10):
        ids, score, count, raw = o.route(codes[:route_q], probe_override=probes)
        out[f"route{probes}_ids"] = ids
        out[f"route{probes}_score"] = score
        out[f"route{probes}_count"] = count
        out[f"route{probes}_raw"] = raw
    res = o.search(Q, K, codes=codes)
    assert not o.unmodelled
    for k in ("ids", "dist", "count", "sel", "sel_count", "metrics"):
        out["search_" + k] = res[k]
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(name, "n=%d d=%d TD=%d" % (n, d, T * D), "retried:", int(res["metrics"][:, 4].sum()), "of", len(Q))


def main():
    O.build()
    commit = subprocess.run(["git", "-C", ROOT, "rev-parse", "HEAD"], capture_output=True, text=True).stdout.strip()
    # --- quickcheck ---------------------------------------------------------------------------------
    d, m, lam = 128, 24, 2
    v = np.arange(128) * 0.01
    alpha, r, w = O.build_random_g(d, m, 1.0, 12345)
    H = O.H(v, alpha, r, w)
    code = O.Ccode(v, alpha, r, w, lam)
    np.savez_compressed(os.path.join(HERE, "quickcheck.npz"), v=v, alpha=alpha, r=r, omega=w, H=H, code=code,
                        alpha00_03=alpha[0, :4], generator_commit=np.array(commit))
    print("quickcheck H[0..3] =", H[:4], "code =", hex(int(code[0])))
    # --- IT-shaped scenes ------------------------------------------------------------------------------
    X = np.array([[i + j for j in range(6)] for i in range(20)], dtype=np.float64)
    scene("it_multi", X, np.array([[5.0] * 6, [0.0] * 6, [19.5] * 6]), T=3, D=4, m=6, lam=3, B=20, K=5, seed=42)
    X = np.array([[i * 1e-3, i * 1e-3] for i in range(1000)], dtype=np.float64)
    scene("it_smoke", X, np.array([[0.5, 0.5], [0.0, 0.0], [0.999, 0.9985], [2.0, -1.0]]), T=3, D=2, m=4, lam=2, B=100, K=10,
          seed=13)
    rng = np.random.default_rng(42)
    X = (5 + 0.1 * rng.standard_normal((1024, 8))).astype(np.float32).astype(np.float64)
    Q = (5 + 0.1 * rng.standard_normal((8, 8))).astype(np.float32).astype(np.float64)
    scene("it_unified", X, Q, T=2, D=4, m=4, lam=3, B=128, K=10, seed=42)
    # --- BASELINE config #1 (data regenerated from the seed by the tests: splitmix_u8) ---------------------
    X = splitmix_u8(10000, 128, 1)
    Q = splitmix_u8(100, 128, 2)
    scene("baseline_cfg1", X, Q, T=8, D=1, m=8, lam=2, B=64, K=10, seed=13, store_index=False, route_q=12)


if __name__ == "__main__":
    main()
