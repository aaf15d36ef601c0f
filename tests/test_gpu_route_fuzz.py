"""GPU parity, randomized: small random scenes (tables, divisions, code shape, probes, deletions, clustered data, id
hashes, limits) through the full select AND the bounded select, each against the oracle's Java-ordered list."""
import numpy as np
import pytest

from conftest import make_scene
from test_gpu_route_edges import LAZY_RUNS, check_route

pytestmark = pytest.mark.gpu


import os

N_SEEDS = int(os.environ.get("FSPANN_FUZZ_SEEDS", "24"))     # FSPANN_FUZZ_SEEDS=300 for a long run


@pytest.mark.parametrize("seed", range(N_SEEDS))
def test_random_scene(pkg, oracle, seed):
    rng = np.random.default_rng(1000 + seed)
    T = int(rng.integers(1, 9))
    D = int(rng.integers(1, 4))
    m = int(rng.choice([4, 6, 8, 12, 16, 20]))
    lam = int(rng.choice([1, 2, 3]))
    n = int(rng.choice([700, 3000, 9000, 20000]))
    d = int(rng.choice([8, 16, 33]))
    clustered = bool(rng.random() < 0.3)
    deleted = float(rng.choice([0.0, 0.0, 0.2, 0.6]))
    B = int(rng.choice([16, 64, 200, 256, 500]))
    probes = int(rng.choice([-1, -1, 1, 3, 10]))
    sc = make_scene(oracle, n=n, d=d, T=T, D=D, m=m, lam=lam, B=B, seed=2000 + seed, clustered=clustered, deleted_frac=deleted)
    sc["params"]["clustered"] = clustered
    jh = None
    if rng.random() < 0.4:      # arbitrary String ids: hashes spread enough that the reference HashMap never treeifies
        jh = rng.permutation(n).astype(np.int64) * 7919 % (2**31 - 1)
        jh = jh.astype(np.int32)
        o = sc["oracle"]
        o.set_id_meta(n, jh, sc["deleted"])
        o.build_index(sc["X64"])
    lims = sorted({B, max(1, B // 3), 1})
    before = len(LAZY_RUNS)
    check_route(pkg, sc, nq=12, probes=probes, limits=(None,) + tuple(lims), java_hash=jh)
    assert len(LAZY_RUNS) > before
