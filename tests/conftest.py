import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import __graft_entry__ as graft  # noqa: E402


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "fullsize: GPU parity at BASELINE.json's full sizes (minutes; part of -m gpu)")


@pytest.fixture(scope="session")
def oracle():
    """CPU oracle (test infrastructure)."""
    O = graft.load_oracle()
    O.build()
    return O


@pytest.fixture(scope="session")
def pkg():
    # torch ships its own copy of the HIP runtime: when both live in one process, torch's must be initialised first
    # (a second runtime initialised after libfspann_hip.so has opened the device sees no GPU)
    try:
        import torch
        if torch.cuda.is_available():
            torch.cuda.init()
    except ImportError:
        pass
    return graft.load_package()


def make_scene(O, n, d, T, D, m, lam, seed=0, B=64, hard_cap=20000, probe_override=-1, tau=0, clustered=False,
               dtype=np.float32, deleted_frac=0.0):
    """Synthetic scene shared by oracle and product tests: data, GFunctions (oracle-generated,
    handed to both sides), oracle instance with its index built."""
    rng = np.random.default_rng(seed)
    if clustered:
        X = (5 + 0.1 * rng.standard_normal((n, d))).astype(dtype)
    else:
        X = rng.standard_normal((n, d)).astype(dtype)
    X64 = X.astype(np.float64)
    alpha, r, w = O.registry_init(X64[: min(n, 1000)], m, 13, T, D)
    o = O.Oracle(T, D, m, lam, d, max_global_candidates=hard_cap, refinement_limit=B, probe_override=probe_override,
                 hamming_threshold=tau)
    o.set_gfunctions(alpha, r, w)
    deleted = None
    if deleted_frac > 0:
        deleted = (rng.random(n) < deleted_frac).astype(np.uint8)
    o.set_id_meta(n, None, deleted)
    o.set_store(X64)
    o.build_index(X64)
    return dict(X=X, X64=X64, alpha=alpha, r=r, omega=w, oracle=o, deleted=deleted, rng=rng,
                params=dict(n=n, d=d, T=T, D=D, m=m, lam=lam, B=B, hard_cap=hard_cap, probe_override=probe_override, tau=tau))
