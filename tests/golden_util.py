import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = ["it_multi", "it_smoke", "it_unified", "baseline_cfg1"]


def splitmix_u8(n, d, seed):
    """Same deterministic generator as tests/golden/make_golden.py (pure uint64 numpy)."""
    with np.errstate(over="ignore"):
        idx = np.arange(1, n * d + 1, dtype=np.uint64)
        z = np.uint64(seed) + idx * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return (z >> np.uint64(56)).astype(np.float64).reshape(n, d)


def load_scene_inputs(name):
    g = np.load(os.path.join(HERE, "golden", name + ".npz"))
    n, d = int(g["n"]), int(g["d"])
    if name == "it_multi":
        X = np.array([[i + j for j in range(6)] for i in range(20)], dtype=np.float64)
    elif name == "it_smoke":
        X = np.array([[i * 1e-3, i * 1e-3] for i in range(1000)], dtype=np.float64)
    elif name == "it_unified":
        rng = np.random.default_rng(42)
        X = (5 + 0.1 * rng.standard_normal((1024, 8))).astype(np.float32).astype(np.float64)
    elif name == "baseline_cfg1":
        X = splitmix_u8(10000, 128, 1)
    else:
        raise KeyError(name)
    assert X.shape == (n, d)
    return g, X
