"""GPU: frozen-index file round trip: save -> load into a fresh context -> identical routing."""
import numpy as np
import pytest

from conftest import make_scene

pytestmark = pytest.mark.gpu


def test_save_load_roundtrip(pkg, oracle, tmp_path):
    sc = make_scene(oracle, n=5000, d=16, T=3, D=2, m=10, lam=2, B=64, seed=3, deleted_frac=0.05)
    p = sc["params"]
    cfg = pkg.PaperRuntimeConfig(tables=p["T"], divisions=p["D"], m=p["m"], lambda_=p["lam"], dim=p["d"], refinement_limit=p["B"])
    path = str(tmp_path / "index.fspann")
    Q = sc["rng"].standard_normal((16, p["d"])).astype(np.float32)
    with pkg.FspannContext(cfg, 0) as a:
        with pytest.raises(pkg.FspannStateError):
            a.save_index(path)                      # not finalized yet
        a.set_gfunctions(sc["alpha"], sc["r"], sc["omega"])
        a.set_id_meta(p["n"], None, sc["deleted"])
        a.build_index(sc["X"])
        a.save_index(path)
        ra = a.route(a.encode(Q), limit=64)
    with pkg.FspannContext(cfg, 0) as b:
        b.load_index(path)
        for td in range(b.TD):
            x, y = b.get_index(td), sc["oracle"].get_index(td)
            for k in x:
                assert np.array_equal(x[k], y[k])
        rb = b.route(b.encode(Q), limit=64)
    for k in ra:
        assert np.array_equal(ra[k], rb[k])
    bad = pkg.PaperRuntimeConfig(tables=p["T"], divisions=p["D"], m=p["m"] + 1, lambda_=p["lam"], dim=p["d"])
    with pkg.FspannContext(bad, 0) as c:
        with pytest.raises(pkg.FspannStateError):
            c.load_index(path)
        with pytest.raises(pkg.FspannArgumentError):
            c.load_index(str(tmp_path / "missing"))
