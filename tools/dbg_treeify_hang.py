"""Dev: reproduce tests/test_gpu_treeify.py::test_treeified_queries_get_the_jdk_order step by step with progress lines."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import __graft_entry__ as g
from conftest import make_scene
import torch; torch.cuda.init()
pkg = g.load_package(); oracle = g.load_oracle(); oracle.build()
def log(*a): print(*a, flush=True)
def _spread_inv(s):
    s = np.asarray(s).astype(np.uint32); return (s ^ (s >> 16)).view(np.int32)
def crowded(rng, n, nbins, cap):
    bins = rng.choice(cap, nbins, replace=False)
    return _spread_inv(bins[rng.integers(0, nbins, n)] + cap * rng.permutation(n).astype(np.int64) % (1 << 31))
hard_cap, B = 20000, 64
for seed in range(6):
    rng = np.random.default_rng(900 + seed)
    n = int(rng.integers(1500, 6000))
    sc = make_scene(oracle, n=n, d=12, T=4, D=2, m=8, lam=2, B=B, hard_cap=hard_cap, seed=40 + seed)
    o = sc["oracle"]
    cap0 = oracle.table_size_for(min(max(hard_cap, B), 1 << 16))
    nb = [3, 20, 200, cap0, 8, cap0][seed]
    jh = crowded(rng, n, min(nb, cap0), cap0)
    o.set_id_meta(n, jh); o.build_index(sc["X64"])
    codes = o.encode(rng.standard_normal((32, 12)))
    for probes in (-1, 10):
        p = sc["params"]
        cfg = pkg.PaperRuntimeConfig(tables=p["T"], divisions=p["D"], m=p["m"], lambda_=p["lam"], dim=p["d"], refinement_limit=B, max_global_candidates=hard_cap)
        with pkg.FspannContext(cfg, 0) as ctx:
            ctx.set_gfunctions(sc["alpha"], sc["r"], sc["omega"]); ctx.set_id_meta(n, jh)
            for td in range(o.TD): ctx.set_index(td, **o.get_index(td))
            ctx.finalize()
            log("seed", seed, "probes", probes, "n", n, "flags ...")
            t = time.time(); fl = ctx.route_flags(codes, probe_override=probes); log("   flags done", int(fl.sum()), round(time.time() - t, 2))
            t = time.time(); res = ctx.route(codes, probe_override=probes); log("   route done", round(time.time() - t, 2), int(res["count"].max()))
log("all done")
