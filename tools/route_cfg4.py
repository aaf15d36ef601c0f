"""Dev measurement: Route at BASELINE config #4's routing shape (32 tables x 64-bit codes, B = 1024, 5 probes) on a 2 M-point index:
full select (probe + select kernels) against the bounded select's 2048-entry class."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pkg = g.load_package()
n, d, T, m, lam, B, Q = 2_000_000, 64, 32, 32, 2, 1024, 1024
rng = np.random.default_rng(4)
X = rng.standard_normal((n, d), dtype=np.float32)
cfg = pkg.PaperRuntimeConfig(tables=T, divisions=1, m=m, lambda_=lam, dim=d, refinement_limit=B)
ctx = pkg.FspannContext(cfg, 0)
ctx.registry_initialize(X[:1000].astype(np.float64)); ctx.set_id_meta(n); ctx.build_index(X)
dev = torch.device("cuda", 0)
F32 = pkg._native.F32
qs = torch.from_numpy(rng.standard_normal((8, Q, d), dtype=np.float32)).to(dev)
codes = torch.zeros((Q, T, 1), dtype=torch.int64, device=dev)
bad = torch.zeros(Q, dtype=torch.int32, device=dev)
sel = [torch.zeros((Q, B), dtype=torch.int32, device=dev) for _ in range(2)]
cnt = [torch.zeros(Q, dtype=torch.int32, device=dev) for _ in range(2)]
for mode, name in ((1, "full select"), (0, "bounded select (auto)")):
    ctx.set_route_mode(mode)
    for i in range(4):
        ctx.encode_dev(Q, qs[i % 8].data_ptr(), F32, codes.data_ptr(), 0, bad.data_ptr())
        ctx.route_dev(Q, codes.data_ptr(), -1, B, B, sel[mode].data_ptr(), 0, cnt[mode].data_ptr(), 0, 0)
    ctx.sync()
    t0 = time.perf_counter()
    for i in range(40):
        ctx.route_dev(Q, codes.data_ptr(), -1, B, B, sel[mode].data_ptr(), 0, cnt[mode].data_ptr(), 0, 0)
    ctx.sync()
    print("%-24s %.1f us per 1024 queries  %s" % (name, (time.perf_counter() - t0) / 40 * 1e6, ctx.last_route_info()))
ok = torch.equal(cnt[0], cnt[1]) and torch.equal(sel[0], sel[1])
print("identical results:", ok)
