"""Dev tool: host (CPU) time per library call — tiny batches, so the GPU is never the limit."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pkg = g.load_package()
n, d, T, m, lam, B, k, Q = 100000, 128, 16, 16, 2, 256, 10, 8
rng = np.random.default_rng(1)
X = rng.standard_normal((n, d), dtype=np.float32)
ctx = pkg.FspannContext(pkg.PaperRuntimeConfig(tables=T, divisions=1, m=m, lambda_=lam, dim=d, refinement_limit=B), 0)
ctx.registry_initialize(X[:1000].astype(np.float64)); ctx.set_id_meta(n); ctx.build_index(X); ctx.store_set(X)
dev = torch.device("cuda", 0)
q = torch.randn((Q, d), device=dev)
codes = torch.zeros((Q, T, 1), dtype=torch.int64, device=dev); bad = torch.zeros(Q, dtype=torch.int32, device=dev)
sel = torch.zeros((Q, B), dtype=torch.int32, device=dev); cnt = torch.zeros(Q, dtype=torch.int32, device=dev)
oi = torch.zeros((Q, k), dtype=torch.int32, device=dev); od = torch.zeros((Q, k), dtype=torch.float64, device=dev); oc = torch.zeros(Q, dtype=torch.int32, device=dev)
F32 = pkg._native.F32
def t(name, f, reps=300):
    for _ in range(20): f()
    ctx.sync(); t0 = time.perf_counter()
    for _ in range(reps): f()
    t1 = time.perf_counter(); ctx.sync(); t2 = time.perf_counter()
    print(f"{name:28s} issue {1e6*(t1-t0)/reps:7.1f} us/call   incl. drain {1e6*(t2-t0)/reps:7.1f}")
t("encode_dev", lambda: ctx.encode_dev(Q, q.data_ptr(), F32, codes.data_ptr(), 0, bad.data_ptr()))
t("route_dev (bounded)", lambda: ctx.route_dev(Q, codes.data_ptr(), -1, B, B, sel.data_ptr(), 0, cnt.data_ptr(), 0, 0))
t("refine_store_dev", lambda: ctx.refine_store_dev(Q, q.data_ptr(), F32, B, sel.data_ptr(), cnt.data_ptr(), k, oi.data_ptr(), od.data_ptr(), oc.data_ptr()))
t("search_store_dev", lambda: ctx.search_store_dev(Q, q.data_ptr(), F32, -1, B, k, oi.data_ptr(), od.data_ptr(), oc.data_ptr()))
e = torch.cuda.Event()
s = torch.cuda.ExternalStream(ctx.stream)
t("event record", lambda: e.record(s))
