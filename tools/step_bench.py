"""Dev tool: whole-step time of fspann_search_store_dev, no events in the loop (one sync at the end)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pkg = g.load_package()
n, d, T, m, lam, B, k, Q = 1000000, 128, 16, 16, 2, 256, 10, 1024
rng = np.random.default_rng(1)
X = rng.standard_normal((n, d), dtype=np.float32)
ctx = pkg.FspannContext(pkg.PaperRuntimeConfig(tables=T, divisions=1, m=m, lambda_=lam, dim=d, refinement_limit=B), 0)
ctx.registry_initialize(X[:1000].astype(np.float64)); ctx.set_id_meta(n); ctx.build_index(X); ctx.store_set(X)
dev = torch.device("cuda", 0)
qs = [torch.randn((Q, d), device=dev) for _ in range(8)]
oi = torch.zeros((Q, k), dtype=torch.int32, device=dev); od = torch.zeros((Q, k), dtype=torch.float64, device=dev); oc = torch.zeros(Q, dtype=torch.int32, device=dev)
F32 = pkg._native.F32
def run(reps):
    for i in range(reps):
        ctx.search_store_dev(Q, qs[i % 8].data_ptr(), F32, -1, B, k, oi.data_ptr(), od.data_ptr(), oc.data_ptr())
run(40); ctx.sync()
t0 = time.perf_counter(); run(400); ctx.sync(); t1 = time.perf_counter()
print("%.2f us/step  %.2f M queries/s" % (1e6 * (t1 - t0) / 400, Q * 400 / (t1 - t0) / 1e6))
ctx.refine_timing_begin(400, 6)
t0 = time.perf_counter(); run(400); ctx.sync(); t1 = time.perf_counter()
nl, ms = ctx.refine_timing_end()
print("with kernel-attached events on every 6th refine dispatch: %.2f us/step (%d timed, avg %.2f us)" % (1e6 * (t1 - t0) / 400, nl, 1e3 * ms / max(nl, 1)))
sel = torch.zeros((Q, B), dtype=torch.int32, device=dev); selc = torch.zeros(Q, dtype=torch.int32, device=dev); sc = torch.zeros(Q, dtype=torch.int32, device=dev); bad = torch.zeros(Q, dtype=torch.int32, device=dev)
def run2(reps):
    for i in range(reps):
        ctx.search_store_dev(Q, qs[i % 8].data_ptr(), F32, -1, B, k, oi.data_ptr(), od.data_ptr(), oc.data_ptr(), sc.data_ptr(), sel.data_ptr(), selc.data_ptr(), bad.data_ptr())
run2(20); ctx.sync()
t0 = time.perf_counter(); run2(400); ctx.sync(); t1 = time.perf_counter()
print("with the optional outputs: %.2f us/step" % (1e6 * (t1 - t0) / 400))
