"""Dev experiment: two contexts (two HIP streams) alternating batches, to overlap the latency-bound Route of
batch i+1 with the bandwidth-bound gather/refine of batch i."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pkg = g.load_package()
n, d, T, D, m, lam, B, Q, k = 1_000_000, 128, 16, 1, 16, 2, 256, 1024, 10
rng = np.random.default_rng(1)
X = rng.standard_normal((n, d), dtype=np.float32)
NCTX = int(os.environ.get("NCTX", "2"))
ctxs, bufs = [], []
F32 = pkg._native.F32
for c in range(NCTX):
    ctx = pkg.FspannContext(pkg.PaperRuntimeConfig(tables=T, divisions=D, m=m, lambda_=lam, dim=d, refinement_limit=B), 0)
    ctx.registry_initialize(X[:1000].astype(np.float64)); ctx.set_id_meta(n); ctx.build_index(X); ctx.store_set(X)
    ctxs.append(ctx)
    q = torch.from_numpy(np.random.default_rng(10 + c).standard_normal((Q, d), dtype=np.float32)).cuda()
    bufs.append(dict(q=q, codes=torch.zeros((Q, T * D, 1), dtype=torch.int64, device="cuda"), bad=torch.zeros(Q, dtype=torch.int32, device="cuda"),
                     sel=torch.zeros((Q, B), dtype=torch.int32, device="cuda"), cnt=torch.zeros(Q, dtype=torch.int32, device="cuda"),
                     kept=torch.zeros(Q, dtype=torch.int32, device="cuda"), raw=torch.zeros(Q, dtype=torch.int32, device="cuda"),
                     cand=torch.zeros((Q, B, d), device="cuda"), oi=torch.zeros((Q, k), dtype=torch.int32, device="cuda"),
                     od=torch.zeros((Q, k), dtype=torch.float64, device="cuda"), oc=torch.zeros(Q, dtype=torch.int32, device="cuda")))
torch.cuda.synchronize()
def step(i):
    ctx, b = ctxs[i % NCTX], bufs[i % NCTX]
    ctx.encode_dev(Q, b["q"].data_ptr(), F32, b["codes"].data_ptr(), 0, b["bad"].data_ptr())
    ctx.route_dev(Q, b["codes"].data_ptr(), -1, B, B, b["sel"].data_ptr(), 0, b["cnt"].data_ptr(), b["kept"].data_ptr(), b["raw"].data_ptr())
    ctx.store_gather_dev(Q, b["sel"].data_ptr(), b["cnt"].data_ptr(), B, b["cand"].data_ptr())
    ctx.refine_dev(Q, b["q"].data_ptr(), F32, b["cand"].data_ptr(), F32, B, b["sel"].data_ptr(), b["cnt"].data_ptr(), k, b["oi"].data_ptr(), b["od"].data_ptr(), b["oc"].data_ptr(), 0)
for i in range(10): step(i)
for c in ctxs: c.sync()
K = 200
t0 = time.perf_counter()
for i in range(K): step(i)
for c in ctxs: c.sync()
dt = time.perf_counter() - t0
print(f"NCTX={NCTX}: {dt/K*1e6:.1f} us/step  {Q*K/dt/1e6:.2f} M queries/s")
