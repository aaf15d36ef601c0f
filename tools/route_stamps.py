"""Dev tool: per-phase time of the route kernel (first query of each block)."""
import ctypes as C, sys, numpy as np, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pkg = g.load_package()
n, d, T, D, m, lam, B, Q = 1_000_000, 128, 16, 1, 16, 2, 256, 1024
rng = np.random.default_rng(1)
X = rng.standard_normal((n, d), dtype=np.float32)
Qh = np.random.default_rng(2).standard_normal((Q, d), dtype=np.float32)
ctx = pkg.FspannContext(pkg.PaperRuntimeConfig(tables=T, divisions=D, m=m, lambda_=lam, dim=d, refinement_limit=B), 0)
ctx.registry_initialize(X[:1000].astype(np.float64)); ctx.set_id_meta(n); ctx.build_index(X)
codes = torch.from_numpy(ctx.encode(Qh).view(np.int64)).cuda()
sel = torch.zeros((Q, B), dtype=torch.int32, device='cuda'); cnt = torch.zeros(Q, dtype=torch.int32, device='cuda')
kept = torch.zeros_like(cnt); raw = torch.zeros_like(cnt)
dbg = torch.zeros((1024, 16), dtype=torch.int64, device='cuda')
L = pkg._native.lib()
L.fspann_debug_route_stamps.argtypes = [C.c_void_p, C.c_void_p]
for it in range(3):
    L.fspann_debug_route_stamps(ctx.handle, dbg.data_ptr())
    ctx.route_dev(Q, codes.data_ptr(), -1, B, B, sel.data_ptr(), 0, cnt.data_ptr(), kept.data_ptr(), raw.data_ptr()); ctx.sync()
s = dbg.cpu().numpy().astype(np.float64)
s = s[s[:, 0] > 0]
print('blocks', len(s))
order = [(0,1,"reset+probe list"),(1,8,"stage ids"),(8,2,"hash build"),(2,9,"repeats/cap/n"),(9,3,"B3 repeats resolve"),(3,11,"C select lvl0/1"),(11,4,"C compaction"),(4,5,"C rank/sort+out"),(5,6,"tail")]
for a,b,nm in order:
    dt=(s[:,b]-s[:,a])/100.0
    print(f"{nm:24s} mean {dt.mean():8.2f} us   max {dt.max():8.2f}")
print("total/query mean", (s[:, 6] - s[:, 0]).mean() / 100.0, "us; nsel mean", s[:, 15].mean(), "max", s[:, 15].max())
import time
t0=time.perf_counter()
for _ in range(20):
    ctx.route_dev(Q, codes.data_ptr(), -1, B, B, sel.data_ptr(), 0, cnt.data_ptr(), kept.data_ptr(), raw.data_ptr())
ctx.sync()
print('route kernel avg ms', (time.perf_counter()-t0)/20*1e3)
print("kept mean", kept.float().mean().item(), "raw mean", raw.float().mean().item())
