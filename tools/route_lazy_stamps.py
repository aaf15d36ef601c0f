"""Dev tool: per-phase time of the bounded (lazy) route select kernel (first query of each block)."""
import ctypes as C, sys, numpy as np, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pkg = g.load_package()
if os.environ.get('AB_LIB'): pkg._native._SO = os.path.abspath(os.environ['AB_LIB'])
n, d, T, D, m, lam, B, Q = 1_000_000, 128, 16, 1, 16, 2, 256, 1024
rng = np.random.default_rng(1)
if os.environ.get("DATA"):          # bench.py's generators: clustered | siftlike:16:6
    import importlib.util
    _sp = importlib.util.spec_from_file_location("fspann_bench", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"))
    _b = importlib.util.module_from_spec(_sp); _sp.loader.exec_module(_b)
    X, _Q, _ = _b.make_data(os.environ["DATA"], n, d, 1, Q, 13, 0)
    Qh = np.ascontiguousarray(_Q[0])
else:
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
    ctx.route_dev(Q, codes.data_ptr(), -1, B, B, sel.data_ptr(), 0, cnt.data_ptr(), 0, 0); ctx.sync()
s = dbg.cpu().numpy().astype(np.float64)
s = s[s[:, 0] > 0]
print('blocks', len(s))
order = [(0,1,"probe list load"),(1,2,"probe sort"),(2,8,"loads + a: whole levels"),(8,10,"b: hash+present+hist"),(10,11,"b: cut"),(11,12,"b: insert"),(12,3,"levels tail"),(3,4,"keys copy"),(4,5,"rank"),(5,6,"collisions + out"),(6,7,"clear")]
for a,b,nm in order:
    dt=(s[:,b]-s[:,a])/100.0
    print(f"{nm:24s} mean {dt.mean():8.2f} us   max {dt.max():8.2f}")
print("total/query mean", (s[:, 7] - s[:, 0]).mean() / 100.0, "us; nsel mean", s[:, 15].mean(), "max", s[:, 15].max(), "u mean", s[:, 14].mean(), "ncoll mean", s[:, 13].mean(), "max", s[:, 13].max())
tot = (s[:, 7] - s[:, 0]) / 100.0
print("total/query percentiles 50/90/99/max:", np.percentile(tot, [50, 90, 99, 100]).round(1), " start skew (us) p50/max:", np.percentile((s[:,0]-s[:,0].min())/100.0,[50,100]).round(1), " end max:", ((s[:,7]-s[:,0].min())/100.0).max().round(1))
lv = (s[:, 3] - s[:, 2]) / 100.0
print("levels phase percentiles 50/90/99/max:", np.percentile(lv, [50, 90, 99, 100]).round(1))
d = s[:, 14].astype(np.int64)
outer, inner, reload_, nitb = d & 255, (d >> 8) & 255, (d >> 16) & 255, d >> 24
print("outer iters mean/max", outer.mean(), outer.max(), " inner mean/max", inner.mean(), inner.max(), " reload frac", (reload_ > 0).mean(), " level partitions p50/p90/max", np.percentile(nitb, [50, 90, 100]))
slow = tot > np.percentile(tot, 95)
print("slowest 5%: outer", outer[slow].mean(), "inner", inner[slow].mean(), "reload", (reload_[slow] > 0).mean(), "nitb", nitb[slow].mean(), "ncoll", s[slow, 13].mean())
print("corr(total, nitb)", np.corrcoef(tot, nitb)[0, 1], "corr(total, ncoll)", np.corrcoef(tot, s[:, 13])[0, 1])
print(ctx.last_route_info())
import time
t0=time.perf_counter()
for _ in range(20):
    ctx.route_dev(Q, codes.data_ptr(), -1, B, B, sel.data_ptr(), 0, cnt.data_ptr(), 0, 0)
ctx.sync()
print('route kernel avg ms', (time.perf_counter()-t0)/20*1e3)

