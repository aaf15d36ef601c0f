"""Dev tool (FSPANN_DEBUG_STAMPS build): time stamps INSIDE the probe of the bounded select — directory entry, every search round,
window, replay — for lane group 0 of each workgroup's first query.  usage: AB_LIB=<debug lib> python tools/route_probe_stamps.py"""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pkg = g.load_package()
if os.environ.get('AB_LIB'): pkg._native._SO = os.path.abspath(os.environ['AB_LIB'])
n, d, T, D, m, lam, B, Q = 1_000_000, 128, 16, 1, 16, 2, 256, 1024
rng = np.random.default_rng(1)
X = rng.standard_normal((n, d), dtype=np.float32)
Qh = np.random.default_rng(2).standard_normal((Q, d), dtype=np.float32)
ctx = pkg.FspannContext(pkg.PaperRuntimeConfig(tables=T, divisions=D, m=m, lambda_=lam, dim=d, refinement_limit=B), 0)
ctx.registry_initialize(X[:1000].astype(np.float64)); ctx.set_id_meta(n); ctx.build_index(X)
codes = torch.from_numpy(ctx.encode(Qh).view(np.int64)).cuda()
sel = torch.zeros((Q, B), dtype=torch.int32, device='cuda'); cnt = torch.zeros(Q, dtype=torch.int32, device='cuda')
dbg = torch.zeros((2048, 16), dtype=torch.int64, device='cuda')
L = pkg._native.lib()
L.fspann_debug_route_stamps.argtypes = [C.c_void_p, C.c_void_p]
for it in range(3):
    dbg.zero_()
    L.fspann_debug_route_stamps(ctx.handle, dbg.data_ptr())
    ctx.route_dev(Q, codes.data_ptr(), -1, B, B, sel.data_ptr(), 0, cnt.data_ptr(), 0, 0); ctx.sync()
s = dbg.cpu().numpy().astype(np.float64)
p = s[1024:2048]
p = p[p[:, 15] >= 5]
print('workgroups', len(p), ' stamps per probe: min/med/max', p[:, 15].min(), np.median(p[:, 15]), p[:, 15].max())
for nst in sorted(set(p[:, 15].astype(int))):
    q = p[p[:, 15] == nst]
    names = ['entry', 'code+table', 'directory'] + ['round %d' % (i + 1) for i in range(nst - 5)] + ['window', 'replay']
    line = []
    for i in range(1, nst):
        dt = (q[:, i] - q[:, i - 1]) / 100.0
        line.append('%s %.2f' % (names[i], np.median(dt)))
    print('%4d probes with %d search rounds: ' % (len(q), nst - 5) + ' | '.join(line) + ' | total %.2f us' % np.median((q[:, nst - 1] - q[:, 0]) / 100.0))
