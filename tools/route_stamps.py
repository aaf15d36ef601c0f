"""Debug tool (FSPANN_BUILD_DEBUG=1 build): phase timeline of the bounded select (route_select_lazy_kernel), one row of stamps per
workgroup = query.  Prints the median / p90 time of every phase over the workgroups, and when workgroups start and end."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g  # noqa: E402

pkg = g.load_package()
L = pkg._native.lib()
L.fspann_debug_route_stamps.argtypes = [C.c_void_p, C.c_void_p]
dev = torch.device("cuda", 0)
n, d, T, m, B, Q = 1_000_000, 128, 16, 16, 256, 1024
rng = np.random.default_rng(1)
X = rng.standard_normal((n, d), dtype=np.float32)
Qs = torch.from_numpy(rng.standard_normal((4, Q, d), dtype=np.float32)).to(dev)
cfg = pkg.PaperRuntimeConfig(tables=T, divisions=1, m=m, lambda_=2, dim=d, refinement_limit=B)
ctx = pkg.FspannContext(cfg, 0)
ctx.registry_initialize(X[:1000].astype(np.float64))
ctx.set_id_meta(n)
ctx.build_index(X)
F32 = pkg._native.F32
codes = torch.zeros((Q, T, 1), dtype=torch.int64, device=dev)
sel = torch.zeros((Q, B), dtype=torch.int32, device=dev)
cnt = torch.zeros(Q, dtype=torch.int32, device=dev)
dbg = torch.zeros((Q, 16), dtype=torch.int64, device=dev)
probes = int(sys.argv[1]) if len(sys.argv) > 1 else -1
for b in range(3):
    ctx.encode_dev(Q, Qs[b].data_ptr(), F32, codes.data_ptr(), 0, 0)
    ctx.route_dev(Q, codes.data_ptr(), probes, B, B, sel.data_ptr(), 0, cnt.data_ptr(), 0, 0)
ctx.encode_dev(Q, Qs[3].data_ptr(), F32, codes.data_ptr(), 0, 0)
ctx.sync()
L.fspann_debug_route_stamps(ctx.handle, dbg.data_ptr())
ctx.route_dev(Q, codes.data_ptr(), probes, B, B, sel.data_ptr(), 0, cnt.data_ptr(), 0, 0)
ctx.sync()
L.fspann_debug_route_stamps(ctx.handle, None)
a = dbg.cpu().numpy().astype(np.float64)
TICK = 0.01  # wall_clock64: 100 MHz -> 10 ns per tick
t0 = a[:, 0].min()
names = {0: "start", 1: "probe done", 2: "probes ordered", 8: "whole levels in", 10: "histogram", 11: "cut", 12: "crossing level in", 3: "walk done",
         4: "keys staged", 5: "ranked + classified", 6: "results issued"}
order = [0, 1, 2, 8, 10, 11, 12, 3, 4, 5, 6]
print("workgroup start  us: min %.2f med %.2f max %.2f" % tuple((np.percentile(a[:, 0], p) - t0) * TICK for p in (0, 50, 100)))
print("workgroup end    us: min %.2f med %.2f max %.2f" % tuple((np.percentile(a[:, 6], p) - t0) * TICK for p in (0, 50, 100)))
prev = 0
for s in order[1:]:
    ok = (a[:, s] > 0) & (a[:, prev] > 0)
    dt = (a[ok, s] - a[ok, prev]) * TICK
    if ok.sum():
        print("%-22s <- %-18s n=%4d  med %.2f  p90 %.2f  max %.2f us" % (names[s], names[prev], ok.sum(), np.median(dt), np.percentile(dt, 90), dt.max()))
    prev = s
tot = (a[:, 6] - a[:, 0]) * TICK
print("workgroup total  us: med %.2f p90 %.2f max %.2f" % (np.median(tot), np.percentile(tot, 90), tot.max()))
x = a[:, 14].astype(np.int64)
print("outer trips med %d, inner med %d, reloads total %d, max nitB %d; ncoll med %d; nsel med %d max %d" % (
    np.median(x & 255), np.median((x >> 8) & 255), ((x >> 16) & 255).sum(), (x >> 24).max(), np.median(a[:, 13]), np.median(a[:, 15]), a[:, 15].max()))
