"""Debug tool (FSPANN_BUILD_DEBUG=1 build): timeline of one refine_stream_kernel launch over a dense [Q][B][d] block from cold HBM —
when workgroups start, when their first tiles arrive, when they reach the top-K epilogue and when they end."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g  # noqa: E402

pkg = g.load_package()
if os.environ.get('AB_LIB'): pkg._native._SO = os.environ['AB_LIB']   # a variant build (dev experiments)
L = pkg._native.lib()
L.fspann_debug_route_stamps.argtypes = [C.c_void_p, C.c_void_p]
dev = torch.device("cuda", 0)
d, B, Q, k, NB = 128, 256, int(os.environ.get("Q", 1024)), 10, 12
cfg = pkg.PaperRuntimeConfig(tables=16, divisions=1, m=16, lambda_=2, dim=d, refinement_limit=B)
ctx = pkg.FspannContext(cfg, 0)
F32 = pkg._native.F32
cand = torch.randn((NB, Q, B, d), dtype=torch.float32, device=dev)
q = torch.randn((Q, d), dtype=torch.float32, device=dev)
ids = torch.arange(Q * B, dtype=torch.int32, device=dev).reshape(Q, B)
cnt = torch.full((Q,), B, dtype=torch.int32, device=dev)
oi = torch.zeros((Q, k), dtype=torch.int32, device=dev)
od = torch.zeros((Q, k), dtype=torch.float64, device=dev)
oc = torch.zeros(Q, dtype=torch.int32, device=dev)
sc = torch.zeros(Q, dtype=torch.int32, device=dev)
G = 4 * 256
dbg = torch.zeros((G * 4, 16), dtype=torch.int64, device=dev)   # [workgroup][wave][16]


def run(b):
    ctx.refine_dev(Q, q.data_ptr(), F32, cand[b].data_ptr(), F32, B, ids.data_ptr(), cnt.data_ptr(), k, oi.data_ptr(), od.data_ptr(), oc.data_ptr(), sc.data_ptr())


for b in range(NB - 1):
    run(b)
ctx.sync()
L.fspann_debug_route_stamps(ctx.handle, dbg.data_ptr())
run(NB - 1)
ctx.sync()
L.fspann_debug_route_stamps(ctx.handle, None)
a = dbg.cpu().numpy().astype(np.float64)
a = a[a[:, 0] > 0]
TICK = 0.01
t0 = a[:, 0].min()
names = ["start", "first two tiles requested", "tile 0 consumed", "tile 1 consumed", "all tiles consumed", "keys ready",
         "cut bisected", "barrier 1 passed", "survivors listed", "barrier 2 passed", "ranked", "written", "after emit"]
print("waves with stamps:", len(a))
for i, nm in enumerate(names):
    v = (a[:, i] - t0) * TICK
    print("%-28s min %6.2f  p10 %6.2f  med %6.2f  p90 %6.2f  max %6.2f us" % (nm, v.min(), np.percentile(v, 10), np.median(v), np.percentile(v, 90), v.max()))
for i in range(1, len(names)):
    dt = (a[:, i] - a[:, i - 1]) * TICK
    print("  %-26s <- %-26s med %5.2f  p90 %5.2f  max %5.2f us" % (names[i], names[i - 1], np.median(dt), np.percentile(dt, 90), dt.max()))
# wave skew inside a workgroup: spread of "keys ready" over its four waves
full = dbg.cpu().numpy().astype(np.float64).reshape(-1, 4, 16)
ok = full[:, :, 0].min(axis=1) > 0
kr = full[ok][:, :, 5]
print("keys ready, spread over the 4 waves of a workgroup: med %.2f p90 %.2f max %.2f us" % (
    np.median(kr.max(axis=1) - kr.min(axis=1)) * TICK, np.percentile(kr.max(axis=1) - kr.min(axis=1), 90) * TICK, (kr.max(axis=1) - kr.min(axis=1)).max() * TICK))
last = full[ok][:, :, 12].max(axis=1)
print("workgroup end (last wave after emit): med %.2f p90 %.2f max %.2f us" % ((np.median(last) - t0) * TICK, (np.percentile(last, 90) - t0) * TICK, (last.max() - t0) * TICK))
for x in range(8):
    rows = full[ok][x::8]
    if len(rows):
        print("XCD %d: tiles consumed med %6.2f max %6.2f | end med %6.2f max %6.2f us" % (
            x, (np.median(rows[:, :, 4]) - t0) * TICK, (rows[:, :, 4].max() - t0) * TICK, (np.median(rows[:, :, 12]) - t0) * TICK, (rows[:, :, 12].max() - t0) * TICK))
