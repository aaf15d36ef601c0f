"""Debug tool (FSPANN_BUILD_DEBUG=1 build): phase timeline of encode_exact_kernel for one 1024-query batch."""
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
n, d, T, m, Q = 20000, 128, 16, 16, 1024
rng = np.random.default_rng(1)
X = rng.standard_normal((n, d), dtype=np.float32)
cfg = pkg.PaperRuntimeConfig(tables=T, divisions=1, m=m, lambda_=2, dim=d, refinement_limit=256)
ctx = pkg.FspannContext(cfg, 0)
ctx.registry_initialize(X[:1000].astype(np.float64))
F32 = pkg._native.F32
qs = torch.from_numpy(rng.standard_normal((4, Q, d), dtype=np.float32)).to(dev)
codes = torch.zeros((Q, T, 1), dtype=torch.int64, device=dev)
bad = torch.zeros(Q, dtype=torch.int32, device=dev)
dbg = torch.zeros((Q, 16), dtype=torch.int64, device=dev)
for b in range(3):
    ctx.encode_dev(Q, qs[b].data_ptr(), F32, codes.data_ptr(), 0, bad.data_ptr())
ctx.sync()
L.fspann_debug_route_stamps(ctx.handle, dbg.data_ptr())
ctx.encode_dev(Q, qs[3].data_ptr(), F32, codes.data_ptr(), 0, bad.data_ptr())
ctx.sync()
L.fspann_debug_route_stamps(ctx.handle, None)
a = dbg.cpu().numpy().astype(np.float64)
a = a[a[:, 0] > 0]
TICK = 0.01
t0 = a[:, 0].min()
names = ["start", "row check loads issued", "projection loop done", "flags zeroed (barrier)", "hashes in LDS (barrier)", "codes written"]
print("workgroups:", len(a))
for i, nm in enumerate(names):
    v = (a[:, i] - t0) * TICK
    print("%-26s min %6.2f med %6.2f max %6.2f us" % (nm, v.min(), np.median(v), v.max()))
for i in range(1, len(names)):
    dt = (a[:, i] - a[:, i - 1]) * TICK
    print("  %-24s <- %-24s med %5.2f  max %5.2f us" % (names[i], names[i - 1], np.median(dt), dt.max()))
