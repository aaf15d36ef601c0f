"""Debug tool (FSPANN_BUILD_DEBUG=1 build): timeline of one tick_kernel launch — when the workgroups of each role start and end."""
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
n, d, T, m, B, Q, k = 1_000_000, 128, 16, 16, 256, 1024, 10
rng = np.random.default_rng(1)
X = rng.standard_normal((n, d), dtype=np.float32)
NB = 8
Qs = torch.from_numpy(rng.standard_normal((NB, Q, d), dtype=np.float32)).to(dev)
cfg = pkg.PaperRuntimeConfig(tables=T, divisions=1, m=m, lambda_=2, dim=d, refinement_limit=B)
ctx = pkg.FspannContext(cfg, 0)
ctx.registry_initialize(X[:1000].astype(np.float64))
ctx.set_id_meta(n)
ctx.build_index(X)
ctx.store_set(X)
F32 = pkg._native.F32
slots = [dict(codes=torch.zeros((Q, T, 1), dtype=torch.int64, device=dev), sel=torch.zeros((Q, B), dtype=torch.int32, device=dev),
              cnt=torch.zeros(Q, dtype=torch.int32, device=dev), hov=torch.zeros(ctx.route_handover_bytes(Q), dtype=torch.uint8, device=dev)) for _ in range(3)]
oi = torch.zeros((Q, k), dtype=torch.int32, device=dev)
od = torch.zeros((Q, k), dtype=torch.float64, device=dev)
oc = torch.zeros(Q, dtype=torch.int32, device=dev)
mode = sys.argv[1] if len(sys.argv) > 1 else "store"
cand = torch.randn((NB, Q, B, d), dtype=torch.float32, device=dev) if mode == "dense" else None
G = 256 + 2 * Q
dbg = torch.zeros((G, 4), dtype=torch.int64, device=dev)
torch.cuda.synchronize()


PARTS = os.environ.get("PARTS", "ERF")


def tick(t, parts="ERF"):
    sE, sR, sF = slots[(t + 2) % 3], slots[(t + 1) % 3], slots[t % 3]
    ctx.tick_dev(encode=dict(nq=Q, q=Qs[(t + 2) % NB].data_ptr(), codes=sE["codes"].data_ptr()) if "E" in parts else None,
                 route=dict(nq=Q, codes=sR["codes"].data_ptr(), limit=B, ids=sR["sel"].data_ptr(), count=sR["cnt"].data_ptr(),
                            handover=sR["hov"].data_ptr()) if "R" in parts else None,
                 refine=dict(nq=Q, q=Qs[t % NB].data_ptr(), B=B, ids=sF["sel"].data_ptr(), count=sF["cnt"].data_ptr(), k=k, out_ids=oi.data_ptr(),
                             out_dist=od.data_ptr(), out_count=oc.data_ptr(), codes=sF["codes"].data_ptr(), handover=sF["hov"].data_ptr(),
                             cand=cand[t % NB].data_ptr() if cand is not None else None) if "F" in parts else None)


for t in range(6):
    tick(t)
ctx.sync()
L.fspann_debug_route_stamps(ctx.handle, dbg.data_ptr())
tick(6, PARTS)
ctx.sync()
L.fspann_debug_route_stamps(ctx.handle, None)
a = dbg.cpu().numpy()
t0 = a[:, 2].min()
us = 0.01                                        # wall_clock64: 100 MHz
a = a[a[:, 3] > 0]
t0 = a[:, 2].min()
print("parts %s: launch span %.1f us, fused=%s, front=%s" % (PARTS, (a[:, 3].max() - t0) * us, ctx.last_tick_fused(), os.environ.get("FSPANN_TICK_FRONT", "50")))
for role, name in ((0, "encode"), (1, "route"), (2, "refine")):
    r = a[a[:, 0] == role]
    if len(r) == 0:
        continue
    st, en = (r[:, 2] - t0) * us, (r[:, 3] - t0) * us
    du = en - st
    print("%-7s n=%4d  start p0/p50/p100 = %5.1f %5.1f %5.1f   end p50/p100 = %5.1f %5.1f   duration p10/p50/p90/max = %5.1f %5.1f %5.1f %5.1f us"
          % (name, len(r), st.min(), np.median(st), st.max(), np.median(en), en.max(), np.percentile(du, 10), np.median(du), np.percentile(du, 90), du.max()))
