"""Dev experiment: what slows Route beside the scan?  Three contexts run Route only (back to back, as tools/parts_overlap.py),
alone and while another stream keeps HBM busy with a plain device-to-device copy (no LDS, a handful of registers: whatever it
costs Route is the memory system, not CU occupancy).  usage: python tools/route_vs_stream.py"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pkg = g.load_package()
n, d, T, m, B, Q, k = 1_000_000, 128, 16, 16, 256, 1024, 10
rng = np.random.default_rng(1)
X = rng.standard_normal((n, d), dtype=np.float32)
F32 = pkg._native.F32
dev = torch.device("cuda", 0)
cfg = pkg.PaperRuntimeConfig(tables=T, divisions=1, m=m, lambda_=2, dim=d, refinement_limit=B)
ctx0 = pkg.FspannContext(cfg, 0)
ctx0.registry_initialize(X[:1000].astype(np.float64)); ctx0.set_id_meta(n); ctx0.build_index(X)
ctxs = [ctx0, ctx0.clone(), ctx0.clone()]
qs = torch.from_numpy(rng.standard_normal((Q, d), dtype=np.float32)).to(dev)
bufs = []
for c_ in ctxs:
    b = dict(codes=torch.zeros((Q, T, 1), dtype=torch.int64, device=dev), bad=torch.zeros(Q, dtype=torch.int32, device=dev),
             sel=torch.zeros((Q, B), dtype=torch.int32, device=dev), cnt=torch.zeros(Q, dtype=torch.int32, device=dev))
    c_.encode_dev(Q, qs.data_ptr(), F32, b["codes"].data_ptr(), 0, b["bad"].data_ptr()); c_.sync()
    bufs.append(b)
src = torch.empty(1 << 30, dtype=torch.uint8, device=dev); dst = torch.empty_like(src)
small = torch.randn(1 << 22, device=dev)
side = torch.cuda.Stream()


def route_steps(steps, busy):
    torch.cuda.synchronize()
    stop = False
    t0 = time.perf_counter()
    for i in range(steps):
        c_, b = ctxs[i % 3], bufs[i % 3]
        c_.route_dev(Q, b["codes"].data_ptr(), -1, B, B, b["sel"].data_ptr(), 0, b["cnt"].data_ptr(), 0, 0)
        if busy == "copy" and i % 24 == 0:
            with torch.cuda.stream(side):
                dst.copy_(src, non_blocking=True)                      # 1 GiB read + 1 GiB write: ~330 us of saturated HBM
        if busy == "valu" and i % 3 == 0:
            with torch.cuda.stream(side):
                for _ in range(4):
                    small.sin_()                                       # 16 MB, cache resident: vector ALU work, little memory
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e6


for busy in ("none", "copy", "valu", "none"):
    route_steps(60, busy)
    print("Route only, 3 contexts, beside %-5s: %.1f us per launch" % (busy, route_steps(600, busy)), flush=True)
