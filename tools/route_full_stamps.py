"""Debug tool (FSPANN_BUILD_DEBUG=1 build): phase timeline of the FULL select (route_select_kernel) at one of the reference's shipped
profiles (config_sift1m.json), one row of stamps per workgroup (its first query).  usage: python tools/route_full_stamps.py P10|P4"""
import ctypes as C
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g  # noqa: E402

pkg = g.load_package()
if os.environ.get('AB_LIB'):
    pkg._native._SO = os.path.abspath(os.environ['AB_LIB'])   # a variant build (debug stamps)
L = pkg._native.lib()
HAVE_STAMPS = hasattr(L, "fspann_debug_route_stamps")      # only in FSPANN_BUILD_DEBUG=1 builds; a release build still prints the timing
if HAVE_STAMPS:
    L.fspann_debug_route_stamps.argtypes = [C.c_void_p, C.c_void_p]
dev = torch.device("cuda", 0)
prof = sys.argv[1] if len(sys.argv) > 1 else "P10"
T, D, m, P, B, HC = dict(P10=(7, 8, 26, 10, 22000, 28000), P4=(5, 8, 20, 4, 8000, 10000), P6=(6, 8, 24, 6, 16000, 20000))[prof]
n, d, Q = 1_000_000, 128, 1024
rng = np.random.default_rng(1)
if os.environ.get("DATA", "").startswith("siftlike"):          # the bench's SIFT-like generator (shipped-profile children)
    import importlib.util
    _sp = importlib.util.spec_from_file_location("fspann_bench", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"))
    _b = importlib.util.module_from_spec(_sp); _sp.loader.exec_module(_b)
    X, _Q, _ = _b.make_data(os.environ["DATA"], n, d, 1, Q, 13, 0)
    Qh = np.ascontiguousarray(_Q[0])
else:
    Cc = rng.standard_normal((4096, d), dtype=np.float32)
    X = Cc[rng.integers(0, 4096, n)] + np.float32(0.15) * rng.standard_normal((n, d), dtype=np.float32)
    Qh = Cc[rng.integers(0, 4096, Q)] + np.float32(0.15) * rng.standard_normal((Q, d), dtype=np.float32)
cfg = pkg.PaperRuntimeConfig(tables=T, divisions=D, m=m, lambda_=2, dim=d, refinement_limit=B, max_global_candidates=HC, probe_override=P)
ctx = pkg.FspannContext(cfg, 0)
ctx.registry_initialize(X[:1000].astype(np.float64))
ctx.set_id_meta(n)
ctx.build_index(X)
F32 = pkg._native.F32
TD = T * D
qd = torch.from_numpy(Qh).to(dev)
codes = torch.zeros((Q, TD, 1), dtype=torch.int64, device=dev)
sel = torch.zeros((Q, B), dtype=torch.int32, device=dev)
cnt = torch.zeros(Q, dtype=torch.int32, device=dev)
kept = torch.zeros(Q, dtype=torch.int32, device=dev)
ctx.encode_dev(Q, qd.data_ptr(), F32, codes.data_ptr(), 0, 0)
for _ in range(2):
    ctx.route_dev(Q, codes.data_ptr(), -1, B, B, sel.data_ptr(), 0, cnt.data_ptr(), kept.data_ptr(), 0)
ctx.sync()
t0 = time.perf_counter()
for _ in range(5):
    ctx.route_dev(Q, codes.data_ptr(), -1, B, B, sel.data_ptr(), 0, cnt.data_ptr(), kept.data_ptr(), 0)
ctx.sync()
print("%s: route_dev %.1f us per 1024 queries; kept med %d max %d" % (prof, (time.perf_counter() - t0) / 5 * 1e6, kept.float().median().item(), kept.max().item()))
if not HAVE_STAMPS:
    sys.exit(0)
grid = 1024
dbg = torch.zeros((grid, 16), dtype=torch.int64, device=dev)
L.fspann_debug_route_stamps(ctx.handle, dbg.data_ptr())
ctx.route_dev(Q, codes.data_ptr(), -1, B, B, sel.data_ptr(), 0, cnt.data_ptr(), kept.data_ptr(), 0)
ctx.sync()
L.fspann_debug_route_stamps(ctx.handle, None)
a = dbg.cpu().numpy().astype(np.float64)
a = a[a[:, 0] > 0]
TICK = 0.01
names = {0: "start", 1: "reset + probe list", 8: "ids staged", 12: "slice 0: table built / repeats staged", 13: "slice 0: repeats marked / repeats walked", 2: "hash built", 9: "repeats dropped, cap", 3: "repeat scores", 11: "level cuts", 4: "compacted", 7: "groups counted", 10: "sub-keys scattered", 14: "groups sorted by waves",
         5: "ordered + written", 6: "treeify check, done"}
order = [0, 1, 8, 12, 13, 2, 9, 3, 11, 4, 7, 10, 14, 5, 6]
print("workgroups with stamps:", len(a))
prev = 0
for s in order[1:]:
    ok = (a[:, s] > 0) & (a[:, prev] > 0)
    if ok.sum():
        dt = (a[ok, s] - a[ok, prev]) * TICK
        print("%-24s <- %-22s n=%4d  med %8.1f  p90 %8.1f  max %8.1f us" % (names[s], names[prev], ok.sum(), np.median(dt), np.percentile(dt, 90), dt.max()))
        prev = s
tot = (a[:, 6] - a[:, 0]) * TICK
v15 = a[:, 15].astype(np.int64)
print("first query of a workgroup, total us: med %.1f p90 %.1f max %.1f; nsel med %d; groups med %d, levels med %d, largest level med %d" %
      (np.median(tot), np.percentile(tot, 90), tot.max(), np.median(v15 & 0xFFFFF), np.median((v15 >> 20) & 0xFFF), np.median((v15 >> 32) & 0xFF), np.median(v15 >> 40)))
