"""Dev experiment (profiles/notes/r03_overlap_experiments.txt): a stand-in for a PERSISTENT scan.  One refinement-scan launch over
24 batches at once (24 576 units, FSPANN_REFINE_STREAM workgroups per CU: they stay resident for the whole run) while three other
contexts launch Route batch after batch — do the two kinds share the CUs when the scan's workgroups never leave?
usage: FSPANN_REFINE_STREAM=2 python tools/persistent_proxy.py"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pkg = g.load_package()
n, d, T, m, B, Q, k, NB = 1_000_000, 128, 16, 16, 256, 1024, 10, 24
RQ = 4 * Q          # queries per Route launch: four batches, so that the host's launch rate does not limit the Route side
rng = np.random.default_rng(1)
X = rng.standard_normal((n, d), dtype=np.float32)
F32 = pkg._native.F32
dev = torch.device("cuda", 0)
cfg = pkg.PaperRuntimeConfig(tables=T, divisions=1, m=m, lambda_=2, dim=d, refinement_limit=B)
ctx0 = pkg.FspannContext(cfg, 0)
ctx0.registry_initialize(X[:1000].astype(np.float64)); ctx0.set_id_meta(n); ctx0.build_index(X)
rctx = [ctx0.clone(), ctx0.clone(), ctx0.clone()]
fctx = ctx0.clone()
cand = torch.randn((NB * Q, B, d), dtype=torch.float32, device=dev)
qs = torch.from_numpy(rng.standard_normal((NB * Q, d), dtype=np.float32)).to(dev)
ids = torch.arange(NB * Q * B, dtype=torch.int32, device=dev).reshape(NB * Q, B)
full = torch.full((NB * Q,), B, dtype=torch.int32, device=dev)
oi = torch.zeros((NB * Q, k), dtype=torch.int32, device=dev); od = torch.zeros((NB * Q, k), dtype=torch.float64, device=dev)
oc = torch.zeros(NB * Q, dtype=torch.int32, device=dev)
rb = []
for c_ in rctx:
    b = dict(codes=torch.zeros((RQ, T, 1), dtype=torch.int64, device=dev), bad=torch.zeros(RQ, dtype=torch.int32, device=dev),
             sel=torch.zeros((RQ, B), dtype=torch.int32, device=dev), cnt=torch.zeros(RQ, dtype=torch.int32, device=dev))
    c_.encode_dev(RQ, qs.data_ptr(), F32, b["codes"].data_ptr(), 0, b["bad"].data_ptr()); c_.sync()
    rb.append(b)


def big_scan():
    fctx.refine_dev(NB * Q, qs.data_ptr(), F32, cand.data_ptr(), F32, B, ids.data_ptr(), full.data_ptr(), k, oi.data_ptr(), od.data_ptr(), oc.data_ptr(), 0)


def routes(nl):
    for i in range(nl):
        c_, b = rctx[i % 3], rb[i % 3]
        c_.route_dev(RQ, b["codes"].data_ptr(), -1, B, B, b["sel"].data_ptr(), 0, b["cnt"].data_ptr(), 0, 0)


def timed(fn):
    torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); return (time.perf_counter() - t0) * 1e6


big_scan(); routes(30); torch.cuda.synchronize()
tF = min(timed(big_scan) for _ in range(3))
tR = min(timed(lambda: routes(NB // 4)) for _ in range(3))
tB = min(timed(lambda: (big_scan(), routes(NB // 4))) for _ in range(3))
print("scan wgs/CU %s: scan of %d batches alone %.0f us (%.1f per batch) | %d Route launches alone %.0f us (%.1f each) | both together %.0f us "
      "(sum would be %.0f, perfect overlap %.0f)" % (os.environ.get("FSPANN_REFINE_STREAM", "default"), NB, tF, tF / NB, NB, tR, tR / NB, tB, tF + tR, max(tF, tR)), flush=True)
