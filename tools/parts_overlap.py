"""Dev experiment: N contexts (N HIP streams) taking turns, each step running only the stages named in PARTS (E encode, R Route,
F Refine on dense blocks): how well kernels of the same kind from different streams overlap, and what each kind costs the others.
usage: NCTX=3 PARTS=R python tools/parts_overlap.py"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pkg = g.load_package()
if os.environ.get('AB_LIB'): pkg._native._SO = os.path.abspath(os.environ['AB_LIB'])   # a variant build (dev experiments)
n, d, T, m, B, Q, k = int(os.environ.get("N", "1000000")), 128, 16, 16, 256, 1024, 10   # N: base vectors (working set of Route)
rng = np.random.default_rng(1)
X = rng.standard_normal((n, d), dtype=np.float32)
F32 = pkg._native.F32
dev = torch.device("cuda", 0)
cfg = pkg.PaperRuntimeConfig(tables=T, divisions=1, m=m, lambda_=2, dim=d, refinement_limit=B)
ctx0 = pkg.FspannContext(cfg, 0)
ctx0.registry_initialize(X[:1000].astype(np.float64)); ctx0.set_id_meta(n); ctx0.build_index(X)
NB = 24
cand = torch.randn((NB, Q, B, d), dtype=torch.float32, device=dev)
qs = torch.from_numpy(rng.standard_normal((NB, Q, d), dtype=np.float32)).to(dev)
ids = torch.arange(Q * B, dtype=torch.int32, device=dev).reshape(Q, B)
full = torch.full((Q,), B, dtype=torch.int32, device=dev)


def mk(nctx):
    ctxs = [ctx0]
    for _ in range(nctx - 1):
        if os.environ.get("SHARE", "1") == "1":          # clones read context 0's index in place (SHARE=0: a copy per context)
            ctxs.append(ctx0.clone())
            continue
        c2 = pkg.FspannContext(cfg, 0)
        c2.set_gfunctions(*ctx0.get_gfunctions()); c2.set_id_meta(n)
        for td in range(T):
            c2.set_index(td, **ctx0.get_index(td))
        c2.finalize()
        ctxs.append(c2)
    bufs = [dict(codes=torch.zeros((Q, T, 1), dtype=torch.int64, device=dev), bad=torch.zeros(Q, dtype=torch.int32, device=dev),
                 sel=torch.zeros((Q, B), dtype=torch.int32, device=dev), cnt=torch.zeros(Q, dtype=torch.int32, device=dev),
                 oi=torch.zeros((Q, k), dtype=torch.int32, device=dev), od=torch.zeros((Q, k), dtype=torch.float64, device=dev),
                 oc=torch.zeros(Q, dtype=torch.int32, device=dev)) for _ in ctxs]
    return ctxs, bufs


def run(ctxs, bufs, parts_all, steps):
    parts = parts_all
    for c_, b in zip(ctxs, bufs):      # valid codes for a route-only run
        c_.encode_dev(Q, qs[0].data_ptr(), F32, b["codes"].data_ptr(), 0, b["bad"].data_ptr())
        c_.sync()

    split = parts.split("|") if "|" in parts else None     # "R|F": context 0 runs only R steps, context 1 only F steps, ...

    def step(i):
        c_, b = ctxs[i % len(ctxs)], bufs[i % len(ctxs)]
        parts = split[(i % len(ctxs)) % len(split)] if split else parts_all
        if "E" in parts:
            c_.encode_dev(Q, qs[i % NB].data_ptr(), F32, b["codes"].data_ptr(), 0, b["bad"].data_ptr())
        if "R" in parts:
            c_.route_dev(Q, b["codes"].data_ptr(), -1, B, B, b["sel"].data_ptr(), 0, b["cnt"].data_ptr(), 0, 0)
        if "F" in parts:
            c_.refine_dev(Q, qs[i % NB].data_ptr(), F32, cand[i % NB].data_ptr(), F32, B, ids.data_ptr(), full.data_ptr(), k, b["oi"].data_ptr(),
                          b["od"].data_ptr(), b["oc"].data_ptr(), 0)
    for i in range(30):
        step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        step(i)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e6


for nctx in [int(x) for x in os.environ.get("NCTXS", "1,2,3,4").split(",")]:
    ctxs, bufs = mk(nctx)
    line = []
    for parts in os.environ.get("PARTS", "E,R,F,ER,RF,ERF").split(","):
        line.append("%s %.1f" % (parts, run(ctxs, bufs, parts, 300)))
    print("contexts %d: us per step  " % nctx + "  ".join(line), flush=True)
    for c_ in ctxs[1:]:
        c_.close() if hasattr(c_, "close") else None
