"""How many tuples does a score-ordered (lazy) select have to look at?  (unique ids with score <= the limit-th score)"""
import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as graft
pkg = graft.load_package()
n, d, T, m, lam, B, Q = 1000000, 128, 16, 16, 2, 256, 1024
rng = np.random.default_rng(1)
X = rng.standard_normal((n, d), dtype=np.float32)
Qh = np.random.default_rng(1001).standard_normal((Q, d), dtype=np.float32)
cfg = pkg.PaperRuntimeConfig(tables=T, divisions=1, m=m, lambda_=lam, dim=d, seed=13, refinement_limit=B)
with pkg.FspannContext(cfg, 0) as c:
    c.registry_initialize(X[:1000].astype(np.float64)); c.set_id_meta(n); c.build_index(X)
    enc = c.encode(Qh); codes = enc["codes"] if isinstance(enc, dict) else enc[0] if isinstance(enc, tuple) else enc
    r = c.route(codes)
    need = []
    for q in range(Q):
        cnt = r["count"][q]; sc = r["score"][q][:cnt]
        s = sc[min(B, cnt) - 1]
        need.append(int((sc <= s).sum()))
    need = np.array(need)
    print("tuples needed: mean %.0f p50 %d p90 %d p99 %d max %d (of %d unique)" % (need.mean(), *np.percentile(need, [50, 90, 99]).astype(int), need.max(), r["count"].mean()))
    lv = [len(np.unique(r["score"][q][:r["count"][q]])) for q in range(64)]
    print("distinct score levels/query:", np.mean(lv))
