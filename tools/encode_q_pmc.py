"""Dev tool (run under rocprofv3 --pmc): the QUERY-side encode of one batch of 1 024 queries in both modes — exact fp64 VALU kernel
(mode 1) and MFMA fp32 pre-filter + exact re-check (mode 2) — so the MFMA busy counters of the two can be read side by side
(VERDICT r03, next #6).  usage: rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CYCLES SQ_WAVES -- python3 tools/encode_q_pmc.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g  # noqa: E402

pkg = g.load_package()
n, d, T, m, Q = 100_000, 128, 16, 16, 1024
rng = np.random.default_rng(1)
X = rng.standard_normal((n, d), dtype=np.float32)
ctx = pkg.FspannContext(pkg.PaperRuntimeConfig(tables=T, divisions=1, m=m, lambda_=2, dim=d, refinement_limit=256), 0)
ctx.registry_initialize(X[:1000].astype(np.float64))
dev = torch.device("cuda", 0)
q = torch.from_numpy(rng.standard_normal((8, Q, d), dtype=np.float32)).to(dev)
codes = torch.zeros((Q, T, 1), dtype=torch.int64, device=dev)
bad = torch.zeros(Q, dtype=torch.int32, device=dev)
for mode in (1, 2):
    ctx.set_encode_mode(mode)
    for i in range(24):
        ctx.encode_dev(Q, q[i % 8].data_ptr(), pkg._native.F32, codes.data_ptr(), 0, bad.data_ptr())
    ctx.sync()
    print("mode", mode, "rechecked pairs of the last call", ctx.last_encode_rechecked() if mode == 2 else None, flush=True)
ctx.close()
