"""Dev tool: encode kernel, exact fp64 VALU path vs MFMA fp32 GEMM + exact re-check."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pkg = g.load_package()
d, T, D, m, lam = 128, 16, 1, 16, 2
NQS = (1024, 262144)
if os.environ.get("SHAPE") == "cfg3": d, T, m, NQS = 960, 16, 16, (512, 4096)          # BASELINE config #3: GIST-shaped, 512 queries per GPU
if os.environ.get("SHAPE") == "cfg4": d, T, m, NQS = 768, 32, 32, (1024, 8192)         # config #4: 32 tables x 64 bits, 1 024 queries per GPU
rng = np.random.default_rng(1)
S = rng.standard_normal((1000, d))
ctx = pkg.FspannContext(pkg.PaperRuntimeConfig(tables=T, divisions=D, m=m, lambda_=lam, dim=d), 0)
ctx.registry_initialize(S)
F32 = pkg._native.F32
stream = torch.cuda.ExternalStream(ctx.stream)
for nq in NQS:
    q = torch.randn((nq, d), device="cuda")
    codes = torch.zeros((nq, T * D, 1), dtype=torch.int64, device="cuda")
    bad = torch.zeros(nq, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    for mode, name in ((1, "exact fp64 VALU"), (2, "MFMA f32 + re-check")):
        ctx.set_encode_mode(mode)
        for _ in range(3):
            ctx.encode_dev(nq, q.data_ptr(), F32, codes.data_ptr(), 0, bad.data_ptr())
        ctx.sync()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(20):
            ctx.encode_dev(nq, q.data_ptr(), F32, codes.data_ptr(), 0, bad.data_ptr())
        e1.record(stream)
        ctx.sync()
        ms = e0.elapsed_time(e1) / 20
        flops = 2.0 * nq * d * T * D * m
        print(f"nq={nq:7d} {name:22s} {ms*1e3:9.1f} us  {flops/ms/1e9:8.2f} TFLOP/s  rechecked={ctx.last_encode_rechecked()}")
