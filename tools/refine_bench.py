"""Dev tool: refine kernel alone, hot (same buffer) vs cold (rotating > Infinity Cache) inputs."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pkg = g.load_package()
if os.environ.get('AB_LIB'): pkg._native._SO = os.environ['AB_LIB']   # a variant build (dev experiments)
import os
Q, B, d, k = 1024, 256, 128, int(os.environ.get('KK', '10'))
ctx = pkg.FspannContext(pkg.PaperRuntimeConfig(tables=1, divisions=1, m=8, lambda_=2, dim=d, refinement_limit=B), 0)
NB = int(os.environ.get("NBUF", "8"))
cands = [torch.randn((Q, B, d), device="cuda") for _ in range(NB)]
q = torch.randn((Q, d), device="cuda")
ids = torch.arange(Q * B, dtype=torch.int32, device="cuda").reshape(Q, B)
cnt = torch.full((Q,), B, dtype=torch.int32, device="cuda")
oi = torch.zeros((Q, k), dtype=torch.int32, device="cuda"); od = torch.zeros((Q, k), dtype=torch.float64, device="cuda")
oc = torch.zeros(Q, dtype=torch.int32, device="cuda")
torch.cuda.synchronize()
stream = torch.cuda.ExternalStream(ctx.stream)
F32 = pkg._native.F32
def run(buf):
    ctx.refine_dev(Q, q.data_ptr(), F32, buf.data_ptr(), F32, B, ids.data_ptr(), cnt.data_ptr(), k, oi.data_ptr(), od.data_ptr(), oc.data_ptr(), 0)
for name, seq in (("hot", [0] * 40), ("cold", [i % NB for i in range(40)])):
    for b in seq[:8]: run(cands[b])
    ctx.sync()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in seq]
    for (e0, e1), b in zip(evs, seq):
        e0.record(stream); run(cands[b]); e1.record(stream)
    ctx.sync()
    ts = np.array([e0.elapsed_time(e1) for e0, e1 in evs])
    byt = Q * (B * d * 4 + d * 4 + k * 8)
    print(f"DC={os.environ.get('FSPANN_REFINE_DC','default')} {name}: median {np.median(ts)*1e3:.1f} us  min {ts.min()*1e3:.1f} us  -> {byt/np.median(ts)/1e6:.0f} GB/s")
