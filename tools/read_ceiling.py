"""Dev tool: practical read-bandwidth ceiling for a 134 MB buffer (torch.sum), hot vs rotating buffers."""
import torch, numpy as np
N = 1024 * 256 * 128
bufs = [torch.randn(N, device="cuda") for _ in range(8)]
for name, seq in (("hot", [0] * 30), ("cold", [i % 8 for i in range(30)])):
    for b in seq[:5]: bufs[b].sum()
    torch.cuda.synchronize()
    ts = []
    for b in seq:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); bufs[b].sum(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    t = np.median(ts)
    print(f"torch.sum 134MB {name}: {t*1e3:.1f} us -> {N*4/t/1e6:.0f} GB/s")
x = bufs[0]; y = torch.empty_like(x)
for _ in range(5): y.copy_(x)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): y.copy_(x)
e1.record(); torch.cuda.synchronize()
t = e0.elapsed_time(e1) / 20
print(f"copy 134MB->134MB: {t*1e3:.1f} us -> {2*N*4/t/1e6:.0f} GB/s (read+write)")
