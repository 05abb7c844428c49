import sys, os, time
import torch
print("cpu_count", os.cpu_count())
a = torch.randn(658, 1536); w = torch.randn(8192, 1536)
for nt in (8, 16, 32, 64, 128):
    torch.set_num_threads(nt)
    for _ in range(2): (a @ w.t())
    t0 = time.perf_counter()
    for _ in range(10): (a @ w.t())
    dt = (time.perf_counter() - t0) / 10
    print(nt, f"{dt*1e3:.2f} ms", f"{2*658*1536*8192/dt/1e9:.1f} GF/s", flush=True)
