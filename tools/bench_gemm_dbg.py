import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import miphei_vit_amd.ops as ops
for (m, n, k) in [(8192, 8192, 8192), (5264, 4608, 1536)]:
    a = torch.randn(m, k, device="cuda").bfloat16(); b = torch.randn(n, k, device="cuda").bfloat16()
    c = torch.empty(m, n, device="cuda", dtype=torch.bfloat16)
    for name, fl in [("full", 0), ("no-dma", 0x100), ("no-compute", 0x200), ("neither", 0x300), ("neither-noepi", 0xB00), ("noloop", 0x1000), ("noloop-noepi", 0x1800)]:
        for _ in range(3): ops.gemm(a, b, c, flags=fl)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): ops.gemm(a, b, c, flags=fl)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        print(f"M={m} N={n} K={k} {name:10s} {ms*1e3:8.1f} us  ({2*m*n*k/ms/1e9:7.1f} TF/s-equivalent)", flush=True)
