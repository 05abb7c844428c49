import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import miphei_vit_amd.ops as ops
M = 16 * 329
for (m, n, k) in [(M, 4608, 1536), (M, 1536, 8192), (8192, 8192, 8192)]:
    a = torch.randn(m, k, device="cuda").bfloat16(); b = torch.randn(n, k, device="cuda").bfloat16()
    c = torch.empty(m, n, device="cuda", dtype=torch.bfloat16)
    for name, fl in [("base", 0), ("setprio", 0x2000), ("base", 0), ("setprio", 0x2000)]:
        for _ in range(3): ops.gemm(a, b, c, flags=fl)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): ops.gemm(a, b, c, flags=fl)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        print(f"M={m} N={n} K={k} {name:8s} {ms*1e3:8.1f} us  {2*m*n*k/ms/1e9:7.1f} TF/s", flush=True)
