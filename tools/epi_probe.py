"""GEMM time vs K at the qkv shape (fixed cost per tile vs per-K-step cost), with and without the epilogue (debug flag)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import miphei_vit_amd.ops as ops
M = 16 * 329
def timeit(fn, it=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e3
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4608
for k in (64, 128, 256, 512, 1024, 1536, 3072):
    a = torch.randn(M, k, device="cuda").bfloat16(); b = torch.randn(n, k, device="cuda").bfloat16()
    c = torch.empty(M, n, device="cuda", dtype=torch.bfloat16)
    t0 = timeit(lambda: ops.gemm(a, b, c)); t1 = timeit(lambda: ops.gemm(a, b, c, flags=0x800))
    t2 = timeit(lambda: torch.matmul(a, b.t(), out=c))
    print(f"N={n} K={k:5d} full {t0:7.1f} us  no-epilogue {t1:7.1f} us  hipBLASLt {t2:7.1f}", flush=True)
