"""Measurement only: the library GEMM and hipBLASLt on random vs all-zero operands (same instruction stream; the difference is
the clock the chip sustains under the lower switching power)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import miphei_vit_amd.ops as ops
def timeit(fn, it=20):
    for _ in range(5): fn()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / it
for name, m, n, k in [("sq8k", 8192, 8192, 8192), ("dfc1", 5264, 1536, 8192), ("qkv", 5264, 4608, 1536)]:
    for kind in ("random", "zeros"):
        a = (torch.randn(m, k, device="cuda") if kind == "random" else torch.zeros(m, k, device="cuda")).bfloat16()
        b = (torch.randn(n, k, device="cuda") if kind == "random" else torch.zeros(n, k, device="cuda")).bfloat16()
        c = torch.empty(m, n, device="cuda", dtype=torch.bfloat16)
        bt = b.t()
        t1 = timeit(lambda: ops.gemm(a, b, c)); t2 = timeit(lambda: torch.matmul(a, bt, out=c))
        fl = 2 * m * n * k
        print(f"{name} {kind:6s}: ours {t1*1e3:7.1f} us {fl/t1/1e9:7.1f} TF/s | hipBLASLt {t2*1e3:7.1f} us {fl/t2/1e9:7.1f} TF/s", flush=True)
