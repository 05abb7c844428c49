"""Yardstick only: mvit_gemm_bf16 beside the vendor GEMM (torch.matmul -> hipBLASLt) on the encoder's shapes.
Not part of the product path; the numbers go into DESIGN.md section 6."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import miphei_vit_amd.ops as ops

M = 16 * 329
shapes = [("qkv", M, 4608, 1536), ("proj", M, 1536, 1536), ("fc1", M, 8192, 1536), ("fc2", M, 1536, 4096),
          ("dqkv", M, 1536, 4608), ("dfc1", M, 1536, 8192), ("dfc2", M, 4096, 1536), ("sq8k", 8192, 8192, 8192)]


def timeit(fn, it=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it


for name, m, n, k in shapes:
    a = torch.randn(m, k, device="cuda").bfloat16()
    b = torch.randn(n, k, device="cuda").bfloat16()
    c = torch.empty(m, n, device="cuda", dtype=torch.bfloat16)
    t_ours = timeit(lambda: ops.gemm(a, b, c))
    bt = b.t()
    t_blas = timeit(lambda: torch.matmul(a, bt, out=c))
    fl = 2 * m * n * k
    print(f"{name:6s} M={m} N={n} K={k}: ours {t_ours*1e3:7.1f} us {fl/t_ours/1e9:7.1f} TF/s | hipBLASLt {t_blas*1e3:7.1f} us "
          f"{fl/t_blas/1e9:7.1f} TF/s", flush=True)
