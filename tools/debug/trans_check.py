"""Correctness + timing of the plain-store GEMM on the measurement library (transposed-accumulate experiment)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from miphei_vit_amd import _lib
if os.environ.get("MIPHEI_DBG_LIB") == "1":
    _lib.LIB_PATH = _lib.DBG_LIB_PATH
import miphei_vit_amd.ops as ops
def rel(a, b): return float((a.double() - b.double()).norm() / b.double().norm())
def timeit(fn, it=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e3
for M, N, K in [(5264, 4608, 1536), (5264, 8192, 1536), (5264, 1536, 4096), (1100, 200, 200), (2000, 384, 136), (5264, 1536, 1536)]:
    a = torch.randn(M, K, device="cuda").bfloat16(); b = torch.randn(N, K, device="cuda").bfloat16(); bias = torch.randn(N, device="cuda")
    ref = a.float() @ b.float().t() + bias
    c = torch.empty(M, N, device="cuda", dtype=torch.bfloat16); ops.gemm(a, b, c, bias=bias)
    cf = torch.empty(M, N, device="cuda"); ops.gemm(a, b, cf, bias=bias, flags=ops.OUT_F32)
    c2 = c.clone(); ops.gemm(a, b, c2, bias=bias, flags=ops.ACCUM_BF16)
    ca = torch.zeros(M, N, device="cuda"); ops.gemm(a, b, ca, flags=ops.OUT_F32 | ops.ATOMIC, ksplit=2)
    print(M, N, K, "bf16", round(rel(c.float(), ref), 5), "f32", round(rel(cf, ref), 7), "accum", round(rel(c2.float(), 2 * ref), 5),
          "atomic", round(rel(ca, ref - bias), 7), "us", round(timeit(lambda: ops.gemm(a, b, c, bias=bias)), 1), flush=True)
