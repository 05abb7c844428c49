"""Measurement only: times the one-wave-per-SIMD GEMM loop with parts of it compiled out (MVIT_ABLATE bit 0 = no operand
DMA in the pinned loop, bit 1 = no LDS fragment reads, bit 2 = no MFMAs; results are garbage by construction).
Usage: python tools/gemm_ablate.py <path to a library built with -DMVIT_ABLATE=n> ; prints us per launch."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import miphei_vit_amd._lib as _lib
if len(sys.argv) > 1 and sys.argv[1] != "-":
    _lib.LIB_PATH = os.path.abspath(sys.argv[1])
import miphei_vit_amd.ops as ops

shapes = [("sq8k", 8192, 8192, 8192), ("dfc1", 5264, 1536, 8192)]
for name, m, n, k in shapes:
    a = torch.randn(m, k, device="cuda").bfloat16()
    b = torch.randn(n, k, device="cuda").bfloat16()
    c = torch.empty(m, n, device="cuda", dtype=torch.bfloat16)
    for _ in range(5):
        ops.gemm(a, b, c)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        ops.gemm(a, b, c)
    e1.record()
    torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 20
    print(f"{os.path.basename(sys.argv[1]) if len(sys.argv) > 1 else 'product'} {name}: {t*1e3:8.1f} us  {2*m*n*k/t/1e9:7.1f} TF/s-equivalent", flush=True)
