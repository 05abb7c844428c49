"""Micro-benchmark of mvit_gemm_bf16 on the H-Optimus-0 GEMM shapes (B=16, 256x256 tiles)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import miphei_vit_amd.ops as ops

M = 16 * 329
shapes = [("qkv", M, 4608, 1536), ("proj", M, 1536, 1536), ("fc1", M, 8192, 1536), ("fc2", M, 1536, 4096),
          ("sq8k", 8192, 8192, 8192)]
for name, m, n, k in shapes:
    a = torch.randn(m, k, device="cuda").bfloat16()
    b = torch.randn(n, k, device="cuda").bfloat16()
    c = torch.empty(m, n, device="cuda", dtype=torch.bfloat16)
    for _ in range(3):
        ops.gemm(a, b, c)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    it = 20
    e0.record()
    for _ in range(it):
        ops.gemm(a, b, c)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / it
    print(f"{name:6s} M={m} N={n} K={k}: {ms*1e3:8.1f} us  {2*m*n*k/ms/1e9:8.1f} TF/s", flush=True)
