"""One GEMM shape, few launches (for rocprofv3 --pmc runs)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import miphei_vit_amd.ops as ops
m, n, k = (int(v) for v in sys.argv[1:4])
a = torch.randn(m, k, device="cuda").bfloat16()
b = torch.randn(n, k, device="cuda").bfloat16()
c = torch.empty(m, n, device="cuda", dtype=torch.bfloat16)
for _ in range(5):
    ops.gemm(a, b, c)
torch.cuda.synchronize()
