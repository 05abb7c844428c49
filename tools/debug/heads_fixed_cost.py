"""Fixed cost (launch + per-block operand set-up) of the head kernels: time at M = full size and at a sliver of it."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import miphei_vit_amd.ops as ops
dev, bf = "cuda", torch.bfloat16
NH, B, H, W = 16, 16, 256, 256
Mfull = B * H * W
x = torch.randn(Mfull, 32, device=dev).to(bf)
W1 = torch.randn(NH, 16, 32, device=dev) * 0.2
b1 = torch.randn(NH * 16, device=dev) * 0.1
scale = torch.rand(NH * 16, device=dev) + 0.5
shift = torch.randn(NH * 16, device=dev) * 0.1
W2 = torch.randn(NH, 16, device=dev) * 0.3
b2 = torch.randn(NH, device=dev) * 0.1
G = torch.empty(Mfull, 16, device=dev, dtype=bf)
W3 = (torch.randn(NH, 32, 9, device=dev) * 0.1).transpose(1, 2).contiguous()
b3 = torch.randn(NH, device=dev) * 0.1
out = torch.empty(B, NH, H, W, device=dev)
def timeit(fn, it=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e3
for M in (Mfull, Mfull // 4, 4096):
    print(f"gate_fwd  M={M:8d}: {timeit(lambda: ops.heads_gate_fwd(x, W1, b1, scale, shift, W2, b2, G, M, NH)):7.1f} us")
for b in (B, B // 4, 1):
    print(f"conv_fwd  B={b:2d}: {timeit(lambda: ops.heads_conv_fwd(x, G, W3, b3, out, b, H, W, NH)):7.1f} us")
