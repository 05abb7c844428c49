"""Measurement only: time of mvit_gemm_bf16 vs K at the encoder's M and a given N (affine fit: per-launch fixed cost and
cost per 64-wide K step), beside hipBLASLt (torch.matmul).  Usage: python tools/gemm_kscan.py [N ...]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import miphei_vit_amd.ops as ops

M = 16 * 329


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
    return e0.elapsed_time(e1) / it * 1e3


for n in [int(v) for v in sys.argv[1:]] or [4608, 1536, 8192]:
    pts = []
    for k in (512, 1024, 1536, 3072, 6144):
        a = torch.randn(M, k, device="cuda").bfloat16()
        b = torch.randn(n, k, device="cuda").bfloat16()
        c = torch.empty(M, n, device="cuda", dtype=torch.bfloat16)
        bt = b.t()
        pts.append((k // 64, timeit(lambda: ops.gemm(a, b, c)), timeit(lambda: torch.matmul(a, bt, out=c))))
    for col, name in ((1, "ours"), (2, "hipBLASLt")):
        xs, ys = [p[0] for p in pts], [p[col] for p in pts]
        mx, my = sum(xs) / len(xs), sum(ys) / len(ys)
        slope = sum((x - mx) * (y - my) for x, y in zip(xs, ys)) / sum((x - mx) ** 2 for x in xs)
        print(f"N={n} {name:9s}: " + " ".join(f"K={64*x}:{y:6.1f}" for x, y in zip(xs, ys)) + f" us | fixed {my - slope*mx:5.1f} us, {slope:5.3f} us per K step", flush=True)
