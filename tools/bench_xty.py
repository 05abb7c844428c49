import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import miphei_vit_amd.ops as ops
M, D = 5264, 1536
t = torch.randn(M, 16, device="cuda").bfloat16()
dqkv = torch.randn(M, 3 * D, device="cuda").bfloat16()
h1 = torch.randn(M, D, device="cuda").bfloat16()
dB = torch.zeros(8, D, device="cuda"); dA = torch.zeros(2, D, 8, device="cuda")
def tm(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
print("dB (R=8, Y=dq strided)  %.1f us" % tm(lambda: ops.skinny_xty(t, dqkv, dB, ldx=16, ldy=3 * D, osr=D, osn=1, M=M, N=D, R=8)))
print("dA (R=16, Y=h1)         %.1f us" % tm(lambda: ops.skinny_xty(t, h1, dA, ldx=16, ldy=D, osb=D * 8, rgrp=8, osr=1, osn=8, M=M, N=D, R=16)))
for dbg, name in ((1, "no global atomics"), (2, "no fma loop")):
    print(name, "dB %.1f us" % tm(lambda: ops.skinny_xty(t, dqkv, dB, ldx=16, ldy=3 * D, osr=D, osn=1 | (dbg << 20), M=M, N=D, R=8)),
          "dA %.1f us" % tm(lambda: ops.skinny_xty(t, h1, dA, ldx=16, ldy=D, osb=D * 8, rgrp=8, osr=1, osn=8 | (dbg << 20), M=M, N=D, R=16)))
