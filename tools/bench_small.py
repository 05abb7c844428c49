"""Micro-benchmarks of the non-GEMM kernels at the training shapes (B=16): LayerNorm (+ fused LoRA down-projection), the direct
3x3 convolution of the last fusion block against the implicit-GEMM path it replaces.  MIPHEI_DBG_LIB=1 selects the measurement library."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from miphei_vit_amd import _lib
if os.environ.get("MIPHEI_DBG_LIB") == "1":
    _lib.LIB_PATH = _lib.DBG_LIB_PATH
import miphei_vit_amd.ops as ops

bf = torch.bfloat16


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


M, D = 16 * 329, 1536
x = torch.randn(M, D, device="cuda")
w, b = torch.randn(D, device="cuda"), torch.randn(D, device="cuda")
h = torch.empty(M, D, device="cuda", dtype=bf)
A = torch.randn(16, D, device="cuda").to(bf)
t = torch.empty(M, 16, device="cuda", dtype=bf)
print(f"ln_fwd                 {timeit(lambda: ops.layernorm_fwd(x, w, b, h)):7.1f} us")
print(f"skinny_xw (t = h A)    {timeit(lambda: ops.skinny_xw(h, A, t)):7.1f} us")
dqkv = torch.randn(M, 3 * D, device="cuda").to(bf)
Bq, Bv = torch.randn(8, D, device="cuda").to(bf), torch.randn(8, D, device="cuda").to(bf)
print(f"skinny_xw2 (dt, R = 8) {timeit(lambda: ops.skinny_xw2(dqkv, Bq, t, dqkv.view(-1)[2 * D:], Bv, t.view(-1)[8:], ldx=3 * D, ldw=D, ldo=16, M=M, K=D, R=8)):7.1f} us")
print(f"ln_fwd + lora fused    {timeit(lambda: ops.layernorm_lora_fwd(x, w, b, h, A, t)):7.1f} us", flush=True)

B, S, cin, cp, cout = 16, 256, 67, 72, 32
wgt = torch.randn(cout, cin, 3, 3, device="cuda") * 0.05
xb = torch.randn(B, S, S, cp, device="cuda").to(bf)
y = torch.empty(B * S * S, cout, device="cuda", dtype=bf)
st = torch.zeros(32 * 2 * cout, device="cuda", dtype=torch.float64)
wp = ops.pack_conv3x3_direct(wgt, cout, cp, rot=3)
wk = torch.empty(cout, 9 * cp, device="cuda", dtype=bf)
wd = torch.empty(cp, 9 * cout, device="cuda", dtype=bf)
ops.pack_conv3x3_weights(wgt, wk, wd, rot=3)
print(f"fus3 fwd  direct       {timeit(lambda: ops.conv3x3_direct(xb, wp, y, B=B, H=S, W=S, cin_pad=cp, ldx=cp, cout=cout, ldy=cout, stats=st, nslots=32)):7.1f} us")
print(f"fus3 fwd  implicit     {timeit(lambda: ops.gemm(xb, wk, y, M=B*S*S, amode=ops.A_CONV3, conv=(S, S, cp, cp, S, S, 1), epi=ops.EPI_STATS, stats=st, nslots=32)):7.1f} us", flush=True)
dy = torch.randn(B * S * S, cout, device="cuda").to(bf)
dx = torch.empty(B * S * S, 64, device="cuda", dtype=bf)
wpb = ops.pack_conv3x3_direct(wgt, 64, cout, rot=3, dgrad=True)
print(f"fus3 dgrad direct      {timeit(lambda: ops.conv3x3_direct(dy, wpb, dx, B=B, H=S, W=S, cin_pad=cout, ldx=cout, cout=64, ldy=64)):7.1f} us")
print(f"fus3 dgrad implicit    {timeit(lambda: ops.gemm(dy, wd, dx, M=B*S*S, N=64, amode=ops.A_CONV3_T, conv=(S, S, cout, cout, S, S, 1), ldc=64)):7.1f} us", flush=True)
dwn = torch.zeros(cout, 9 * cp, device="cuda")
dwt = torch.zeros(9 * cp, cout, device="cuda")
print(f"fus3 wgrad direct      {timeit(lambda: ops.conv3x3_direct_wgrad(xb, dy, dwn, B=B, H=S, W=S, cin_pad=cp, ldx=cp, cout=cout, ldy=cout)):7.1f} us")
print(f"fus3 wgrad TN GEMM     {timeit(lambda: ops.gemm_tn(xb, dy, dwt, M=B*S*S, I=9*cp, J=cout, ldb=cout, ldci=cout, msplit=170, conv=(S, S, cp, cp, S, S, 1))):7.1f} us", flush=True)
dh = torch.randn(M, D, device="cuda").to(bf)
dx = torch.randn(M, D, device="cuda")
gam = torch.randn(D, device="cuda")
dyb = torch.empty(M, D, device="cuda", dtype=bf)
print(f"ln_bwd                 {timeit(lambda: ops.layernorm_bwd(dh, x, w, dx, gam, dyb, 1e-6, True)):7.1f} us   (128 MB algorithmic)")
