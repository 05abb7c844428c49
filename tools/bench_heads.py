"""Timing of the fused output-head kernels alone (B=16, 256x256, 16 heads)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from miphei_vit_amd import _lib
if os.environ.get("MIPHEI_LIB"):                      # a variant / older build of the library for same-box comparisons
    _lib.LIB_PATH = os.path.abspath(os.environ["MIPHEI_LIB"])
import miphei_vit_amd.ops as ops

B, H, W, NH = 16, 256, 256, 16
M, nch, dev = B * H * W, NH * 16, "cuda"
x = torch.randn(M, 32, device=dev).bfloat16()
W1, b1 = torch.randn(nch, 32, device=dev) * 0.3, torch.randn(nch, device=dev) * 0.1
gamma, beta = torch.ones(nch, device=dev), torch.zeros(nch, device=dev)
W2, b2 = torch.randn(nch, device=dev) * 0.3, torch.zeros(NH, device=dev)
W3, b3 = torch.randn(NH, 9, 32, device=dev) * 0.1, torch.zeros(NH, device=dev)
mom = torch.zeros(32 * 1056, device=dev, dtype=torch.float64)
mom_sum = torch.zeros(1056, device=dev, dtype=torch.float64)
rm, rv = torch.zeros(nch, device=dev), torch.ones(nch, device=dev)
scale, shift, mean, rstd = (torch.empty(nch, device=dev) for _ in range(4))
G = torch.empty(M, 16, device=dev, dtype=torch.bfloat16)
out = torch.empty(B, NH, H, W, device=dev)
dY = torch.randn(B, NH, H, W, device=dev)
cs = torch.empty(ops.heads_conv_bwd_scratch_bytes(M) // 4 + 1, device=dev)
gs = torch.empty(ops.heads_gate_bwd_scratch_bytes() // 4, device=dev)
dG, dXc = torch.empty(M, 16, device=dev), torch.empty(M, 32, device=dev)
dW3, db3 = torch.empty(NH * 9, 32, device=dev), torch.zeros(64, 32, device=dev)
dW1, dg, db, dW2, db2 = (torch.zeros(n, device=dev) for n in (nch * 32, nch, nch, nch, NH))
dF = torch.empty(M, 32, device=dev, dtype=torch.bfloat16)


def t(name, fn, it=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it):
        fn()
    e1.record()
    torch.cuda.synchronize()
    print(f"{name:10s} {e0.elapsed_time(e1) / it * 1e3:8.1f} us", flush=True)


t("moments", lambda: ops.heads_moments(x, mom, M, 32))
t("bn_stats", lambda: ops.heads_bn_from_moments(mom, W1, b1, gamma, beta, rm, rv, scale, shift, mean, rstd, mom_sum, NH, 32, M, 1e-5, 0.1, True))
t("gate_fwd", lambda: ops.heads_gate_fwd(x, W1, b1, scale, shift, W2, b2, G, M, NH))
t("conv_fwd", lambda: ops.heads_conv_fwd(x, G, W3, b3, out, B, H, W, NH))
t("conv_bwd", lambda: ops.heads_conv_bwd(dY, out, x, G, W3, cs, dG, dXc, dW3, db3, B, H, W, NH))
t("gate_bwd", lambda: ops.heads_gate_bwd(x, G, dG, dXc, W1, b1, scale, shift, mean, rstd, gamma, W2, mom_sum, gs, dW1, dg, db, dW2, db2, dF, M, NH))
