"""Attention fwd/bwd at the H-Optimus-0 shape (B=16, N=329, H=24, Dh=64): timing + achieved TFLOP/s."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from miphei_vit_amd import _lib
if os.environ.get("MIPHEI_LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["MIPHEI_LIB"])
elif os.environ.get("MIPHEI_DBG_LIB") == "1":
    _lib.LIB_PATH = _lib.DBG_LIB_PATH
import miphei_vit_amd.ops as ops
B, N, H, Dh = 16, int(sys.argv[1]) if len(sys.argv) > 1 else 329, 24, 64
qkv = torch.randn(B, N, 3, H, Dh, device="cuda").bfloat16()
out = torch.empty(B, N, H * Dh, device="cuda", dtype=torch.bfloat16)
lse = torch.empty(B, H, N, device="cuda")
dO = torch.randn_like(out); dqkv = torch.empty_like(qkv); dsum = torch.empty_like(lse)
sc = Dh ** -0.5
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
res = torch.empty_like(out)
f0 = t(lambda: ops.attention_fwd(qkv, out, lse, B, N, H, Dh, sc))
f = t(lambda: ops.attention_fwd(qkv, out, lse, B, N, H, Dh, sc, out_res=res))
b0 = t(lambda: ops.attention_bwd(qkv, out, dO, lse, dsum, dqkv, B, N, H, Dh, sc))
b = t(lambda: ops.attention_bwd(qkv, out, dO, lse, dsum, dqkv, B, N, H, Dh, sc, out_res=res))
fl = 4.0 * B * H * N * N * Dh
print(f"N={N} fwd {f*1e3:.1f} us ({fl/f/1e9:.1f} TF/s; {f0*1e3:.1f} us without the O residual)   "
      f"bwd {b*1e3:.1f} us ({2.5*fl/b/1e9:.1f} TF/s useful; {b0*1e3:.1f} us without)")
if len(sys.argv) > 2 and sys.argv[2] == "ours":
    sys.exit(0)
# yardstick: the vendor attention behind torch SDPA (CK / AOTriton flash attention) on the same problem
import torch.nn.functional as F
q, k, v = (qkv[:, :, i].permute(0, 2, 1, 3).contiguous().requires_grad_(True) for i in range(3))   # [B, H, N, Dh]
try:
    fs = t(lambda: F.scaled_dot_product_attention(q, k, v))
    o = F.scaled_dot_product_attention(q, k, v)
    go = torch.randn_like(o)
    def fb():
        o2 = F.scaled_dot_product_attention(q, k, v)
        o2.backward(go)
    fbt = t(fb)
    print(f"torch SDPA: fwd {fs*1e3:.1f} us   fwd+bwd {fbt*1e3:.1f} us (bwd ~{(fbt-fs)*1e3:.1f} us)")
except Exception as e:  # noqa: BLE001
    print("torch SDPA unavailable:", str(e)[:200])
