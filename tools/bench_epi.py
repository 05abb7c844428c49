"""Encoder GEMMs WITH their fused epilogues at the training shapes (B=16, N=329): the product dispatch against alternatives.
  python tools/bench_epi.py                 # product library
  MIPHEI_DBG_LIB=1 MVIT_GEMM_BIG_TILE=0 python tools/bench_epi.py   # measurement library (make -C miphei-vit_amd/csrc dbg)
Prints us per launch; `plain` = same shape with the plain bf16 store; hipBLASLt = torch.matmul of the bare product."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from miphei_vit_amd import _lib
if os.environ.get("MIPHEI_DBG_LIB") == "1":
    _lib.LIB_PATH = _lib.DBG_LIB_PATH
if os.environ.get("MIPHEI_LIB"):                      # a compile-time variant build (make BUILD=... LIB=... EXTRA=-D...)
    _lib.LIB_PATH = os.path.abspath(os.environ["MIPHEI_LIB"])
import miphei_vit_amd.ops as ops
from miphei_vit_amd.ops import EPI_DSWIGLU, EPI_RESID, EPI_SWIGLU, OUT_F32

M, D, Hd = 16 * 329, 1536, 8192
dev = "cuda"
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


def rnd(*s, dt=bf):
    return torch.randn(*s, device=dev).to(dt)


x, w_qkv, w_proj, w_fc1, w_fc2 = rnd(M, D), rnd(3 * D, D), rnd(D, D), rnd(Hd, D), rnd(D, Hd // 2)
w_fc2t = rnd(Hd // 2, D)
g, u = rnd(M, Hd // 2), torch.empty(M, Hd, device=dev, dtype=bf)
res, out32 = rnd(M, D, dt=torch.float32), torch.empty(M, D, device=dev, dtype=torch.float32)
gam, bias_d, bias_h = rnd(D, dt=torch.float32), rnd(D, dt=torch.float32), rnd(Hd, dt=torch.float32)
c_qkv, c_g, c_d, c_h, c_u = (torch.empty(M, n, device=dev, dtype=bf) for n in (3 * D, Hd // 2, D, Hd // 2, Hd))
rows = [
    ("qkv   store   N=4608 K=1536", lambda: ops.gemm(x, w_qkv, c_qkv, bias=bias_h[:3 * D]), None, (x, w_qkv)),
    ("proj  resid   N=1536 K=1536", lambda: ops.gemm(x, w_proj, out32, bias=bias_d, gamma=gam, aux=res, epi=EPI_RESID, flags=OUT_F32),
     lambda: ops.gemm(x, w_proj, c_d), (x, w_proj)),
    ("fc1   swiglu  N=8192 K=1536", lambda: ops.gemm(x, w_fc1, c_g, bias=bias_h, aux=u, epi=EPI_SWIGLU),
     lambda: ops.gemm(x, w_fc1, c_u), (x, w_fc1)),
    ("fc2   resid   N=1536 K=4096", lambda: ops.gemm(g, w_fc2, out32, bias=bias_d, gamma=gam, aux=res, epi=EPI_RESID, flags=OUT_F32),
     lambda: ops.gemm(g, w_fc2, c_d), (g, w_fc2)),
    ("dfc2  dswiglu N=4096 K=1536", lambda: ops.gemm(x, w_fc2t, c_u, aux=u, epi=EPI_DSWIGLU),
     lambda: ops.gemm(x, w_fc2t, c_h), (x, w_fc2t)),
]
u.copy_(rnd(M, Hd))
for name, fused, plain, (a, b) in rows:
    t = timeit(fused)
    tp = timeit(plain) if plain else float("nan")
    tb = timeit(lambda: torch.matmul(a, b.t()))
    print(f"{name}: fused {t:7.1f} us   plain {tp:7.1f}   hipBLASLt {tb:7.1f}", flush=True)
