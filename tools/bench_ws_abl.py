"""Plain-store GEMMs of the encoder on a chosen library build (MIPHEI_LIB=path: e.g. an ablation build of csrc/gemm_ws.hip,
make DEBUG_KNOBS=1 BUILD=build_aN LIB=../libmiphei_aN.so EXTRA=-DMVIT_WS_ABLATE=N): us per launch and per K tile."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from miphei_vit_amd import _lib
if os.environ.get("MIPHEI_LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["MIPHEI_LIB"])
import miphei_vit_amd.ops as ops

M = 16 * 329
shapes = [("proj", M, 1536, 1536), ("fc2", M, 1536, 4096), ("dfc1", M, 1536, 8192), ("qkv", M, 4608, 1536), ("dfc2", M, 4096, 1536)]


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


res = {}
for name, m, n, k in shapes:
    a = torch.randn(m, k, device="cuda").bfloat16()
    b = torch.randn(n, k, device="cuda").bfloat16()
    c = torch.empty(m, n, device="cuda", dtype=torch.bfloat16)
    res[name] = timeit(lambda: ops.gemm(a, b, c))
per = (res["dfc1"] - res["proj"]) / ((8192 - 1536) / 64)
print(" ".join(f"{k} {v:6.1f}" for k, v in res.items()), f"| us per K tile (one round, from dfc1 - proj): {per:.3f}")
