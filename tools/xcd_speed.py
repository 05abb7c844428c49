"""Does placing the band items of fc1 on the fastest XCDs pay?  Measures the per-XCD speed (miphei_vit_amd/xcd.py), then times
fc1 + SwiGLU (M = 5264, N = 8192, K = 1536: 1280 tiles + 192 band items on 256 CUs) with the items spread evenly (no ranking), on
the six FASTEST XCDs (the product's choice), on the six SLOWEST (worst case), and a full training-step A/B through bench.py."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from miphei_vit_amd import _lib as L
if os.environ.get("MIPHEI_LIB"):
    L.LIB_PATH = os.path.abspath(os.environ["MIPHEI_LIB"])
import miphei_vit_amd.ops as ops
from miphei_vit_amd import xcd

M, N, K = 16 * 329, 8192, 1536
torch.manual_seed(0)
x = torch.randn(M, K, device="cuda").bfloat16()
w = (torch.randn(N, K, device="cuda") * 0.03).bfloat16()
bias = torch.zeros(N, device="cuda")
u = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
g = torch.empty(M, N // 2, device="cuda", dtype=torch.bfloat16)


def fc1():
    ops.gemm(x, w, g, bias=bias, aux=u, epi=ops.EPI_SWIGLU)


def timeit(fn, it=40):
    for _ in range(8):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e3


def set_rank(rank):
    arr = (C.c_int * 8)(*rank) if rank is not None else None
    L.check(L.lib().mvit_set_xcd_rank(arr), "mvit_set_xcd_rank")


for rep in range(3):
    dur, rank = xcd.measure()
    print("probe durations per XCD (us):", [round(d / 100, 1) for d in dur], "rank (0 = fastest):", rank, flush=True)
worst = [7 - r for r in rank]
for rep in range(3):
    res = []
    for name, rk in (("even spread", None), ("fastest six", rank), ("slowest six", worst), ("identity", list(range(8)))):
        set_rank(rk)
        res.append(f"{name} {timeit(fc1):6.1f}")
    print("fc1 + SwiGLU (us):", " | ".join(res), flush=True)
set_rank(None)
