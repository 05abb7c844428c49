"""Micro-benchmark of the seven decoder convolutions at the training shapes (B=16, 256x256): forward (+BN statistics), input
gradient, weight gradient, each on the path the engine uses, with GFLOP and TFLOP/s.  MIPHEI_DBG_LIB=1 selects the measurement
library (dispatch knobs from the environment, e.g. MVIT_GEMM_BIG_TILE=0)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from miphei_vit_amd import _lib
if os.environ.get("MIPHEI_DBG_LIB") == "1":
    _lib.LIB_PATH = _lib.DBG_LIB_PATH
if os.environ.get("MIPHEI_LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["MIPHEI_LIB"])
import miphei_vit_amd.ops as ops

bf = torch.bfloat16
B = int(os.environ.get("B", 16))


def timeit(fn, it=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e3


# name, r_in, cin(pad), r_out, cout, stride
layers = [("conv0", 256, 8, 128, 48, 2), ("conv1", 128, 48, 64, 96, 2), ("conv2", 64, 96, 32, 192, 2),
          ("fus0", 32, 1728, 32, 256, 1), ("fus1", 64, 352, 64, 128, 1), ("fus2", 128, 176, 128, 64, 1), ("fus3", 256, 72, 256, 32, 1)]
only = os.environ.get("ONLY")
for name, r_in, cp, r_out, cout, stride in layers:
    if only and name not in only.split(","):
        continue
    x = torch.randn(B, r_in, r_in, cp, device="cuda").to(bf)
    Mo = B * r_out * r_out
    wk = (torch.randn(cout, 9 * cp, device="cuda") * 0.05).to(bf)
    wd = (torch.randn(cp, 9 * cout, device="cuda") * 0.05).to(bf)
    y = torch.empty(Mo, cout, device="cuda", dtype=bf)
    st = torch.zeros(ops.STAT_SLOTS * 2 * cout, device="cuda", dtype=torch.float64)
    gf = 2.0 * Mo * 9 * cp * cout / 1e9
    t_f = timeit(lambda: ops.gemm(x, wk, y, M=Mo, amode=ops.A_CONV3, conv=(r_in, r_in, cp, cp, r_out, r_out, stride), epi=ops.EPI_STATS,
                                  stats=st, nslots=ops.STAT_SLOTS))
    dy = torch.randn(Mo, cout, device="cuda").to(bf)
    dx = torch.empty(B * r_in * r_in, cp, device="cuda", dtype=bf)
    t_d = timeit(lambda: ops.gemm(dy, wd, dx, M=B * r_in * r_in, N=cp, amode=ops.A_CONV3_T, conv=(r_out, r_out, cout, cout, r_in, r_in, stride),
                                  ldc=cp))
    K9 = 9 * cp
    it_, jt = (128, 32) if cout <= 32 else (64, 128)
    tiles = ((K9 + it_ - 1) // it_) * ((cout + jt - 1) // jt)
    ms = max(1, min(int(os.environ.get("WG_SLOTS", "768")) // tiles, (Mo + 255) // 256))      # (WG_SLOTS: scan of the block budget)
    dwt = torch.zeros(K9, cout, device="cuda")
    t_w = timeit(lambda: ops.gemm_tn(x, dy, dwt, M=Mo, I=K9, J=cout, ldb=cout, ldci=cout, msplit=ms, conv=(r_in, r_in, cp, cp, r_out, r_out, stride)))
    line = (f"{name}: {gf:6.1f} GF | fwd {t_f:7.1f} us {gf / t_f:5.2f} PF | dgrad {t_d:7.1f} us {gf / t_d:5.2f} PF | "
            f"wgrad {t_w:7.1f} us {gf / t_w:5.2f} PF (msplit {ms})")
    if stride == 1 and cp % 8 == 0 and cout % 8 == 0:
        w4 = torch.randn(cout, cp, 3, 3, device="cuda") * 0.05
        wf, wb = ops.pack_conv3x3_chunked(w4), ops.pack_conv3x3_chunked(w4, dgrad=True)
        t_cf = timeit(lambda: ops.conv3x3_chunked(x, wf, y, B=B, H=r_in, W=r_in, cin=cp, ldx=cp, cout=cout, ldy=cout, stats=st, nslots=ops.STAT_SLOTS))
        t_cd = timeit(lambda: ops.conv3x3_chunked(dy, wb, dx, B=B, H=r_in, W=r_in, cin=cout, ldx=cout, cout=cp, ldy=cp))
        t_pk = timeit(lambda: ops.pack_conv3x3_chunked(w4))
        dwn = torch.zeros(cout, 9 * cp, device="cuda")
        t_cw = timeit(lambda: ops.conv3x3_chunked_wgrad(x, dy, dwn, B=B, H=r_in, W=r_in, cin=cp, cin_pad=cp, ldx=cp, cout=cout, ldy=cout))
        line += (f" | chunked fwd {t_cf:7.1f} us {gf / t_cf:5.2f} PF, dgrad {t_cd:7.1f} us {gf / t_cd:5.2f} PF, wgrad {t_cw:7.1f} us "
                 f"{gf / t_cw:5.2f} PF, pack {t_pk:5.1f} us")
    print(line, flush=True)
