"""Where the fixed cost of a wave-specialised GEMM launch goes (measurement build of csrc/gemm_ws.hip):

    make -C miphei-vit_amd/csrc DEBUG_KNOBS=1 BUILD=build_tm LIB=../libmiphei_tm.so EXTRA=-DMVIT_WS_TIMING
    MIPHEI_LIB=miphei-vit_amd/libmiphei_tm.so python tools/ws_timing.py

Wave 0 (consumer) and wave 8 (producer) of every block stamp the phases of the block's first tile with s_memtime (shader cycles) and
the block's begin / end with s_memrealtime (100 MHz, common to all CUs).  Per shape: launch skew over the blocks, cycles from block
start to the first landed K tile (operand cold start), cycles per K tile of the first tile, epilogue issue, store drain, the span
first-block-start -> last-block-end against the HIP-event time of the launch (the difference is launch / completion overhead)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from miphei_vit_amd import _lib
if os.environ.get("MIPHEI_LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["MIPHEI_LIB"])
import miphei_vit_amd.ops as ops

M = 16 * 329
shapes = [("dproj", M, 1536, 1536, "store"), ("dqkv", M, 1536, 4608, "store"), ("dfc1", M, 1536, 8192, "store"),
          ("qkv", M, 4608, 1536, "store"), ("dfc2", M, 4096, 1536, "store"), ("proj+res", M, 1536, 1536, "resid"),
          ("fc2+res", M, 1536, 4096, "resid"), ("fc1+swiglu", M, 8192, 1536, "swiglu"), ("dfc2+dswiglu", M, 4096, 1536, "dswiglu")]
if os.environ.get("WS_TIMING_ONLY"):
    shapes = [s_ for s_ in shapes if s_[0] in os.environ["WS_TIMING_ONLY"].split(",")]
NB = 256


def med(t):
    return float(t.double().median())


def run(name, m, n, k, epi, warm_prev=None):
    a = torch.randn(m, k, device="cuda").bfloat16()
    b = (torch.randn(n, k, device="cuda") * 0.05).bfloat16()
    if epi == "store":
        c = torch.empty(m, n, device="cuda", dtype=torch.bfloat16)
        kw = {}
    elif epi == "swiglu":      # n = 2 * hidden packed [a32 | b32]; output [m, n / 2], saved pre-activation [m, n]
        c = torch.empty(m, n // 2, device="cuda", dtype=torch.bfloat16)
        kw = dict(aux=torch.empty(m, n, device="cuda", dtype=torch.bfloat16), bias=torch.zeros(n, device="cuda"), epi=ops.EPI_SWIGLU)
    elif epi == "dswiglu":     # n = hidden; output [m, 2 n] packed, saved pre-activation [m, 2 n]
        c = torch.empty(m, 2 * n, device="cuda", dtype=torch.bfloat16)
        kw = dict(aux=torch.randn(m, 2 * n, device="cuda").bfloat16(), epi=ops.EPI_DSWIGLU)
    else:
        c = torch.empty(m, n, device="cuda", dtype=torch.float32)
        kw = dict(aux=torch.randn(m, n, device="cuda"), gamma=torch.ones(n, device="cuda"), bias=torch.zeros(n, device="cuda"),
                  epi=ops.EPI_RESID, flags=ops.OUT_F32)
    prof = torch.zeros(NB * 16, device="cuda", dtype=torch.float64)
    # a 600 MB sweep between the launches: the operands of the timed launch come from HBM (as the weights do in the step; the
    # activations of the step were written by the previous kernel and may still sit in the Infinity Cache -- `warm` rows below)
    junk = torch.empty(300 * 1024 * 1024 // 4, device="cuda")
    rows = {}
    for mode in ("cold", "warm"):
        ts = []
        keep = None
        for it in range(6):
            if mode == "cold":
                junk.add_(1.0)
            else:
                ops.gemm(a, b, c, **kw)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            ops.gemm(a, b, c, stats=prof, nslots=1, **kw)
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
            keep = prof.view(torch.int64).view(NB, 16).clone()
        if os.environ.get("WS_TIMING_DUMP"):
            import numpy as np
            os.makedirs(os.environ["WS_TIMING_DUMP"], exist_ok=True)
            np.save(os.path.join(os.environ["WS_TIMING_DUMP"], f"{name.replace('+', '_')}_{mode}.npy"), keep.cpu().numpy())
        q = keep[keep[:, 0] != 0]
        nb = q.shape[0]
        c_rt0, c_c0, c_first, c_kend, c_epi, c_end, c_drain, c_rt1 = (q[:, i] for i in range(8))
        p_rt0, p_c0, p_iss, p_land, p_end, p_rt1 = (q[:, 8 + i] for i in range(6))
        t0 = int(c_rt0.min())
        span = (int(c_rt1.max()) - t0) / 100.0
        skew = (c_rt0 - t0).double() / 100.0
        endskew = (int(c_rt1.max()) - c_rt1).double() / 100.0
        nk = k // 64
        clk = (c_drain - c_c0).double() / ((c_rt1 - c_rt0).double() * 10.0)   # cycles per ns = GHz
        first_req = med(p_iss - p_c0)
        rows[mode] = (sorted(ts)[len(ts) // 2], span, float(skew.max()), med(skew), med(p_land - p_c0), float((p_land - p_c0).max()),
                      med(c_first - c_c0), med(c_kend - c_first) / nk, med(c_epi - c_kend), med(c_drain - c_end), float(endskew.max()),
                      med(endskew), med(clk), nb)
        xcd = torch.arange(keep.shape[0], device=keep.device)[keep[:, 0] != 0] % 8
        dur = (c_rt1 - c_rt0).double() / 100.0
        rows[mode] = rows[mode] + ([round(float(dur[xcd == j].mean()), 1) for j in range(8)],
                                   [round(float(clk[xcd == j].mean()), 3) for j in range(8)], first_req)
    for mode, r in rows.items():
        print(f"{name:9s} {mode}: event {r[0]:6.1f} us | span {r[1]:6.1f} | start skew max {r[2]:4.1f} med {r[3]:4.1f} us | first K tile landed "
              f"{r[4]:6.0f} cyc (max {r[5]:6.0f}) | consumer start->first tile {r[6]:6.0f} | K tile {r[7]:6.0f} cyc | epilogue issue {r[8]:6.0f} | "
              f"store drain {r[9]:6.0f} | end skew max {r[10]:4.1f} med {r[11]:4.1f} us | clock {r[12]:.2f} GHz | blocks {r[13]}\n"
              f"          block duration by XCD (us) {r[14]} | clock by XCD (GHz) {r[15]} | producers' first request issued {r[16]:.0f} cyc after entry", flush=True)


if __name__ == "__main__":
    torch.manual_seed(0)
    for sh in shapes:
        run(*sh)
