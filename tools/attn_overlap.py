"""Measurement (dbg library): the two attention-backward kernels one after the other on one stream against side by side on two
streams.  The dQ kernel (168 VGPRs, 1.5 rounds of blocks) and the dK/dV kernel (245 VGPRs, 2.25 rounds) both end in a sparsely filled
last round; if blocks of the two grids can share the chip, those tails overlap.  The dK/dV kernel reads the D vector the dQ kernel
writes, so a product form needs D from somewhere else; here it is left over from the serial run (same inputs, same values)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from miphei_vit_amd import _lib
_lib.LIB_PATH = _lib.DBG_LIB_PATH
import miphei_vit_amd.ops as ops
L = _lib
B, N, H, Dh = 16, 329, 24, 64
dev = "cuda"
torch.manual_seed(0)
qkv = (torch.randn(B, N, 3, H, Dh, device=dev) * 0.5).bfloat16()
o = torch.empty(B * N, H * Dh, device=dev, dtype=torch.bfloat16)
ores = torch.empty_like(o)
lse = torch.empty(B * H * N + 64, device=dev)
do = torch.randn(B * N, H * Dh, device=dev).bfloat16()
dsum = torch.empty(B * H * N, device=dev)
dqkv = torch.empty_like(qkv)
scale = Dh ** -0.5
ops.attention_fwd(qkv, o, lse, B, N, H, Dh, scale, out_res=ores)
vp, ci, cf = C.c_void_p, C.c_int, C.c_float
fn = getattr(C.CDLL(_lib.LIB_PATH), "mvit_attention_bwd_part")
fn.argtypes = [ci, vp, vp, vp, vp, vp, vp, vp, ci, ci, ci, ci, cf, vp]
fn.restype = ci
P = lambda t: C.c_void_p(t.data_ptr())


def part(which, stream):
    rc = fn(which, P(qkv), P(o), P(ores), P(do), P(lse), P(dsum), P(dqkv), B, N, H, Dh, scale, C.c_void_p(stream.cuda_stream))
    assert rc == 0, rc


main, side = torch.cuda.current_stream(), torch.cuda.Stream()


def ev():
    return torch.cuda.Event(enable_timing=True)


def run(mode, n=30):
    ts = []
    for _ in range(n):
        torch.cuda.synchronize()
        e0, e1, go, done = ev(), ev(), torch.cuda.Event(), torch.cuda.Event()
        e0.record()
        if mode == "serial":
            part(0, main); part(1, main)
        elif mode == "dq":
            part(0, main)
        elif mode == "dkv":
            part(1, main)
        else:
            go.record()
            side.wait_event(go)
            if mode == "dkv_first":
                part(1, side); part(0, main)
            else:
                part(0, side); part(1, main)
            done.record(side)
            main.wait_event(done)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]


ref = None
for mode in ("serial", "dq", "dkv", "dq_side", "dkv_first", "serial"):
    t = run(mode)
    if mode == "serial" and ref is None:
        ref = dqkv.clone()
    same = bool(torch.equal(ref, dqkv))
    print(f"{mode:10s} {t:7.1f} us   (dqkv identical to the serial run: {same})", flush=True)
