"""Does the Infinity-Cache warmer (csrc/standin.hip, ops.prefetch_cache) pay?  Three questions on one box:

 1. residency: dfc2 + d(SwiGLU) (its epilogue re-reads the 86 MB packed pre-activation) with that tensor cold (600 MB swept through
    the caches since it was written), warmed by the warmer, and warm (just read by the same GEMM);
 2. co-residency: a long GEMM on the main stream with the warmer running beside it on a side stream, per pacing value -- what the
    GEMM pays, how long the warmer takes;
 3. the sequence the backward pass would run: [dqkv GEMM || warmer(u)] -> dfc2 + d(SwiGLU), against the same sequence without the warmer.
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from miphei_vit_amd import _lib
if os.environ.get("MIPHEI_LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["MIPHEI_LIB"])
import miphei_vit_amd.ops as ops

M, D, H = 16 * 329, 1536, 4096
dev = "cuda"
torch.manual_seed(0)
dy = torch.randn(M, D, device=dev).bfloat16()
w2t = (torch.randn(H, D, device=dev) * 0.03).bfloat16()
u = torch.randn(M, 2 * H, device=dev).bfloat16()
du = torch.empty(M, 2 * H, device=dev, dtype=torch.bfloat16)
dqkv_a = torch.randn(M, 4608, device=dev).bfloat16()
wq = (torch.randn(D, 4608, device=dev) * 0.02).bfloat16()
dx = torch.empty(M, D, device=dev, dtype=torch.bfloat16)
junk = torch.empty(300 * 1024 * 1024 // 4, device=dev)
side = torch.cuda.Stream()


def dfc2():
    ops.gemm(dy, w2t, du, aux=u, epi=ops.EPI_DSWIGLU)


def dqkv():
    ops.gemm(dqkv_a, wq, dx)


def ev():
    return torch.cuda.Event(enable_timing=True)


def timed(fn, prep, n=8):
    ts = []
    for _ in range(n):
        prep()
        torch.cuda.synchronize()
        e0, e1 = ev(), ev()
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]


def sweep():
    junk.add_(1.0)


def warm_u(pace=0, waves=256):
    ops.prefetch_cache(u, waves=waves, pace=pace)


print("1. residency (dfc2 + d(SwiGLU), us per launch; operands dy / W2t warmed by a bare dfc2-shaped read in every row)")


def prep_cold():
    sweep()
    dy.sum(); w2t.sum()


def prep_warmer():
    sweep()
    dy.sum(); w2t.sum()
    warm_u()


def prep_warm():
    sweep()
    dy.sum(); w2t.sum()
    u.view(torch.int32).sum()


for name, prep in (("u cold", prep_cold), ("u through the warmer", prep_warmer), ("u read by a torch reduction", prep_warm)):
    print(f"   {name:30s} {timed(dfc2, prep):7.1f}")
print(f"   warmer alone (86 MB, pace 0, 256 waves): {timed(lambda: warm_u(), sweep):7.1f} us; 1024 waves: {timed(lambda: warm_u(0, 1024), sweep):7.1f}")

print("2. co-residency: dqkv GEMM (main stream) with the warmer beside it (side stream); us: GEMM alone / GEMM beside / warmer beside")
g_alone = timed(dqkv, lambda: None)
for pace in (0, 4, 8, 16, 32, 64):
    for waves in (256, 512):
        res = []
        for _ in range(6):
            sweep()
            torch.cuda.synchronize()
            e0, e1, s0, s1, go = ev(), ev(), ev(), ev(), torch.cuda.Event()
            go.record()
            side.wait_event(go)
            with torch.cuda.stream(side):
                s0.record()
                warm_u(pace, waves)
                s1.record()
            e0.record()
            dqkv()
            e1.record()
            torch.cuda.synchronize()
            res.append((e0.elapsed_time(e1) * 1e3, s0.elapsed_time(s1) * 1e3))
        res.sort()
        g, w = res[len(res) // 2]
        print(f"   pace {pace:3d} waves {waves:4d}: {g_alone:6.1f} / {g:6.1f} / {w:6.1f}")

print("3. sequence [dqkv || warmer(u)] -> dfc2 + d(SwiGLU): us for the two GEMMs on the main stream")


def seq(pace, waves, use):
    sweep()
    dy.sum(); w2t.sum(); dqkv_a.sum()
    torch.cuda.synchronize()
    e0, e1, go, done = ev(), ev(), torch.cuda.Event(), torch.cuda.Event()
    go.record()
    if use:
        side.wait_event(go)
        with torch.cuda.stream(side):
            warm_u(pace, waves)
            done.record()
    e0.record()
    dqkv()
    dfc2()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3


def med(f, n=8):
    ts = sorted(f() for _ in range(n))
    return ts[len(ts) // 2]


print(f"   without the warmer: {med(lambda: seq(0, 0, False)):7.1f}")
for pace in (0, 4, 8, 16, 32):
    for waves in (256, 512):
        print(f"   pace {pace:3d} waves {waves:4d}: {med(lambda: seq(pace, waves, True)):7.1f}")
