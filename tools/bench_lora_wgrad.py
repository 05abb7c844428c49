"""Micro-benchmark of the batched LoRA weight-gradient products of one group of ViT blocks (engine.py _vit_backward): dB = t^T [dq | . | dv]
and dA = [dt_q | dt_v]^T h on the TN MFMA GEMM, at the training shapes (M = 16 * 329, D = 1536, rank 8 -- LORA_RANK=16 for the wider case -- 10 blocks per launch).
Prints us per launch and the operand bytes streamed per second; argv: msplit values to scan for the two launches (default: the engine's)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from miphei_vit_amd import _lib
if os.environ.get("MIPHEI_LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["MIPHEI_LIB"])
import miphei_vit_amd.ops as ops

bf = torch.bfloat16
M, D, r, n = 16 * 329, 1536, int(os.environ.get("LORA_RANK", "8")), int(os.environ.get("LORA_GROUP", "10"))   # rank 8, 10 blocks per launch = the training configuration


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


# 4 groups of operands so that consecutive launches do not find their operands in the Infinity Cache (as in the step)
NG = 4 if n <= 10 else 2
t = [torch.randn(n, M, 2 * r, device="cuda").to(bf) for _ in range(NG)]
dqkv = [torch.randn(n, M, 3 * D, device="cuda").to(bf) for _ in range(NG)]
h1 = [torch.randn(n, M, D, device="cuda").to(bf) for _ in range(NG)]
dBq, dBv, dAq, dAv = (torch.zeros(n, 4 * r * D, device="cuda") for _ in range(4))
lsplit = max(1, min(512 // ((D + 127) // 128), (M + 255) // 256))
gs0 = max(1, min(lsplit, -(-1024 // (n * 2 * ((D + 127) // 128)))))
ga0 = max(1, min(lsplit, 768 // (n * ((D + 127) // 128))))           # the engine's slice count of the dA launch
scan = [int(v) for v in sys.argv[1:]]
splits = scan or [gs0]
k = [0]


def dB(ms):
    g = k[0] % NG; k[0] += 1
    ops.gemm_tn(t[g], dqkv[g], dBq, M=M, I=2 * r, J=3 * D, lda=2 * r, ldb=3 * D, ldci=D, ldcj=1, msplit=ms, c2=dBv, isplit=r, j1=D,
                jlo2=2 * D, batch=n, stride_a=M * 2 * r, stride_b=M * 3 * D, stride_c=4 * r * D)


def dA(ms):
    g = k[0] % NG; k[0] += 1
    ops.gemm_tn(t[g], h1[g], dAq, M=M, I=2 * r, J=D, lda=2 * r, ldb=D, ldci=1, ldcj=r, msplit=ms, c2=dAv, isplit=r, batch=n,
                stride_a=M * 2 * r, stride_b=M * D, stride_c=4 * r * D)


for ms in splits:
    u = timeit(lambda: dB(ms))
    print(f"dB  msplit {ms:3d}: {u:7.1f} us  {n * M * 2 * D * 2 / u / 1e6:5.2f} TB/s")
for ms in ([2 * v for v in scan] or [ga0]):
    u = timeit(lambda: dA(ms))
    print(f"dA  msplit {ms:3d}: {u:7.1f} us  {n * M * D * 2 / u / 1e6:5.2f} TB/s")
