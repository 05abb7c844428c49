"""Where the waves of attn_bwd_fused_kernel spend their cycles (measurement build:
    make -C miphei-vit_amd/csrc BUILD=build_tm LIB=variants/libmiphei_tm.so EXTRA=-DMVIT_ATTN_TIMING
    MIPHEI_LIB=miphei-vit_amd/csrc/variants/libmiphei_tm.so python tools/debug/attn_fused_timing.py)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from miphei_vit_amd import _lib
if os.environ.get("MIPHEI_LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["MIPHEI_LIB"])
import miphei_vit_amd.ops as ops

B, N, H, Dh = 16, int(sys.argv[1]) if len(sys.argv) > 1 else 329, 24, 64
qkv = torch.randn(B, N, 3, H, Dh, device="cuda").bfloat16()
out, res = (torch.empty(B, N, H * Dh, device="cuda", dtype=torch.bfloat16) for _ in range(2))
lse = torch.zeros(B * H * N, device="cuda")
ops.attention_fwd(qkv, out, lse, B, N, H, Dh, Dh ** -0.5, out_res=res)
dO = torch.randn(B, N, H * Dh, device="cuda").bfloat16()
dqkv = torch.empty_like(qkv)
nblk, NW = B * H, (N + 47) // 48 + 1
dsum = torch.zeros(B * H * N + nblk * 8 * 16 * 2 + 64, device="cuda")
for _ in range(3):
    ops.attention_bwd(qkv, out, dO, lse, dsum, dqkv, B, N, H, Dh, Dh ** -0.5, out_res=res)
torch.cuda.synchronize()
prof = dsum[B * H * N:B * H * N + nblk * 8 * 16 * 2].view(torch.int64).view(nblk, 8, 16).cpu().double()
key, hlp = prof[:, :NW - 1], prof[:, NW - 1]
print(f"attn_bwd_fused_kernel, N = {N}: {nblk} blocks x ({NW - 1} key waves + 1 helper); cycles per wave, mean over blocks [p10 .. p90]")
def line(name, v):
    v = v.flatten()
    print(f"  {name:58s} {v.mean():9.0f}   [{v.quantile(0.1):8.0f} .. {v.quantile(0.9):8.0f}]")
G0 = (NW + 1) // 2
for k, n in enumerate(["prologue (K DMA, K/V fragments, first tiles) -> first barrier", "X + Y: S, dP, softmax, dS write (all steps)", "Z: dV, dK (all steps)",
                       "barrier waits (all steps)", "epilogue (dK / dV rows through LDS, drain)"]):
    line("key waves: " + n, key[:, :, k])
    if 1 <= k <= 3:
        line("   group 0 (waves 0-3)", key[:, :G0, k])
        line("   group 1", key[:, G0:, k])
line("key waves: total", key[:, :, 9] - key[:, :, 8])
for k, n in ((0, "prologue"), (5, "vmcnt(0): next tile landed"), (1, "D of the next block, tile DMA issue (step 0: K^T fragments)"),
             (2, "W: dQ product (8 tiles per step)"), (3, "barrier waits"), (4, "epilogue")):
    line("helper: " + n, hlp[:, k])
line("helper: total", hlp[:, 9] - hlp[:, 8])
# rounds per XCD (the cycle counter is per XCD)
st, en, xcd = prof[:, 0, 8], prof[:, :NW, 9].max(1).values, prof[:, 0, 10]
for x in range(8):
    m = xcd == x
    if m.sum() == 0:
        continue
    s0 = st[m].min()
    late = st[m] > en[m].min()
    print(f"  XCD {x}: {int(m.sum())} blocks, span {en[m].max() - s0:8.0f} cycles; first round: blocks end at {(en[m][~late] - s0).mean():8.0f} (duration {(en[m][~late] - st[m][~late]).mean():8.0f}); "
          f"{int(late.sum())} later blocks: duration {(en[m][late] - st[m][late]).mean() if late.sum() else 0:8.0f}")
