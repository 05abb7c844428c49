"""Where a wave of attn_fwd_kernel spends its cycles (measurement build: make DEBUG_KNOBS=1 BUILD=build_t LIB=../libmiphei_t.so
EXTRA=-DMVIT_ATTN_TIMING; MIPHEI_LIB=miphei-vit_amd/libmiphei_t.so python tools/debug/attn_timing.py)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from miphei_vit_amd import _lib
if os.environ.get("MIPHEI_LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["MIPHEI_LIB"])
import miphei_vit_amd.ops as ops

B, N, H, Dh = 16, 329, 24, 64
qkv = torch.randn(B, N, 3, H, Dh, device="cuda").bfloat16()
out, res = (torch.empty(B, N, H * Dh, device="cuda", dtype=torch.bfloat16) for _ in range(2))
nblk = ((N + 127) // 128) * B * H
lse = torch.zeros(B * H * N + nblk * 4 * 8 * 2 + 64, device="cuda")        # + 8 longs per wave behind the lse values
for _ in range(3):
    ops.attention_fwd(qkv, out, lse, B, N, H, Dh, Dh ** -0.5, out_res=res)
torch.cuda.synchronize()
prof = lse[B * H * N:B * H * N + nblk * 4 * 8 * 2].view(torch.int64).view(nblk, 4, 8).cpu().double()
live = prof[:, :, 4] > 0
names = ["launch -> first step", "waits at step tops (own DMA + barrier)", "step work", "epilogue (normalise, stores, drain)"]
tot = prof[:, :, :4].sum(-1)
print(f"{nblk} blocks x 4 waves; cycles per wave (mean over waves that ran steps), steps per wave {prof[:, :, 4][live].mean():.1f}")
for k, n in enumerate(names):
    v = prof[:, :, k][live]
    print(f"  {n:42s} {v.mean():9.0f}  ({100 * v.mean() / tot[live].mean():4.1f} %)   p10 {v.quantile(0.1):8.0f}  p90 {v.quantile(0.9):8.0f}")
print(f"  total per wave {tot[live].mean():.0f} cycles; per step: wait {prof[:, :, 1][live].sum() / prof[:, :, 4][live].sum():.0f}, work {prof[:, :, 2][live].sum() / prof[:, :, 4][live].sum():.0f}")
end = prof[:, :, 5]
t0 = end.min() - tot.max()
print(f"  kernel span ~{(end.max() - (end - tot).min()):.0f} cycles; first-round blocks end at p50 {(end[:768 // 1].flatten().quantile(0.5) - (end - tot).min()):.0f}")
