"""Where a wave of attn_fwd_kernel / attn_bwd_dq_kernel / attn_bwd_dkv_kernel spends its cycles (measurement build: make DEBUG_KNOBS=1 BUILD=build_t LIB=../libmiphei_t.so
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


# ---- the two backward kernels (round 5): same stamps, written behind the D values of the `dsum` buffer; [5] = end, [6] = start of the wave
dO = torch.randn(B, N, H * Dh, device="cuda").bfloat16()
dqkv = torch.empty_like(qkv)
dsum = torch.zeros(B * H * N + 2 * nblk * 4 * 8 * 2 + 64, device="cuda")
for _ in range(3):
    ops.attention_bwd(qkv, out, dO, lse, dsum, dqkv, B, N, H, Dh, Dh ** -0.5, out_res=res)
torch.cuda.synchronize()
allp = dsum[B * H * N:B * H * N + 2 * nblk * 4 * 8 * 2].view(torch.int64).view(2, nblk, 4, 8).cpu().double()
for kname, prof in (("attn_bwd_dq_kernel", allp[0]), ("attn_bwd_dkv_kernel", allp[1])):
    live = prof[:, :, 4] > 0
    tot = prof[:, :, :4].sum(-1)
    print(f"\n{kname}: {nblk} blocks x 4 waves; cycles per wave, steps per wave {prof[:, :, 4][live].mean():.1f}")
    for k, n in enumerate(["launch -> first step", "waits at step tops (own DMA + barrier)", "step work", "epilogue (stores, drain)"]):
        v = prof[:, :, k][live]
        print(f"  {n:42s} {v.mean():9.0f}  ({100 * v.mean() / tot[live].mean():4.1f} %)   p10 {v.quantile(0.1):8.0f}  p90 {v.quantile(0.9):8.0f}")
    print(f"  total per wave {tot[live].mean():.0f} cycles; per step: wait {prof[:, :, 1][live].sum() / prof[:, :, 4][live].sum():.0f}, work {prof[:, :, 2][live].sum() / prof[:, :, 4][live].sum():.0f}")
    # rounds, per XCD (the cycle counter is per XCD): blocks that start after the first block of their XCD has ended
    st, en, xcd = prof[:, 0, 6], prof[:, :, 5].max(1).values, prof[:, 0, 7]
    for x in range(8):
        m = xcd == x
        if m.sum() == 0:
            continue
        s0 = st[m].min()
        late = (st[m] > en[m].min()).sum()
        print(f"  XCD {x}: {int(m.sum())} blocks, span {en[m].max() - s0:8.0f} cycles, first block ends at {en[m].min() - s0:8.0f}, {int(late)} blocks start later than that; "
              f"start quantiles {[int(v) for v in (st[m] - s0).quantile(torch.tensor([0.5, 0.67, 0.75, 0.9, 1.0], dtype=torch.double))]}")
