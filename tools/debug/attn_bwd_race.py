"""Is mvit_attention_bwd run-to-run identical?  (tools/debug/step_soak2.py: 2 of 60000 backward launches inside the training step returned a
different dqkv from identical inputs.)  One input, many launches, dqkv and the D row sums compared bit for bit with the first.
  python tools/debug/attn_bwd_race.py [iters=100000] [busy=0]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from miphei_vit_amd import _lib
if os.environ.get("MIPHEI_LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["MIPHEI_LIB"])
import miphei_vit_amd.ops as ops

B, N, H, Dh = 16, 329, 24, 64
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
busy = int(sys.argv[2]) if len(sys.argv) > 2 else 0
g = torch.Generator(device="cuda").manual_seed(1)
qkv = torch.randn(B, N, 3, H, Dh, generator=g, device="cuda").bfloat16()
scale = Dh ** -0.5
out = torch.empty(B, N, H * Dh, device="cuda", dtype=torch.bfloat16)
lse = torch.empty(B, H, N, device="cuda")
res = torch.empty_like(out)
ops.attention_fwd(qkv, out, lse, B, N, H, Dh, scale, out_res=res)
dO = torch.randn(B, N, H * Dh, generator=g, device="cuda").bfloat16()
dq0, ds0 = torch.zeros_like(qkv), torch.empty(B, H, N, device="cuda")
ops.attention_bwd(qkv, out, dO, lse, ds0, dq0, B, N, H, Dh, scale, out_res=res)
a = torch.randn(5264, 1536, device="cuda").bfloat16()
w = torch.randn(1536, 1536, device="cuda").bfloat16()
c = torch.empty(5264, 1536, device="cuda", dtype=torch.bfloat16)
bad = 0
dq, ds = torch.zeros_like(qkv), torch.empty_like(ds0)
for it in range(iters):
    if busy:
        ops.gemm(a, w, c)
    ops.attention_bwd(qkv, out, dO, lse, ds, dq, B, N, H, Dh, scale, out_res=res)
    if not torch.equal(dq, dq0) or not torch.equal(ds, ds0):
        bad += 1
        d = (dq.float() - dq0.float()).abs()
        idx = (d > 0).nonzero()
        if bad <= 10:
            rows = idx[:, 1].unique().tolist()
            print(f"iter {it}: {idx.shape[0]} elements differ, max |d| {float(d.max()):.4g} (max |ref| {float(dq0.float().abs().max()):.3g}); batch "
                  f"{idx[:, 0].unique().tolist()[:4]} which(q/k/v) {idx[:, 2].unique().tolist()} heads {idx[:, 3].unique().tolist()[:6]} rows "
                  f"{rows[:3]}..{rows[-1]} ({len(rows)}) dims {sorted(set(idx[:, 4].tolist()))[:4]}..{max(idx[:, 4].tolist())} "
                  f"({len(set(idx[:, 4].tolist()))}); dsum diffs {int((ds != ds0).sum())}", flush=True)
print(f"{bad} of {iters} backward launches differ from the first")
