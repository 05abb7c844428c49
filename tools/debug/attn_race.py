"""Is mvit_attention_fwd run-to-run identical?  Same packed qkv, many launches, every output compared bit for bit with the first."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from miphei_vit_amd import _lib
if os.environ.get("MIPHEI_LIB"):          # a variant build of the library (make BUILD=... LIB=... EXTRA=-D...)
    _lib.LIB_PATH = os.path.abspath(os.environ["MIPHEI_LIB"])
import miphei_vit_amd.ops as ops

B, N, H, Dh = 16, 329, 24, 64
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
with_res = int(sys.argv[2]) if len(sys.argv) > 2 else 1
busy = int(sys.argv[3]) if len(sys.argv) > 3 else 0       # 1: a GEMM between the launches (LDS / cache state as in the model)
g = torch.Generator(device="cuda").manual_seed(1)
qkv = (torch.randn(B, N, 3, H, Dh, generator=g, device="cuda") * 1.0).bfloat16()
scale = Dh ** -0.5
ref = torch.empty(B, N, H * Dh, device="cuda", dtype=torch.bfloat16)
lse0 = torch.empty(B, H, N, device="cuda")
res0 = torch.empty_like(ref) if with_res else None
ops.attention_fwd(qkv, ref, lse0, B, N, H, Dh, scale, out_res=res0)
a = torch.randn(5264, 1536, device="cuda").bfloat16()
w = torch.randn(1536, 1536, device="cuda").bfloat16()
c = torch.empty(5264, 1536, device="cuda", dtype=torch.bfloat16)
bad = 0
for it in range(iters):
    out = torch.full_like(ref, 3.0)
    lse = torch.empty_like(lse0)
    res = torch.empty_like(ref) if with_res else None
    if busy:
        ops.gemm(a, w, c)
    ops.attention_fwd(qkv, out, lse, B, N, H, Dh, scale, out_res=res)
    if not torch.equal(out, ref) or not torch.equal(lse, lse0):
        bad += 1
        d = (out.float() - ref.float()).abs()
        idx = (d > 0).nonzero()
        bs, ns, cs = idx[:, 0].unique().tolist(), idx[:, 1].unique().tolist(), (idx[:, 2] // Dh).unique().tolist()
        dl = (lse - lse0).abs()
        if bad <= 12:
            dims = sorted(set((idx[:, 2] % Dh).tolist()))
            hb, hh = bs[0], cs[0]
            dd = d[hb, :, hh * Dh:(hh + 1) * Dh]
            rows_bad = (dd > 0).any(1).nonzero().flatten().tolist()
            r0 = rows_bad[0]
            print("   dims that differ:", dims, "| per-row count of first bad row", int((dd[r0] > 0).sum()), "| |d| by dim (row %d):" % r0,
                  [round(float(v), 3) for v in dd[r0].tolist()][:64])
            print(f"iter {it}: {idx.shape[0]} elements differ, max |d| {float(d.max()):.4g} (ref max {float(ref.float().abs().max()):.3g}); batches {bs[:6]} "
                  f"heads {cs[:8]} rows {ns[:12]}{'...' if len(ns) > 12 else ''} ({len(ns)} rows); lse diffs {int((dl > 0).sum())} max {float(dl.max()):.3g}")
print(f"{bad} of {iters} forward launches differ from the first")
# backward: dq / dk / dv and the D row sums, same protocol
dO = torch.randn(B, N, H * Dh, generator=g, device="cuda").bfloat16()
dq0, ds0 = torch.zeros_like(qkv), torch.empty(B, H, N, device="cuda")
ops.attention_bwd(qkv, ref, dO, lse0, ds0, dq0, B, N, H, Dh, scale, out_res=res0)
badb = 0
for it in range(iters // 2):
    dq, ds = torch.zeros_like(qkv), torch.empty_like(ds0)
    if busy:
        ops.gemm(a, w, c)
    ops.attention_bwd(qkv, ref, dO, lse0, ds, dq, B, N, H, Dh, scale, out_res=res0)
    if not torch.equal(dq, dq0) or not torch.equal(ds, ds0):
        badb += 1
        d = (dq.float() - dq0.float()).abs()
        idx = (d > 0).nonzero()
        if badb <= 8:
            print(f"bwd iter {it}: {idx.shape[0]} elements differ, max |d| {float(d.max()):.4g}; batches {idx[:, 0].unique().tolist()[:4]} which(q/k/v) "
                  f"{idx[:, 2].unique().tolist()} heads {idx[:, 3].unique().tolist()[:6]} rows {len(idx[:, 1].unique())} dims {sorted(set(idx[:, 4].tolist()))[:8]}.. "
                  f"dsum diffs {int((ds != ds0).sum())}")
print(f"{badb} of {iters // 2} backward launches differ from the first")
