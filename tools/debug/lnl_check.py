import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
import miphei_vit_amd.ops as ops
for M, D, r in [(5264, 1536, 8), (530, 1536, 8), (21056, 1536, 8)]:
    g = torch.Generator(device="cuda").manual_seed(1)
    x = torch.randn(M, D, generator=g, device="cuda") * 3 + 0.5
    w = torch.randn(D, generator=g, device="cuda") * 0.2 + 1
    b = torch.randn(D, generator=g, device="cuda") * 0.1
    A = (torch.randn(2 * r, D, generator=g, device="cuda") * D ** -0.5).bfloat16()
    h0 = torch.empty(M, D, device="cuda", dtype=torch.bfloat16)
    ops.layernorm_fwd(x, w, b, h0, 1e-6)
    h = torch.zeros_like(h0)
    t = torch.zeros(M, 2 * r, device="cuda", dtype=torch.bfloat16)
    ops.layernorm_lora_fwd(x, w, b, h, A, t, 1e-6)
    d = (h.float() - h0.float()).abs()
    rows = (d > 0).any(1).nonzero().flatten()
    print(M, "rows differing:", rows.numel(), rows[:24].tolist(), "max", float(d.max()), "elements", int((d > 0).sum()))
    if rows.numel():
        r0 = int(rows[0])
        cols = (d[r0] > 0).nonzero().flatten()
        print("  first bad row", r0, "cols", cols[:10].tolist(), "n", cols.numel(), "h", h[r0, cols[:4]].tolist(), "h0", h0[r0, cols[:4]].tolist())
    ref = h0.float() @ A.float().t()
    print("  t rel err", float((t.float() - ref).norm() / ref.norm()))
