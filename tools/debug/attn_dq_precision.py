"""Where the attention backward loses precision: HIP kernel vs fp32 autograd vs an fp32 emulation of the flash algorithm with the
kernel's bf16 rounding points (P, dS as bf16 MFMA operands; D = rowsum(dO * O) from the bf16 O), and vs the unfused bf16-autocast
sequence of the reference (softmax in fp32).  Logit scale sweeps from near-uniform to peaky attention."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import miphei_vit_amd.ops as ops

B, N, H, Dh = 2, 329, 24, 64
scale = Dh ** -0.5
rel = lambda a, b: float((a.double() - b.double()).norm() / b.double().norm())
bfr = lambda t: t.bfloat16().float()
for amp in (0.25, 0.5, 1.0, 2.0):
    g = torch.Generator(device="cuda").manual_seed(7)
    qkv = (torch.randn(B, N, 3, H, Dh, generator=g, device="cuda") * amp).bfloat16()
    dO = (torch.randn(B, N, H * Dh, generator=g, device="cuda") * 0.1).bfloat16()
    out = torch.empty(B, N, H * Dh, device="cuda", dtype=torch.bfloat16)
    lse = torch.empty(B, H, N, device="cuda")
    ops.attention_fwd(qkv, out, lse, B, N, H, Dh, scale)
    dqkv = torch.zeros_like(qkv)
    dsum = torch.empty(B, H, N, device="cuda")
    ops.attention_bwd(qkv, out, dO, lse, dsum, dqkv, B, N, H, Dh, scale)
    x = qkv.double().requires_grad_(True)
    q, k, v = x.permute(2, 0, 3, 1, 4).unbind(0)
    s = (q @ k.transpose(-1, -2)) * scale
    P = s.softmax(-1)
    ref = (P @ v).transpose(1, 2).reshape(B, N, H * Dh)
    ref.backward(dO.double())
    gq, gk, gv = (x.grad[:, :, i] for i in range(3))
    # emulation of the flash backward in fp32 with bf16 rounding points
    qf, kf, vf = (t.float() for t in qkv.permute(2, 0, 3, 1, 4).unbind(0))
    dOf = dO.float().view(B, N, H, Dh).transpose(1, 2)
    Of = out.float().view(B, N, H, Dh).transpose(1, 2)
    Pf = ((qf @ kf.transpose(-1, -2)) * scale).softmax(-1)
    dP = dOf @ vf.transpose(-1, -2)
    for name, Dterm in (("D from bf16 O", (dOf * Of).sum(-1, keepdim=True)), ("D = sum P dP (f32)", (Pf * dP).sum(-1, keepdim=True))):
        dS = bfr(Pf * (dP - Dterm))
        dq_e = (dS @ kf * scale).transpose(1, 2)
        dk_e = (dS.transpose(-1, -2) @ qf * scale).transpose(1, 2)
        print(f"  amp {amp}: emulated flash [{name}]: dq {rel(dq_e, gq):.4f} dk {rel(dk_e, gk):.4f}")
    # the reference's unfused autocast sequence: bf16 matmuls, fp32 softmax
    xa = qkv.float().requires_grad_(True)
    qa, ka, va = xa.permute(2, 0, 3, 1, 4).unbind(0)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        att = (qa @ ka.transpose(-2, -1)) * scale
        att = att.softmax(dim=-1)
        oa = (att @ va).transpose(1, 2).reshape(B, N, H * Dh)
    oa.backward(dO.to(oa.dtype))
    print(f"amp {amp}: HIP dq {rel(dqkv[:, :, 0].float(), gq):.4f} dk {rel(dqkv[:, :, 1].float(), gk):.4f} dv {rel(dqkv[:, :, 2].float(), gv):.4f} | "
          f"autocast dq {rel(xa.grad[:, :, 0], gq):.4f} dk {rel(xa.grad[:, :, 1], gk):.4f} dv {rel(xa.grad[:, :, 2], gv):.4f}", flush=True)
