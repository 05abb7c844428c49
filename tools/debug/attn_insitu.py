"""In-situ precision of the attention backward inside the full-size model: at blocks 39 / 20 / 0 of a real backward pass, compare the
kernel's d(qkv) with an fp64 recomputation from the same saved qkv and the same dO, and with fp32 emulations of the flash algorithm."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from miphei_vit_amd.generators import get_vitmatte
from miphei_vit_amd.loss import WeightedMSELoss
from oracle.model import orion_marker_weights

nc, B, img = 16, 2, 256
dev = torch.device("cuda:0")
with torch.device(dev):
    model = get_vitmatte("hoptimus0", img, nc, use_lora=True, pretrained=False)
bench.synthetic_init_(model, seed=3)
x, y = bench.synthetic_batch(77, B, img, nc, dev)
eng = model._engine
eng.lora_group = 1
model.train()
out = eng.forward(x, train=True)
w = eng._saved.w
_, dY = eng.loss_and_grad(out, y, orion_marker_weights(nc).to(dev), 50.0)
grab = {}


def hook(l):
    if l in (39, 20, 0):
        grab[l] = (w.do.clone(), w.dqkv_all[l].clone())


eng.backward(dY, on_lora_block_done=hook)
c = eng._config()
N, H, Dh = c.ntok, c.H, c.Dh
scale = Dh ** -0.5
rel = lambda a, b: float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-300))
for l in (39, 20, 0):
    dO, dqkv = grab[l]
    qkv = w.qkv[l].view(B, N, 3, H, Dh)
    o = w.o[l]
    ores = w.ores[l]
    xx = qkv.double().requires_grad_(True)
    q, k, v = xx.permute(2, 0, 3, 1, 4).unbind(0)
    s = (q @ k.transpose(-1, -2)) * scale
    P = s.softmax(-1)
    ref = (P @ v).transpose(1, 2).reshape(B, N, H * Dh)
    ref.backward(dO.view(B, N, H * Dh).double())
    g = xx.grad
    d = dqkv.view(B, N, 3, H, Dh)
    ent = float(-(P * (P + 1e-300).log()).sum(-1).mean())
    print(f"block {l}: mean attention entropy {ent:.3f} nats (uniform = {torch.log(torch.tensor(float(N))):.3f}); "
          f"|dq| {float(g[:, :, 0].norm()):.3e} |dk| {float(g[:, :, 1].norm()):.3e} |dv| {float(g[:, :, 2].norm()):.3e}")
    print(f"   HIP rel err: dq {rel(d[:, :, 0], g[:, :, 0]):.4f} dk {rel(d[:, :, 1], g[:, :, 1]):.4f} dv {rel(d[:, :, 2], g[:, :, 2]):.4f}; "
          f"O rel err vs fp64 {rel(o.view(B, N, H * Dh), ref.detach()):.4f} (O + residual: {rel(o.float().view(B, N, H * Dh) + ores.float().view(B, N, H * Dh), ref.detach()):.2e})")
    qf, kf, vf = (t.float() for t in qkv.permute(2, 0, 3, 1, 4).unbind(0))
    dOf = dO.float().view(B, N, H, Dh).transpose(1, 2)
    Of = o.float().view(B, N, H, Dh).transpose(1, 2)
    Pf = ((qf @ kf.transpose(-1, -2)) * scale).softmax(-1)
    dP = dOf @ vf.transpose(-1, -2)
    bfr = lambda t: t.bfloat16().float()
    for name, Dt in (("D from bf16 O", (dOf * Of).sum(-1, keepdim=True)), ("D = sum P dP", (Pf * dP).sum(-1, keepdim=True))):
        dS = bfr(Pf * (dP - Dt))
        dq_e = (dS @ kf * scale).transpose(1, 2)
        dk_e = (dS.transpose(-1, -2) @ qf * scale).transpose(1, 2)
        print(f"   emulated flash [{name}]: dq {rel(dq_e, g[:, :, 0]):.4f} dk {rel(dk_e, g[:, :, 1]):.4f}")
    # the explicit autocast sequence of the CPU oracle (softmax in fp32) on the same tensors
    xa = qkv.float().requires_grad_(True)
    qa, ka, va = xa.permute(2, 0, 3, 1, 4).unbind(0)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        att = ((qa @ ka.transpose(-2, -1)) * scale).softmax(dim=-1)
        oa = (att @ va).transpose(1, 2).reshape(B, N, H * Dh)
    oa.backward(dO.view(B, N, H * Dh).to(oa.dtype))
    print(f"   explicit autocast: dq {rel(xa.grad[:, :, 0], g[:, :, 0]):.4f} dk {rel(xa.grad[:, :, 1], g[:, :, 1]):.4f} dv {rel(xa.grad[:, :, 2], g[:, :, 2]):.4f}", flush=True)
