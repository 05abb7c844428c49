"""Component fixtures from the REFERENCE's own classes (build container only; see make_golden.py for the import machinery):
  comp_lora_qkv.npz : QkvWithLoRA forward + gradients w.r.t. x, A_q, B_q, A_v, B_v (src/generators/lora.py:8-33)
  comp_wmse.npz     : WeightedMSELoss value + gradient w.r.t. the prediction (src/loss.py:47-57)
  comp_heads.npz    : NH x SegmentationHead(32, 1, use_attention=True, Tanh) in train mode on one shared feature map
                      (src/generators/unet.py:407-438, used at mipheivit.py:198-218): outputs, BN running stats, gradients
                      w.r.t. the feature map and every head parameter
Inputs and weights come from oracle.detgen (regenerated identically on the GPU box); the fixtures hold outputs only.
Usage:  python oracle/make_golden_components.py
"""
from __future__ import annotations

import os
import sys

import numpy as np
import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle.detgen import det_normal  # noqa: E402
from oracle.make_golden import load_reference  # noqa: E402
from oracle.model import orion_marker_weights  # noqa: E402


def T(seed, name, shape, std=1.0):
    return torch.from_numpy(np.asarray(det_normal(seed, name, shape, 0.0, std), dtype=np.float32))


def main():
    VT, refgen, refsrc = load_reference()
    out_dir = os.path.join(ROOT, "tests", "golden")
    # ---- LoRA-adapted fused qkv projection
    seed, B, N, D, r, alpha = 31, 2, 37, 96, 8, 1.0
    lin = nn.Linear(D, 3 * D)
    qkv = sys.modules["refgen.lora"].QkvWithLoRA(lin, rank=r, alpha=alpha)
    with torch.no_grad():
        lin.weight.copy_(T(seed, "w", (3 * D, D), D ** -0.5))
        lin.bias.copy_(T(seed, "b", (3 * D,), 0.02))
        qkv.lora_q.A.copy_(T(seed, "Aq", (D, r), r ** -0.5))
        qkv.lora_q.B.copy_(T(seed, "Bq", (r, D), 0.05))
        qkv.lora_v.A.copy_(T(seed, "Av", (D, r), r ** -0.5))
        qkv.lora_v.B.copy_(T(seed, "Bv", (r, D), 0.05))
    x = T(seed, "x", (B, N, D)).requires_grad_(True)
    up = T(seed, "up", (B, N, 3 * D))
    out = qkv(x)
    (out * up).sum().backward()
    np.savez_compressed(os.path.join(out_dir, "comp_lora_qkv.npz"), seed=seed, B=B, N=N, D=D, rank=r, alpha=alpha,
                        out=out.detach().numpy(), dx=x.grad.numpy(), dAq=qkv.lora_q.A.grad.numpy(),
                        dBq=qkv.lora_q.B.grad.numpy(), dAv=qkv.lora_v.A.grad.numpy(), dBv=qkv.lora_v.B.grad.numpy())
    # ---- weighted MSE
    seed, B, C, H, W = 32, 3, 16, 24, 40
    w = orion_marker_weights(C)
    lossf = sys.modules["refsrc.loss"].WeightedMSELoss(50, w)
    pred = torch.tanh(T(seed, "pred", (B, C, H, W))).requires_grad_(True)
    target = T(seed, "target", (B, C, H, W), 0.5).clamp(-0.9, 0.9)
    loss = lossf(y_true=target, y_pred=pred)
    loss.backward()
    np.savez_compressed(os.path.join(out_dir, "comp_wmse.npz"), seed=seed, B=B, C=C, H=H, W=W, lambda_factor=50.0,
                        loss=float(loss), dpred=pred.grad.numpy())
    # ---- the per-marker heads
    seed, NH, B, H, W = 33, 5, 2, 24, 40
    Head = sys.modules["refgen.unet"].SegmentationHead
    heads = [Head(32, 1, kernel_size=3, activation=nn.Tanh(), use_attention=True) for _ in range(NH)]
    with torch.no_grad():
        for i, hd in enumerate(heads):
            hd[0].psi[0].weight.copy_(T(seed, f"W1_{i}", (16, 32, 1, 1), 0.3))
            hd[0].psi[0].bias.copy_(T(seed, f"b1_{i}", (16,), 0.2))
            hd[0].psi[1].weight.copy_(1 + T(seed, f"g_{i}", (16,), 0.3))
            hd[0].psi[1].bias.copy_(T(seed, f"be_{i}", (16,), 0.3))
            hd[0].psi[3].weight.copy_(T(seed, f"W2_{i}", (1, 16, 1, 1), 0.5))
            hd[0].psi[3].bias.copy_(T(seed, f"b2_{i}", (1,), 0.2))
            hd[1].weight.copy_(T(seed, f"W3_{i}", (1, 32, 3, 3), 0.1))
            hd[1].bias.copy_(T(seed, f"b3_{i}", (1,), 0.1))
            hd.train()
    x = T(seed, "x", (B, 32, H, W)).to(torch.bfloat16).float().requires_grad_(True)   # bf16-representable feature map
    up = T(seed, "up", (B, NH, H, W))
    out = torch.cat([hd(x) for hd in heads], dim=1)
    (out * up).sum().backward()
    rec = dict(seed=seed, NH=NH, B=B, H=H, W=W, out=out.detach().numpy(), dx=x.grad.numpy())
    for i, hd in enumerate(heads):
        rec[f"rm_{i}"] = hd[0].psi[1].running_mean.numpy()
        rec[f"rv_{i}"] = hd[0].psi[1].running_var.numpy()
        for nm, prm in (("dW1", hd[0].psi[0].weight), ("db1", hd[0].psi[0].bias), ("dg", hd[0].psi[1].weight),
                        ("dbe", hd[0].psi[1].bias), ("dW2", hd[0].psi[3].weight), ("db2", hd[0].psi[3].bias),
                        ("dW3", hd[1].weight), ("db3", hd[1].bias)):
            rec[f"{nm}_{i}"] = prm.grad.numpy()
    np.savez_compressed(os.path.join(out_dir, "comp_heads.npz"), **rec)
    print("wrote comp_lora_qkv.npz, comp_wmse.npz, comp_heads.npz; loss", float(loss))


if __name__ == "__main__":
    main()
