"""LoRA adapters on the packed qkv projection (host-side parameter containers).

Mirrors the surface of the reference's ``src/generators/lora.py`` (``LoRALayer`` :8-18,
``QkvWithLoRA`` :21-33, ``apply_lora`` :48-83): same attribute names, parameter shapes and state-dict
keys (``...attn.qkv.qkv.weight``, ``...attn.qkv.lora_q.A`` ...).  The arithmetic
``qkv[..., :D] += alpha*(x@A_q)@B_q ; qkv[..., -D:] += alpha*(x@A_v)@B_v`` is executed by the HIP GEMM as a
rank-2r extension of its K loop (see ``engine.py``), never by these modules' ``forward``.
"""
from __future__ import annotations

import torch
import torch.nn as nn


class LoRALayer(nn.Module):
    """Holds one adapter pair: A [in_dim, rank] ~ N(0, 1/rank), B [rank, out_dim] = 0 (so the adapter starts as a no-op)."""

    def __init__(self, in_dim, out_dim, rank, alpha):
        super().__init__()
        self.rank, self.alpha = int(rank), alpha
        self.A = nn.Parameter(torch.randn(in_dim, self.rank) * float(self.rank) ** -0.5)
        self.B = nn.Parameter(torch.zeros(self.rank, out_dim))


class QkvWithLoRA(nn.Module):
    """Wraps the packed qkv ``nn.Linear`` (kept as ``.qkv`` -> state-dict prefix ``attn.qkv.qkv``) with Q and V adapters."""

    def __init__(self, qkv, rank, alpha):
        super().__init__()
        self.dim = qkv.in_features
        self.qkv = qkv
        for name in ("lora_q", "lora_v"):
            setattr(self, name, LoRALayer(self.dim, self.dim, rank, alpha))

    in_features = property(lambda self: self.qkv.in_features)
    out_features = property(lambda self: self.qkv.out_features)


def apply_lora(model, rank, alpha):
    """Wrap every block's ``attn.qkv`` with rank-`rank` adapters on Q and V; freeze everything else."""
    from .foundation_models import VisionTransformer

    if not isinstance(model, VisionTransformer):
        raise ValueError(f"apply_lora expects a VisionTransformer, got {type(model)}")
    for block in model.blocks:
        if not isinstance(block.attn.qkv, QkvWithLoRA):
            dev = block.attn.qkv.weight.device
            block.attn.qkv = QkvWithLoRA(block.attn.qkv, rank, alpha).to(dev)
    for p in model.parameters():
        p.requires_grad = False
    for block in model.blocks:
        for p in list(block.attn.qkv.lora_q.parameters()) + list(block.attn.qkv.lora_v.parameters()):
            p.requires_grad = True
    return model
