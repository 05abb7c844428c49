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
    def __init__(self, in_dim, out_dim, rank, alpha):
        super().__init__()
        std_dev = 1 / torch.sqrt(torch.tensor(rank).float())
        self.A = nn.Parameter(torch.randn(in_dim, rank) * std_dev)
        self.B = nn.Parameter(torch.zeros(rank, out_dim))
        self.alpha = alpha
        self.rank = rank


class QkvWithLoRA(nn.Module):
    def __init__(self, qkv, rank, alpha):
        super().__init__()
        self.qkv = qkv
        self.dim = qkv.in_features
        self.lora_q = LoRALayer(self.dim, self.dim, rank, alpha)
        self.lora_v = LoRALayer(self.dim, self.dim, rank, alpha)

    @property
    def in_features(self):
        return self.qkv.in_features

    @property
    def out_features(self):
        return self.qkv.out_features


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
