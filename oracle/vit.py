"""timm-1.0.15 VisionTransformer arithmetic, restated (test infrastructure).

The reference builds its encoder with ``timm.create_model`` (
``/root/reference/src/generators/foundation_models.py:50-57``); timm is an
un-vendored dependency pinned ``timm==1.0.15``
(``/root/reference/requirements.txt:17``) and is absent here, so its published
algorithm is restated from SURVEY.md App. A and cross-checked against Hugging
Face ``Dinov2WithRegistersModel`` (``tests/test_oracle_golden.py``).
LoRA follows ``/root/reference/src/generators/lora.py:8-33``.
"""

from __future__ import annotations

from dataclasses import dataclass

import torch
import torch.nn.functional as F


@dataclass(frozen=True)
class ViTConfig:
    patch: int = 14
    dim: int = 1536
    depth: int = 40
    heads: int = 24
    mlp: str = "swiglu"  # "swiglu" (SwiGLUPacked, SiLU) | "gelu" (Mlp, erf GELU)
    hidden: int = 8192  # fc1 out_features (packed a|b for swiglu)
    reg_tokens: int = 4
    ln_eps: float = 1e-6
    lora_rank: int = 8
    lora_alpha: float = 1.0

    @property
    def num_prefix(self) -> int:
        return 1 + self.reg_tokens

    @property
    def fc2_in(self) -> int:
        return self.hidden // 2 if self.mlp == "swiglu" else self.hidden

    def grid(self, img: int) -> int:
        return img // self.patch

    def tokens(self, img: int) -> int:
        return self.grid(img) ** 2 + self.num_prefix


VIT_CONFIGS = {
    # H-Optimus-0: vit_giant_patch14_reg4_dinov2 (foundation_models.py:53-57)
    "hoptimus0": ViTConfig(),
    # BASELINE.json config 1 "Tiny-ViT (2 layers, 64-d)": patch16 D64 L2 H4 GELU
    "tiny": ViTConfig(patch=16, dim=64, depth=2, heads=4, mlp="gelu", hidden=256, reg_tokens=4),
    # small SwiGLU/patch-14 config exercising every H-Optimus-0 code path
    "tiny_swiglu": ViTConfig(patch=14, dim=96, depth=2, heads=3, mlp="swiglu", hidden=512, reg_tokens=4),
    # depth-4 variants for the UNETR baseline (ViTPyramidEncoder needs >= 4 blocks, unet.py:131-137)
    "tiny4": ViTConfig(patch=16, dim=64, depth=4, heads=4, mlp="gelu", hidden=256, reg_tokens=4),
    "tiny4_swiglu": ViTConfig(patch=14, dim=96, depth=4, heads=3, mlp="swiglu", hidden=512, reg_tokens=4),
}


def vit_state_shapes(cfg: ViTConfig, img: int, prefix: str = "encoder.vit.", lora: bool = True) -> dict:
    """State-dict key -> shape."""
    D, g = cfg.dim, cfg.grid(img)
    s = {
        "cls_token": (1, 1, D),
        "reg_token": (1, cfg.reg_tokens, D),
        "pos_embed": (1, g * g, D),
        "patch_embed.proj.weight": (D, 3, cfg.patch, cfg.patch),
        "patch_embed.proj.bias": (D,),
        "norm.weight": (D,),
        "norm.bias": (D,),
    }
    for i in range(cfg.depth):
        b = f"blocks.{i}."
        qkv = "attn.qkv.qkv." if lora else "attn.qkv."
        s[b + "norm1.weight"] = (D,)
        s[b + "norm1.bias"] = (D,)
        s[b + qkv + "weight"] = (3 * D, D)
        s[b + qkv + "bias"] = (3 * D,)
        if lora:
            for w in ("lora_q", "lora_v"):
                s[b + f"attn.qkv.{w}.A"] = (D, cfg.lora_rank)
                s[b + f"attn.qkv.{w}.B"] = (cfg.lora_rank, D)
        s[b + "attn.proj.weight"] = (D, D)
        s[b + "attn.proj.bias"] = (D,)
        s[b + "ls1.gamma"] = (D,)
        s[b + "norm2.weight"] = (D,)
        s[b + "norm2.bias"] = (D,)
        s[b + "mlp.fc1.weight"] = (cfg.hidden, D)
        s[b + "mlp.fc1.bias"] = (cfg.hidden,)
        s[b + "mlp.fc2.weight"] = (D, cfg.fc2_in)
        s[b + "mlp.fc2.bias"] = (D,)
        s[b + "ls2.gamma"] = (D,)
    return {prefix + k: v for k, v in s.items()}


def lora_qkv(h: torch.Tensor, w: torch.Tensor, b: torch.Tensor, Aq: torch.Tensor, Bq: torch.Tensor, Av: torch.Tensor,
             Bv: torch.Tensor, alpha: float) -> torch.Tensor:
    """QkvWithLoRA.forward (lora.py:29-33): fused qkv projection, alpha*(h A B) added on the q (first D) and v (last D) columns."""
    D = h.shape[-1]
    qkv = F.linear(h, w, b)
    dq = alpha * (h @ Aq @ Bq)
    dv = alpha * (h @ Av @ Bv)
    return torch.cat([qkv[..., :D] + dq, qkv[..., D : 2 * D], qkv[..., 2 * D :] + dv], dim=-1)


def vit_block(p: dict, pre: str, x: torch.Tensor, cfg: ViTConfig, lora: bool, drop_path=None) -> torch.Tensor:
    """One timm Block: x + dp1(ls1*attn(norm1 x)); x + dp2(ls2*mlp(norm2 x))  (App. A).  drop_path: None (eval / rate 0) or a [2, B]
    tensor of the per-sample DropPath factors (0 or 1/keep) of the attention and MLP branch (timm.layers.DropPath, train mode)."""
    B, N, D = x.shape
    H, Dh = cfg.heads, D // cfg.heads
    h = F.layer_norm(x, (D,), p[pre + "norm1.weight"], p[pre + "norm1.bias"], cfg.ln_eps)
    if lora:
        qkv = lora_qkv(h, p[pre + "attn.qkv.qkv.weight"], p[pre + "attn.qkv.qkv.bias"], p[pre + "attn.qkv.lora_q.A"],
                       p[pre + "attn.qkv.lora_q.B"], p[pre + "attn.qkv.lora_v.A"], p[pre + "attn.qkv.lora_v.B"], cfg.lora_alpha)
    else:
        qkv = F.linear(h, p[pre + "attn.qkv.weight"], p[pre + "attn.qkv.bias"])
    qkv = qkv.reshape(B, N, 3, H, Dh).permute(2, 0, 3, 1, 4)
    q, k, v = qkv.unbind(0)
    att = (q @ k.transpose(-2, -1)) * (Dh ** -0.5)
    att = att.softmax(dim=-1)
    o = (att @ v).transpose(1, 2).reshape(B, N, D)
    o = F.linear(o, p[pre + "attn.proj.weight"], p[pre + "attn.proj.bias"])
    o = p[pre + "ls1.gamma"] * o
    x = x + (o if drop_path is None else o * drop_path[0].view(B, 1, 1))
    h = F.layer_norm(x, (D,), p[pre + "norm2.weight"], p[pre + "norm2.bias"], cfg.ln_eps)
    u = F.linear(h, p[pre + "mlp.fc1.weight"], p[pre + "mlp.fc1.bias"])
    if cfg.mlp == "swiglu":
        a, b = u.chunk(2, dim=-1)  # SwiGLUPacked, gate_last=False: silu(x1) * x2
        g = F.silu(a) * b
    else:
        g = F.gelu(u)  # erf form
    y = F.linear(g, p[pre + "mlp.fc2.weight"], p[pre + "mlp.fc2.bias"])
    y = p[pre + "ls2.gamma"] * y
    return x + (y if drop_path is None else y * drop_path[1].view(B, 1, 1))


def vit_embed(p: dict, x: torch.Tensor, cfg: ViTConfig, prefix: str) -> torch.Tensor:
    """PatchEmbed + _pos_embed with no_embed_class=True (App. A)."""
    B = x.shape[0]
    t = F.conv2d(x, p[prefix + "patch_embed.proj.weight"], p[prefix + "patch_embed.proj.bias"], stride=cfg.patch)
    t = t.flatten(2).transpose(1, 2)  # [B, g*g, D]; trailing img % patch pixels dropped
    t = t + p[prefix + "pos_embed"]
    cls = p[prefix + "cls_token"].expand(B, -1, -1)
    reg = p[prefix + "reg_token"].expand(B, -1, -1)
    return torch.cat([cls, reg, t], dim=1)


def vit_forward(p: dict, x: torch.Tensor, cfg: ViTConfig, prefix: str = "encoder.vit.", lora: bool = True,
                return_blocks: bool = False):
    """VisionTransformer.forward with global_pool='' / num_classes=0 -> [B, N, D]."""
    t = vit_embed(p, x, cfg, prefix)
    mids = [t]
    for i in range(cfg.depth):
        t = vit_block(p, f"{prefix}blocks.{i}.", t, cfg, lora)
        mids.append(t)
    D = cfg.dim
    out = F.layer_norm(t, (D,), p[prefix + "norm.weight"], p[prefix + "norm.bias"], cfg.ln_eps)
    return (out, mids) if return_blocks else out
