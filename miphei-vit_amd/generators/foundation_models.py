"""Pathology foundation-model encoders as parameter containers with timm's attribute / key layout.

The reference builds its encoder with ``timm.create_model("vit_giant_patch14_reg4_dinov2", ...)``
(``/root/reference/src/generators/foundation_models.py:50-57``); timm is not a dependency here.
``VisionTransformer`` below holds the same parameters under the same state-dict names
and exposes the attributes the reference reads (``patch_embed.{grid_size,patch_size,img_size}``,
``num_prefix_tokens``, ``embed_dim``, ``no_embed_class``, ``blocks[i].attn.qkv``, ``set_input_size``);
its arithmetic is executed by the HIP engine.
"""
from __future__ import annotations

import math
import os

import torch
import torch.nn as nn
import torch.nn.functional as F


class PatchEmbed(nn.Module):
    def __init__(self, img_size, patch_size, in_chans, embed_dim):
        super().__init__()
        self.patch_size = (patch_size, patch_size)
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=patch_size, stride=patch_size, bias=True)
        self.set_input_size(img_size)

    def set_input_size(self, img_size):
        img_size = (img_size, img_size) if isinstance(img_size, int) else tuple(img_size)
        self.img_size = img_size
        self.grid_size = (img_size[0] // self.patch_size[0], img_size[1] // self.patch_size[1])
        self.num_patches = self.grid_size[0] * self.grid_size[1]


class LayerScale(nn.Module):
    def __init__(self, dim, init_values):
        super().__init__()
        self.gamma = nn.Parameter(init_values * torch.ones(dim))


class Attention(nn.Module):
    def __init__(self, dim, num_heads):
        super().__init__()
        self.num_heads = num_heads
        self.head_dim = dim // num_heads
        self.scale = self.head_dim ** -0.5
        self.qkv = nn.Linear(dim, 3 * dim, bias=True)
        self.proj = nn.Linear(dim, dim)


class Mlp(nn.Module):
    """fc1 -> act -> fc2; for SwiGLUPacked fc1 emits [a | b] and fc2 consumes silu(a)*b (hidden/2 wide)."""

    def __init__(self, dim, hidden, swiglu):
        super().__init__()
        self.swiglu = swiglu
        self.fc1 = nn.Linear(dim, hidden)
        self.fc2 = nn.Linear(hidden // 2 if swiglu else hidden, dim)


class Block(nn.Module):
    def __init__(self, dim, num_heads, hidden, swiglu, init_values, eps):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim, eps=eps)
        self.attn = Attention(dim, num_heads)
        self.ls1 = LayerScale(dim, init_values)
        self.norm2 = nn.LayerNorm(dim, eps=eps)
        self.mlp = Mlp(dim, hidden, swiglu)
        self.ls2 = LayerScale(dim, init_values)


class VisionTransformer(nn.Module):
    def __init__(self, img_size=224, patch_size=14, embed_dim=1536, depth=40, num_heads=24, mlp="swiglu", hidden=8192,
                 reg_tokens=4, init_values=1e-5, ln_eps=1e-6, global_pool=""):
        super().__init__()
        if mlp not in ("swiglu", "gelu"):
            raise ValueError(mlp)
        if mlp == "swiglu" and hidden % 2:
            raise ValueError("SwiGLUPacked needs an even hidden size")
        self.embed_dim = self.num_features = embed_dim
        self.num_heads = num_heads
        self.mlp_type = mlp
        self.hidden = hidden
        self.ln_eps = ln_eps
        self.reg_tokens = reg_tokens
        self.num_prefix_tokens = 1 + reg_tokens
        self.no_embed_class = True
        self.global_pool = global_pool
        self.patch_embed = PatchEmbed(img_size, patch_size, 3, embed_dim)
        self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
        self.reg_token = nn.Parameter(torch.zeros(1, reg_tokens, embed_dim)) if reg_tokens else None
        self.pos_embed = nn.Parameter(torch.randn(1, self.patch_embed.num_patches, embed_dim) * 0.02)
        self.blocks = nn.ModuleList(
            [Block(embed_dim, num_heads, hidden, mlp == "swiglu", init_values, ln_eps) for _ in range(depth)])
        self.norm = nn.LayerNorm(embed_dim, eps=ln_eps)
        nn.init.normal_(self.cls_token, std=1e-6)
        if reg_tokens:
            nn.init.normal_(self.reg_token, std=1e-6)

    def set_input_size(self, img_size=None, patch_size=None):
        """Re-grid the position embedding for a new input size (timm resample_abs_pos_embed semantics)."""
        if img_size is None:
            return
        old = self.patch_embed.grid_size
        self.patch_embed.set_input_size(img_size)
        new = self.patch_embed.grid_size
        if new != old:
            self.pos_embed = nn.Parameter(_resample_pos_embed(self.pos_embed.data, old, new),
                                          requires_grad=self.pos_embed.requires_grad)
        eng = getattr(self, "_engine_owner", None)
        if eng is not None:
            eng.invalidate()

    def forward(self, x):
        """timm `forward` with num_classes=0 through the HIP engine: final-norm tokens, pooled per `global_pool`
        ('' -> [B, N, D]; 'token' -> class token [B, D]; 'avg' -> mean of the patch tokens [B, D]).  The embedding
        extraction script of the reference calls the registry model this way with global_pool='token' on fp16 inputs
        (preprocessings/artifacts_detection/extract_embeddings.py:41-42,78); the result keeps the input dtype."""
        from ..engine import encoder_tokens
        tok = encoder_tokens(self, x)
        if self.global_pool == "token":
            tok = tok[:, 0]
        elif self.global_pool == "avg":
            tok = tok[:, self.num_prefix_tokens:].mean(dim=1)
        elif self.global_pool != "":
            raise NotImplementedError(f"global_pool={self.global_pool!r}")
        return tok.to(x.dtype) if x.is_floating_point() else tok


def _resample_pos_embed(posemb, old_grid, new_grid):
    """bicubic + antialias re-grid in fp32 (load-time only; reference: foundation_models.py:198-208)."""
    D = posemb.shape[-1]
    p = posemb.float().reshape(1, old_grid[0], old_grid[1], D).permute(0, 3, 1, 2)
    p = F.interpolate(p, size=new_grid, mode="bicubic", antialias=True)
    return p.permute(0, 2, 3, 1).reshape(1, new_grid[0] * new_grid[1], D).to(posemb.dtype)


def resize_pos_embed_statedict(state_dict, model, img_size):
    if "pos_embed" in state_dict:
        g = model.patch_embed.grid_size
        pe = state_dict["pos_embed"]
        n_old = pe.shape[1]
        side = int(math.sqrt(n_old))
        if side * side != n_old:  # checkpoints that carry a class-token slot
            pe = pe[:, n_old - int(math.sqrt(n_old - 1)) ** 2:]
            side = int(math.sqrt(pe.shape[1]))
        if (side, side) != tuple(g):
            pe = _resample_pos_embed(pe, (side, side), tuple(g))
        state_dict["pos_embed"] = pe
    return state_dict


def _load_checkpoint(model, ckpt_path, img_size):
    if str(ckpt_path).endswith(".safetensors"):
        from safetensors.torch import load_file
        sd = load_file(str(ckpt_path))
    else:
        # tensors only: a foundation-model checkpoint is downloaded data, never unpickle arbitrary objects from it
        sd = torch.load(str(ckpt_path), map_location="cpu", weights_only=True)
    sd = resize_pos_embed_statedict(dict(sd), model, img_size)
    model.load_state_dict(sd)


def _build(img_size, pretrained, ckpt_path, name, drop_path_rate=0., **kw):
    model = VisionTransformer(img_size=img_size, **kw)
    # stochastic depth of timm's blocks (linspace(0, rate, depth) per block, per-sample Bernoulli in train mode): the engine draws
    # the factors per step and applies them inside the residual epilogues (engine._drop_path_factors)
    model.drop_path_rate = float(drop_path_rate or 0.)
    if ckpt_path is not None:
        _load_checkpoint(model, ckpt_path, img_size)
    elif pretrained and os.environ.get("MIPHEI_RANDOM_INIT", "0") != "1":
        _load_checkpoint(model, _hub_checkpoint(name), img_size)
    return model


# hub ids of the pretrained encoders (reference foundation_models.py:13-21; only the encoder on the hot path is kept)
FOUNDATION_HF_CKPT_REGISTRY = {"hoptimus0": "bioptimus/H-optimus-0"}


def _hub_checkpoint(name):
    """Local path of the pretrained weights of registry model `name`, fetched from the Hugging Face hub or its local cache as the
    reference does through timm's `load_state_dict_from_hf(model_id, weights_only=True)` (/root/reference/src/generators/
    foundation_models.py:59-64): `model.safetensors` first, then `pytorch_model.bin`.  Honours HF_HUB_OFFLINE / HF_HOME.  A box without
    network access and without a cached copy fails loudly, naming the ways out -- never silent random weights."""
    repo = FOUNDATION_HF_CKPT_REGISTRY.get(name)
    errors = []
    if repo is not None:
        try:
            from huggingface_hub import hf_hub_download
            for fname in ("model.safetensors", "pytorch_model.bin"):
                try:
                    return hf_hub_download(repo_id=repo, filename=fname)
                except Exception as e:  # noqa: BLE001  (missing file, gated repo, no network: try the next name, then report all)
                    errors.append(f"{fname}: {type(e).__name__}: {str(e).splitlines()[0][:160] if str(e) else ''}")
        except ImportError as e:
            errors.append(f"huggingface_hub: {e}")
    raise RuntimeError(
        f"{name}: no checkpoint given (cfg.model.encoder.encoder_weights) and the pretrained weights could not be fetched from the "
        f"Hugging Face hub ({repo or 'no hub id for this encoder'}; {'; '.join(errors) or 'not attempted'}); pass ckpt_path, put the "
        "repository into the local hub cache (HF_HOME), or use pretrained=False / MIPHEI_RANDOM_INIT=1 for randomly initialised weights")


def hoptimus0(img_size, pretrained=True, ckpt_path=None, drop_path_rate=0., global_pool=""):
    """H-Optimus-0 = vit_giant_patch14_reg4_dinov2 (reference foundation_models.py:50-69)."""
    return _build(img_size, pretrained, ckpt_path, "hoptimus0", patch_size=14, embed_dim=1536, depth=40, num_heads=24,
                  mlp="swiglu", hidden=8192, reg_tokens=4, init_values=1e-5, global_pool=global_pool, drop_path_rate=drop_path_rate)


def tiny_gelu(img_size, pretrained=False, ckpt_path=None, drop_path_rate=0., global_pool=""):
    """BASELINE.json config 1 'Tiny-ViT (2 layers, 64-d)': patch16 D64 L2 H4 GELU (not in the reference registry)."""
    return _build(img_size, False, ckpt_path, "tiny", patch_size=16, embed_dim=64, depth=2, num_heads=4, mlp="gelu",
                  hidden=256, reg_tokens=4, init_values=1e-5, global_pool=global_pool, drop_path_rate=drop_path_rate)


def tiny_swiglu(img_size, pretrained=False, ckpt_path=None, drop_path_rate=0., global_pool=""):
    return _build(img_size, False, ckpt_path, "tiny_swiglu", patch_size=14, embed_dim=96, depth=2, num_heads=3,
                  mlp="swiglu", hidden=512, reg_tokens=4, init_values=1e-5, global_pool=global_pool, drop_path_rate=drop_path_rate)


def tiny4_gelu(img_size, pretrained=False, ckpt_path=None, drop_path_rate=0., global_pool=""):
    """depth-4 variant of `tiny` (the UNETR baseline needs >= 4 blocks)"""
    return _build(img_size, False, ckpt_path, "tiny4", patch_size=16, embed_dim=64, depth=4, num_heads=4, mlp="gelu",
                  hidden=256, reg_tokens=4, init_values=1e-5, global_pool=global_pool, drop_path_rate=drop_path_rate)


def tiny4_swiglu(img_size, pretrained=False, ckpt_path=None, drop_path_rate=0., global_pool=""):
    return _build(img_size, False, ckpt_path, "tiny4_swiglu", patch_size=14, embed_dim=96, depth=4, num_heads=3,
                  mlp="swiglu", hidden=512, reg_tokens=4, init_values=1e-5, global_pool=global_pool, drop_path_rate=drop_path_rate)


FOUNDATION_MODEL_REGISTRY = {
    "hoptimus0": hoptimus0,
    "tiny": tiny_gelu,
    "tiny_swiglu": tiny_swiglu,
    "tiny4": tiny4_gelu,
    "tiny4_swiglu": tiny4_swiglu,
}
