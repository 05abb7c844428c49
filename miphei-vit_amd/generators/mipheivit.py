"""MIPHEI-ViT generator: ViT encoder wrapper + ViTMatte-style decoder, as parameter containers.

Same classes, constructor arguments, attributes and state-dict keys as the reference
(``/root/reference/src/generators/mipheivit.py``: ``Basic_Conv3x3`` :20-41, ``ConvStream`` :44-73,
``Fusion_Block`` :76-93, ``ViTMatte`` :96-121, ``Encoder`` :124-163, ``Detail_Capture`` :166-220,
``get_vitmatte`` :224-233; heads from ``src/generators/unet.py:407-438``, init ``:522-531``).
``ViTMatte.forward`` runs the MI355X HIP engine (``engine.py``); the sub-modules only own parameters.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from .foundation_models import FOUNDATION_MODEL_REGISTRY, VisionTransformer
from .lora import apply_lora


class Basic_Conv3x3(nn.Module):
    def __init__(self, in_chans, out_chans, stride=2, padding=1):
        super().__init__()
        self.conv = nn.Conv2d(in_chans, out_chans, 3, stride, padding, bias=False)
        self.bn = nn.BatchNorm2d(out_chans)
        self.relu = nn.ReLU(True)


class ConvStream(nn.Module):
    def __init__(self, in_chans=4, out_chans=[48, 96, 192]):
        super().__init__()
        self.convs = nn.ModuleList()
        self.conv_chans = [in_chans] + list(out_chans)
        for i in range(len(self.conv_chans) - 1):
            self.convs.append(Basic_Conv3x3(self.conv_chans[i], self.conv_chans[i + 1]))


class Fusion_Block(nn.Module):
    def __init__(self, in_chans, out_chans):
        super().__init__()
        self.conv = Basic_Conv3x3(in_chans, out_chans, stride=1, padding=1)


class AttentionBlock(nn.Module):
    def __init__(self, F_l, F_int):
        super().__init__()
        self.psi = nn.Sequential(
            nn.Conv2d(F_l, F_int, kernel_size=1, stride=1, padding=0, bias=True),
            nn.BatchNorm2d(F_int),
            nn.ReLU(inplace=True),
            nn.Conv2d(F_int, 1, kernel_size=1, stride=1, padding=0, bias=True),
            nn.Sigmoid())


class SegmentationHead(nn.Sequential):
    def __init__(self, in_channels, out_channels, kernel_size=3, activation=None, use_attention=False):
        layers = []
        if use_attention:
            layers.append(AttentionBlock(in_channels, in_channels // 2))
        layers.append(nn.Conv2d(in_channels, out_channels, kernel_size=kernel_size, padding=kernel_size // 2))
        layers.append(activation if activation is not None else nn.Identity())
        super().__init__(*layers)


def initialize_decoder_head(module):
    for m in module.modules():
        if isinstance(m, (nn.Conv2d, nn.ConvTranspose2d)):
            nn.init.normal_(m.weight, 0.0, 0.02)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)
        elif isinstance(m, nn.BatchNorm2d):
            nn.init.normal_(m.weight, 1.0, 0.02)
            nn.init.constant_(m.bias, 0)


class Detail_Capture(nn.Module):
    def __init__(self, emb_chans, in_chans=3, out_chans=1, convstream_out=[48, 96, 192], fusion_out=[256, 128, 64, 32],
                 use_attention=True, activation=torch.nn.Identity()):
        super().__init__()
        assert len(fusion_out) == len(convstream_out) + 1
        if not use_attention or not isinstance(activation, nn.Tanh):
            raise NotImplementedError("the HIP head kernel implements the MIPHEI-ViT configuration: attention heads + Tanh")
        self.convstream = ConvStream(in_chans=in_chans)
        self.conv_chans = self.convstream.conv_chans
        self.num_heads = out_chans
        self.fusion_blks = nn.ModuleList()
        self.fus_channs = [emb_chans] + list(fusion_out)
        for i in range(len(self.fus_channs) - 1):
            self.fusion_blks.append(Fusion_Block(self.fus_channs[i] + self.conv_chans[-(i + 1)], self.fus_channs[i + 1]))
        for idx in range(self.num_heads):
            setattr(self, f"segmentation_head_{idx}",
                    SegmentationHead(fusion_out[-1], 1, kernel_size=3, activation=activation, use_attention=use_attention))


class Encoder(nn.Module):
    def __init__(self, vit):
        super().__init__()
        if not isinstance(vit, VisionTransformer):
            raise ValueError(f"Model should be a VisionTransformer or SwinTransformer, got {type(vit)}")
        self.vit = vit
        self.is_swint = False
        self.grid_size = self.vit.patch_embed.grid_size
        self.num_prefix_tokens = self.vit.num_prefix_tokens
        self.embed_dim = self.vit.embed_dim
        patch_size = self.vit.patch_embed.patch_size
        img_size = self.vit.patch_embed.img_size
        assert img_size[0] % 16 == 0
        assert img_size[1] % 16 == 0
        if patch_size != (16, 16):
            target = (int(img_size[0] / 16), int(img_size[1] / 16))
            self.scale_factor = (target[0] / self.grid_size[0], target[1] / self.grid_size[1])
            self.interpolate = True
        else:
            self.scale_factor = None          # as the reference (mipheivit.py:150-157): patch-16 encoders need no re-grid
            self.interpolate = False


class ViTMatte(nn.Module):
    def __init__(self, encoder, decoder):
        super().__init__()
        self.encoder = encoder
        self.decoder = decoder
        self.initialize()
        from ..engine import HipEngine
        object.__setattr__(self, "_engine", HipEngine(self))
        object.__setattr__(self.encoder.vit, "_engine_owner", self._engine)
        self._register_load_state_dict_pre_hook(lambda *a, **k: self._engine.invalidate())
        self.register_load_state_dict_post_hook(lambda *a, **k: self._engine.invalidate())

    def forward(self, x):
        return self._engine.forward_autograd(x)

    def initialize(self):
        initialize_decoder_head(self.decoder)

    def set_input_size(self, img_size):
        if any((s & (s - 1)) != 0 or s == 0 for s in img_size):
            raise ValueError("Both height and width in img_size must be powers of 2")
        if any(s < 128 for s in img_size):
            raise ValueError("Height and width must be greater or equal to 128")
        self.encoder.vit.set_input_size(img_size=img_size)
        self.encoder.grid_size = self.encoder.vit.patch_embed.grid_size
        self._engine.invalidate()

    def _apply(self, fn, *a, **k):
        out = super()._apply(fn, *a, **k)
        eng = self.__dict__.get("_engine")
        if eng is not None:
            eng.invalidate()
        return out


def get_vitmatte(encoder_name, img_size, num_classes, use_lora=False, ckpt_path=None, drop_path_rate=0, pretrained=True):
    vit = FOUNDATION_MODEL_REGISTRY[encoder_name](
        img_size, pretrained=pretrained, ckpt_path=ckpt_path, drop_path_rate=drop_path_rate, global_pool="")
    if use_lora:
        vit = apply_lora(vit, rank=8, alpha=1.)
    encoder = Encoder(vit)
    decoder = Detail_Capture(emb_chans=encoder.embed_dim, out_chans=num_classes, use_attention=True,
                             activation=nn.Tanh())
    return ViTMatte(encoder=encoder, decoder=decoder)
