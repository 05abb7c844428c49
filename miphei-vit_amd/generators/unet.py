"""UNETR baseline generator (`model_name: unet*`, reference src/generators/unet.py) as parameter containers with the
reference's module tree and state-dict keys; the arithmetic runs in the HIP engine (`engine_unetr.UnetrEngine`).

Scope: the ViT-pyramid variant used with the pathology foundation models (`ViTPyramidEncoder` + `ViTFeatureUpsampler` +
`Decoder` + per-marker `SegmentationHead`, unet.py:13-92,116-236,288-404).  The ResNet / Swin / smp encoders of the reference's
`Unet` are outside the MI355X path and raise NotImplementedError.
"""
from __future__ import annotations

import numpy as np
import torch.nn as nn

from .foundation_models import FOUNDATION_MODEL_REGISTRY, VisionTransformer
from .lora import apply_lora
from .mipheivit import SegmentationHead, initialize_decoder_head


class Conv2DBlock(nn.Module):
    """conv3x3 (bias) -> BatchNorm -> ReLU -> Dropout (unet.py:441-474)"""

    def __init__(self, in_channels, out_channels, kernel_size=3, dropout=0.):
        super().__init__()
        if kernel_size != 3:
            raise NotImplementedError("3x3 blocks only")
        self.block = nn.Sequential(nn.Conv2d(in_channels, out_channels, 3, stride=1, padding=1), nn.BatchNorm2d(out_channels),
                                   nn.ReLU(), nn.Dropout(dropout))
        self.out_channels = out_channels


class Deconv2DBlock(nn.Module):
    """ConvTranspose2d(k2, s2) -> conv3x3 (bias) -> BatchNorm -> ReLU -> Dropout (unet.py:477-519)"""

    def __init__(self, in_channels, out_channels, kernel_size=3, dropout=0.):
        super().__init__()
        self.block = nn.Sequential(nn.ConvTranspose2d(in_channels, out_channels, 2, stride=2, padding=0, output_padding=0),
                                   nn.Conv2d(out_channels, out_channels, 3, stride=1, padding=1), nn.BatchNorm2d(out_channels),
                                   nn.ReLU(), nn.Dropout(dropout))
        self.out_channels = out_channels


class ViTFeatureUpsampler(nn.Module):
    def __init__(self, embed_dim, drop_rate, scale_factor=None):
        super().__init__()
        self.embed_dim, self.drop_rate, self.scale_factor = embed_dim, drop_rate, scale_factor
        if embed_dim < 512:
            self.skip_dim_11, self.skip_dim_12, self.bottleneck_dim = 256, 128, 312
        else:
            self.skip_dim_11, self.skip_dim_12, self.bottleneck_dim = 512, 256, 512
        first = (lambda: nn.Upsample(scale_factor=scale_factor, mode="nearest")) if scale_factor else nn.Identity
        self.convsteam = nn.Sequential(Conv2DBlock(3, 32, 3, dropout=drop_rate), Conv2DBlock(32, 64, 3, dropout=drop_rate))
        self.upsampler0 = nn.Sequential(first(), Deconv2DBlock(embed_dim, self.skip_dim_11, dropout=drop_rate),
                                        Deconv2DBlock(self.skip_dim_11, self.skip_dim_12, dropout=drop_rate),
                                        Deconv2DBlock(self.skip_dim_12, 128, dropout=drop_rate))
        self.upsampler1 = nn.Sequential(first(), Deconv2DBlock(embed_dim, self.skip_dim_11, dropout=drop_rate),
                                        Deconv2DBlock(self.skip_dim_11, 256, dropout=drop_rate))
        self.upsampler2 = nn.Sequential(first(), Deconv2DBlock(embed_dim, self.bottleneck_dim, dropout=drop_rate))
        self.upsampler3 = nn.Sequential(first())
        self.out_channels = [64, 128, 256, self.bottleneck_dim, embed_dim]
        initialize_decoder_head(self)


class ViTPyramidEncoder(nn.Module):
    def __init__(self, img_size, encoder_name, ckpt_path=None, drop_path_rate=0., use_lora=False, pretrained=True):
        super().__init__()
        try:
            model = FOUNDATION_MODEL_REGISTRY[encoder_name](img_size, pretrained=pretrained, ckpt_path=ckpt_path,
                                                            drop_path_rate=drop_path_rate)
        except KeyError:
            raise NotImplementedError(f"Unknown model: try ones in {list(FOUNDATION_MODEL_REGISTRY.keys())}")
        if not isinstance(model, VisionTransformer):
            raise ValueError(f"Model should be a VisionTransformer or SwinTransformer, got {type(model)}")
        self.model = model
        if use_lora:
            apply_lora(self.model, rank=8, alpha=1.)
        depth = len(model.blocks)
        if depth == 4:
            self.extract_layers = [0, 1, 2, 3]
        elif depth > 4:
            self.extract_layers = np.round(np.linspace(depth // 4, depth - 1, 4)).astype(int).tolist()
        else:
            raise ValueError("Vit Should have a depth higher than 3")
        self.patch_size = 16
        self.drop_rate = drop_path_rate
        assert img_size % self.patch_size == 0
        real_patch_size = self.model.patch_embed.patch_size[0]
        scale_factor = int(img_size / 16) / int(img_size / real_patch_size) if real_patch_size != 16 else None
        self.feature_upsampler = ViTFeatureUpsampler(self.model.embed_dim, scale_factor=scale_factor, drop_rate=self.drop_rate)
        self.out_channels = self.feature_upsampler.out_channels


class Decoder(nn.Module):
    def __init__(self, encoder_out_channels, out_channels=32, drop_rate=0.):
        super().__init__()
        if len(encoder_out_channels) != 5:
            raise ValueError(f"Encoder should return 5 features, got {len(encoder_out_channels)}")
        embed_dim, bott = encoder_out_channels[-1], encoder_out_channels[3]
        d2, d3, d4 = encoder_out_channels[2], encoder_out_channels[1], encoder_out_channels[0]
        self.drop_rate = drop_rate
        ct = lambda ci, co: nn.ConvTranspose2d(ci, co, kernel_size=2, stride=2, padding=0, output_padding=0)
        cb = lambda ci, co: Conv2DBlock(ci, co, dropout=drop_rate)
        self.bottleneck_upsampler = ct(embed_dim, bott)
        self.decoder3_upsampler = nn.Sequential(cb(bott * 2, bott), cb(bott, bott), cb(bott, bott), ct(bott, d2))
        self.decoder2_upsampler = nn.Sequential(cb(d2 * 2, d2), cb(d2, d2), ct(d2, d3))
        self.decoder1_upsampler = nn.Sequential(cb(d3 * 2, d3), cb(d3, d3), ct(d3, d4))
        self.decoder0_header = nn.Sequential(cb(d4 * 2, d4), cb(d4, d4), nn.Conv2d(d4, out_channels, kernel_size=1))
        initialize_decoder_head(self)


class Unet(nn.Module):
    """reference `Unet` (unet.py:13-92) for the foundation-model ViT encoders"""

    def __init__(self, img_size, encoder_name, encoder_weights=None, decoder_out_channels=32, head_use_attention=True,
                 use_lora=False, classes=1, activation=None, drop_rate: float = 0, pretrained=True):
        super().__init__()
        if encoder_name not in FOUNDATION_MODEL_REGISTRY:
            raise NotImplementedError(f"encoder '{encoder_name}': only the ViT registry encoders run on the MI355X path")
        if not 0.0 <= float(drop_rate) < 1.0:
            raise ValueError("drop_rate must be in [0, 1)")
        if decoder_out_channels != 32 or not head_use_attention or classes > 16:
            raise NotImplementedError("heads: 32-channel decoder output, attention heads, at most 16 markers")
        self.encoder = ViTPyramidEncoder(img_size, encoder_name, ckpt_path=encoder_weights, drop_path_rate=drop_rate,
                                         use_lora=use_lora, pretrained=pretrained)
        self.decoder = Decoder(self.encoder.out_channels, out_channels=decoder_out_channels, drop_rate=drop_rate)
        self.num_heads = classes
        for idx in range(self.num_heads):
            setattr(self, f"segmentation_head_{idx}", SegmentationHead(in_channels=decoder_out_channels, out_channels=1,
                                                                        activation=nn.Tanh(), kernel_size=3, use_attention=True))
        self.initialize()
        from ..engine_unetr import UnetrEngine
        object.__setattr__(self, "_engine", UnetrEngine(self))
        self._register_load_state_dict_pre_hook(lambda *a, **k: self._engine.invalidate())
        self.register_load_state_dict_post_hook(lambda *a, **k: self._engine.invalidate())

    def initialize(self):
        initialize_decoder_head(self.decoder)
        initialize_decoder_head(self.encoder.feature_upsampler)
        for idx in range(self.num_heads):
            initialize_decoder_head(getattr(self, f"segmentation_head_{idx}"))

    def freeze_encoder(self):
        for _, p in self.encoder.named_parameters():
            p.requires_grad = False
        for _, p in self.encoder.feature_upsampler.named_parameters():
            p.requires_grad = True

    def unfreeze_encoder(self):
        for p in self.encoder.parameters():
            p.requires_grad = True

    def forward(self, x):
        return self._engine.forward(x)

    def _apply(self, fn, *a, **k):
        out = super()._apply(fn, *a, **k)
        eng = self.__dict__.get("_engine")
        if eng is not None:
            eng.invalidate()
        return out
