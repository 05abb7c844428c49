"""CPU restatement of the reference's UNETR baseline generator (`model_name: unet_lora`).
Test infrastructure only.

Reference: src/generators/unet.py -- `Unet` :13-92 (ViTPyramidEncoder + Decoder + per-marker SegmentationHead),
`ViTPyramidEncoder` :116-171 (timm `forward_intermediates(indices, norm=False, output_fmt='NCHW')` at four depths, LoRA on q/v),
`ViTFeatureUpsampler` :174-236 (conv stem on the image + nearest re-grid + Deconv2DBlock pyramids), `Decoder` :288-404,
`Conv2DBlock` :441-474 (conv3x3 with bias -> BatchNorm -> ReLU), `Deconv2DBlock` :477-519 (ConvTranspose2d k2 s2 -> conv3x3 ->
BatchNorm -> ReLU -> Dropout), `initialize_decoder_head` :522-531.  The reference trains this baseline with `model.dropout: 0.1`
(configs/model/unet.yaml:2): nn.Dropout behind every block's ReLU and timm DropPath in the ViT blocks.  Both are random; the
restatement takes the masks as inputs (`drop=` hooks) so that a test can feed it the masks the HIP kernels used.
State-dict keys follow the reference modules (encoder.model.* = the timm ViT, encoder.feature_upsampler.*, decoder.*,
segmentation_head_<i>.*).
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F

from .decoder import _bn, segmentation_head
from .vit import ViTConfig, vit_block, vit_embed, vit_state_shapes


def extract_layers(depth: int) -> list:
    """unet.py:131-137"""
    if depth == 4:
        return [0, 1, 2, 3]
    if depth > 4:
        return np.round(np.linspace(depth // 4, depth - 1, 4)).astype(int).tolist()
    raise ValueError("Vit Should have a depth higher than 3")


def pyramid_dims(embed_dim: int):
    """(skip_dim_11, skip_dim_12, bottleneck_dim) and encoder out_channels, unet.py:179-187,214-220"""
    s11, s12, bott = (256, 128, 312) if embed_dim < 512 else (512, 256, 512)
    return s11, s12, bott, [64, 128, 256, bott, embed_dim]


def _conv_block_shapes(s, pre, cin, cout):        # Conv2DBlock: block.0 conv (bias), block.1 BN
    s[pre + "block.0.weight"] = (cout, cin, 3, 3)
    s[pre + "block.0.bias"] = (cout,)
    _bn_shapes(s, pre + "block.1.", cout)


def _deconv_block_shapes(s, pre, cin, cout):      # Deconv2DBlock: block.0 convT, block.1 conv (bias), block.2 BN
    s[pre + "block.0.weight"] = (cin, cout, 2, 2)
    s[pre + "block.0.bias"] = (cout,)
    s[pre + "block.1.weight"] = (cout, cout, 3, 3)
    s[pre + "block.1.bias"] = (cout,)
    _bn_shapes(s, pre + "block.2.", cout)


def _bn_shapes(s, pre, c):
    s[pre + "weight"], s[pre + "bias"] = (c,), (c,)
    s[pre + "running_mean"], s[pre + "running_var"], s[pre + "num_batches_tracked"] = (c,), (c,), ()


def unetr_state_shapes(cfg: ViTConfig, img: int, nc_out: int, lora: bool = True) -> dict:
    D = cfg.dim
    s11, s12, bott, oc = pyramid_dims(D)
    s = dict(vit_state_shapes(cfg, img, prefix="encoder.model.", lora=lora))
    up = "encoder.feature_upsampler."
    _conv_block_shapes(s, up + "convsteam.0.", 3, 32)
    _conv_block_shapes(s, up + "convsteam.1.", 32, 64)
    # nn.Upsample / nn.Identity occupies Sequential slot 0 of every upsampler, the Deconv2DBlocks follow at 1, 2, 3
    for name, chain in (("upsampler0", [(D, s11), (s11, s12), (s12, 128)]), ("upsampler1", [(D, s11), (s11, 256)]),
                        ("upsampler2", [(D, bott)])):
        for k, (ci, co) in enumerate(chain):
            _deconv_block_shapes(s, f"{up}{name}.{k + 1}.", ci, co)
    d = "decoder."
    s[d + "bottleneck_upsampler.weight"], s[d + "bottleneck_upsampler.bias"] = (D, bott, 2, 2), (bott,)
    _conv_block_shapes(s, d + "decoder3_upsampler.0.", 2 * bott, bott)
    _conv_block_shapes(s, d + "decoder3_upsampler.1.", bott, bott)
    _conv_block_shapes(s, d + "decoder3_upsampler.2.", bott, bott)
    s[d + "decoder3_upsampler.3.weight"], s[d + "decoder3_upsampler.3.bias"] = (bott, 256, 2, 2), (256,)
    _conv_block_shapes(s, d + "decoder2_upsampler.0.", 512, 256)
    _conv_block_shapes(s, d + "decoder2_upsampler.1.", 256, 256)
    s[d + "decoder2_upsampler.2.weight"], s[d + "decoder2_upsampler.2.bias"] = (256, 128, 2, 2), (128,)
    _conv_block_shapes(s, d + "decoder1_upsampler.0.", 256, 128)
    _conv_block_shapes(s, d + "decoder1_upsampler.1.", 128, 128)
    s[d + "decoder1_upsampler.2.weight"], s[d + "decoder1_upsampler.2.bias"] = (128, 64, 2, 2), (64,)
    _conv_block_shapes(s, d + "decoder0_header.0.", 128, 64)
    _conv_block_shapes(s, d + "decoder0_header.1.", 64, 64)
    s[d + "decoder0_header.2.weight"], s[d + "decoder0_header.2.bias"] = (32, 64, 1, 1), (32,)
    for h in range(nc_out):
        b = f"segmentation_head_{h}."
        s[b + "0.psi.0.weight"], s[b + "0.psi.0.bias"] = (16, 32, 1, 1), (16,)
        _bn_shapes(s, b + "0.psi.1.", 16)
        s[b + "0.psi.3.weight"], s[b + "0.psi.3.bias"] = (1, 16, 1, 1), (1,)
        s[b + "1.weight"], s[b + "1.bias"] = (1, 32, 3, 3), (1,)
    return s


_DROP = None   # test hook: callable(block prefix, activation NCHW) -> multiplier tensor (nn.Dropout mask / (1-p)), or None


def _dropout(pre, y):
    return y if _DROP is None else y * _DROP(pre, y)


def conv_block(p, pre, x, training, new_stats=None):
    x = F.conv2d(x, p[pre + "block.0.weight"], p[pre + "block.0.bias"], padding=1)
    return _dropout(pre, F.relu(_bn(p, pre + "block.1.", x, training, new_stats)))


def deconv_block(p, pre, x, training, new_stats=None):
    x = F.conv_transpose2d(x, p[pre + "block.0.weight"], p[pre + "block.0.bias"], stride=2)
    x = F.conv2d(x, p[pre + "block.1.weight"], p[pre + "block.1.bias"], padding=1)
    return _dropout(pre, F.relu(_bn(p, pre + "block.2.", x, training, new_stats)))


def vit_intermediates(p, x, cfg: ViTConfig, prefix: str, lora: bool, drop_path=None):
    """timm 1.0.15 VisionTransformer.forward_intermediates(indices=extract_layers, norm=False, output_fmt='NCHW',
    intermediates_only=True): block outputs without the prefix tokens, as [B, D, g, g]."""
    t = vit_embed(p, x, cfg, prefix)
    take, outs = extract_layers(cfg.depth), []
    B, g = x.shape[0], x.shape[-1] // cfg.patch
    for i in range(cfg.depth):
        t = vit_block(p, f"{prefix}blocks.{i}.", t, cfg, lora, None if drop_path is None else drop_path[i])
        if i in take:
            outs.append(t[:, cfg.num_prefix:].reshape(B, g, g, cfg.dim).permute(0, 3, 1, 2))
    return outs


def feature_upsampler(p, x, feats, cfg: ViTConfig, img: int, training, new_stats=None, prefix="encoder.feature_upsampler."):
    g = img // cfg.patch
    sf = None if cfg.patch == 16 else int(img / 16) / int(img / cfg.patch)           # unet.py:146-150

    def regrid(f):
        return f if sf is None else F.interpolate(f, scale_factor=sf, mode="nearest")  # nn.Upsample(scale_factor, 'nearest')

    z0 = conv_block(p, prefix + "convsteam.1.", conv_block(p, prefix + "convsteam.0.", x, training, new_stats), training, new_stats)
    f0 = regrid(feats[0])
    for k in (1, 2, 3):
        f0 = deconv_block(p, f"{prefix}upsampler0.{k}.", f0, training, new_stats)
    f1 = regrid(feats[1])
    for k in (1, 2):
        f1 = deconv_block(p, f"{prefix}upsampler1.{k}.", f1, training, new_stats)
    f2 = deconv_block(p, f"{prefix}upsampler2.1.", regrid(feats[2]), training, new_stats)
    f3 = regrid(feats[3])
    return [z0, f0, f1, f2, f3]


def unetr_decoder(p, z, training, new_stats=None, prefix="decoder."):
    z0, z1, z2, z3, z4 = z
    ct = lambda x, pre: F.conv_transpose2d(x, p[pre + "weight"], p[pre + "bias"], stride=2)
    b4 = ct(z4, prefix + "bottleneck_upsampler.")
    h = torch.cat([z3, b4], 1)
    for k in (0, 1, 2):
        h = conv_block(p, f"{prefix}decoder3_upsampler.{k}.", h, training, new_stats)
    b3 = ct(h, prefix + "decoder3_upsampler.3.")
    h = torch.cat([z2, b3], 1)
    for k in (0, 1):
        h = conv_block(p, f"{prefix}decoder2_upsampler.{k}.", h, training, new_stats)
    b2 = ct(h, prefix + "decoder2_upsampler.2.")
    h = torch.cat([z1, b2], 1)
    for k in (0, 1):
        h = conv_block(p, f"{prefix}decoder1_upsampler.{k}.", h, training, new_stats)
    b1 = ct(h, prefix + "decoder1_upsampler.2.")
    h = torch.cat([z0, b1], 1)
    for k in (0, 1):
        h = conv_block(p, f"{prefix}decoder0_header.{k}.", h, training, new_stats)
    return F.conv2d(h, p[prefix + "decoder0_header.2.weight"], p[prefix + "decoder0_header.2.bias"])


def unetr_forward(p: dict, x: torch.Tensor, cfg: ViTConfig, nc_out: int, training: bool = False, lora: bool = True,
                  new_stats: dict | None = None, drop=None, drop_path=None):
    """Unet.forward (unet.py:83-92) with Tanh heads.  drop: callable(block prefix, activation) -> dropout multipliers;
    drop_path: [L, 2, B] per-sample DropPath factors (both None = eval mode / rate 0)."""
    global _DROP
    _DROP = drop
    try:
        return _unetr_forward(p, x, cfg, nc_out, training, lora, new_stats, drop_path)
    finally:
        _DROP = None


def _unetr_forward(p, x, cfg, nc_out, training, lora, new_stats, drop_path):
    img = x.shape[-1]
    feats = vit_intermediates(p, x, cfg, "encoder.model.", lora, drop_path)
    z = feature_upsampler(p, x, feats, cfg, img, training, new_stats)
    f = unetr_decoder(p, z, training, new_stats)
    return torch.cat([segmentation_head(p, f"segmentation_head_{h}.", f, training, new_stats) for h in range(nc_out)], 1)
