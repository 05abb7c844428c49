"""ViTMatte-style decoder arithmetic, restated (test infrastructure).

Follows ``/root/reference/src/generators/mipheivit.py``:
``Encoder.forward`` :153-163, ``Basic_Conv3x3`` :20-41, ``ConvStream`` :44-73,
``Fusion_Block`` :76-93, ``Detail_Capture.forward`` :207-220, and the heads in
``/root/reference/src/generators/unet.py``: ``AttentionBlock`` :407-422,
``SegmentationHead`` :425-438.
"""

from __future__ import annotations

import torch
import torch.nn.functional as F

CONVSTREAM_OUT = (48, 96, 192)
FUSION_OUT = (256, 128, 64, 32)
BN_EPS = 1e-5
BN_MOMENTUM = 0.1


def decoder_state_shapes(emb: int, nc_out: int, in_chans: int = 3, prefix: str = "decoder.") -> dict:
    s = {}
    chans = (in_chans,) + CONVSTREAM_OUT

    def bn(pre, c):
        s[pre + "weight"] = (c,)
        s[pre + "bias"] = (c,)
        s[pre + "running_mean"] = (c,)
        s[pre + "running_var"] = (c,)
        s[pre + "num_batches_tracked"] = ()

    for i in range(3):
        s[f"convstream.convs.{i}.conv.weight"] = (chans[i + 1], chans[i], 3, 3)
        bn(f"convstream.convs.{i}.bn.", chans[i + 1])
    fus = (emb,) + FUSION_OUT
    for i in range(4):
        cin = fus[i] + chans[-(i + 1)]
        s[f"fusion_blks.{i}.conv.conv.weight"] = (fus[i + 1], cin, 3, 3)
        bn(f"fusion_blks.{i}.conv.bn.", fus[i + 1])
    c = FUSION_OUT[-1]
    for h in range(nc_out):
        b = f"segmentation_head_{h}."
        s[b + "0.psi.0.weight"] = (c // 2, c, 1, 1)
        s[b + "0.psi.0.bias"] = (c // 2,)
        bn(b + "0.psi.1.", c // 2)
        s[b + "0.psi.3.weight"] = (1, c // 2, 1, 1)
        s[b + "0.psi.3.bias"] = (1,)
        s[b + "1.weight"] = (1, c, 3, 3)
        s[b + "1.bias"] = (1,)
    return {prefix + k: v for k, v in s.items()}


def encoder_regrid(tokens: torch.Tensor, num_prefix: int, grid: int, img: int, patch: int) -> torch.Tensor:
    """Encoder.forward tail (mipheivit.py:158-162): drop prefix tokens, view as
    [B, D, g, g] and, when patch != 16, bicubic (A=-0.75, align_corners=False,
    no antialias) regrid by scale (img/16)/g."""
    B, N, D = tokens.shape
    f = tokens[:, num_prefix:].permute(0, 2, 1).reshape(B, D, grid, grid)
    if patch != 16:
        sf = (img / 16) / grid
        f = F.interpolate(f, scale_factor=(sf, sf), mode="bicubic")
    return f


def _bn(p, pre, x, training, new_stats):
    w, b = p[pre + "weight"], p[pre + "bias"]
    rm, rv = p[pre + "running_mean"], p[pre + "running_var"]
    if training:
        dims = (0, 2, 3)
        mean = x.mean(dims)
        var = x.var(dims, unbiased=False)
        n = x.numel() // x.shape[1]
        if new_stats is not None:
            new_stats[pre + "running_mean"] = (1 - BN_MOMENTUM) * rm + BN_MOMENTUM * mean.detach()
            new_stats[pre + "running_var"] = (1 - BN_MOMENTUM) * rv + BN_MOMENTUM * var.detach() * (n / max(n - 1, 1))
            new_stats[pre + "num_batches_tracked"] = p[pre + "num_batches_tracked"] + 1
    else:
        mean, var = rm, rv
    xh = (x - mean[None, :, None, None]) * torch.rsqrt(var[None, :, None, None] + BN_EPS)
    return xh * w[None, :, None, None] + b[None, :, None, None]


def segmentation_head(p: dict, pre: str, f: torch.Tensor, training: bool, new_stats: dict | None = None) -> torch.Tensor:
    """SegmentationHead(32, 1, use_attention=True, Tanh) (unet.py:407-438): y = tanh(conv3x3(f * sigmoid(psi(f))))."""
    t = F.conv2d(f, p[pre + "0.psi.0.weight"], p[pre + "0.psi.0.bias"])
    t = F.relu(_bn(p, pre + "0.psi.1.", t, training, new_stats))
    g = torch.sigmoid(F.conv2d(t, p[pre + "0.psi.3.weight"], p[pre + "0.psi.3.bias"]))
    return torch.tanh(F.conv2d(f * g, p[pre + "1.weight"], p[pre + "1.bias"], padding=1))


def decoder_forward(p: dict, features: torch.Tensor, images: torch.Tensor, nc_out: int, training: bool = False,
                    prefix: str = "decoder.", new_stats: dict | None = None, return_mids: bool = False):
    """Detail_Capture.forward (mipheivit.py:207-220) with Tanh heads."""
    mids = {}
    D = {"D0": images}
    x = images
    for i in range(3):
        pre = f"{prefix}convstream.convs.{i}."
        x = F.conv2d(x, p[pre + "conv.weight"], None, stride=2, padding=1)
        x = F.relu(_bn(p, pre + "bn.", x, training, new_stats))
        D[f"D{i + 1}"] = x
        mids[f"D{i + 1}"] = x
    f = features
    for i in range(4):
        pre = f"{prefix}fusion_blks.{i}.conv."
        up = F.interpolate(f, scale_factor=2, mode="bilinear", align_corners=False)
        cat = torch.cat([D[f"D{3 - i}"], up], dim=1)
        f = F.conv2d(cat, p[pre + "conv.weight"], None, stride=1, padding=1)
        f = F.relu(_bn(p, pre + "bn.", f, training, new_stats))
        mids[f"F{i}"] = f
    outs = []
    for h in range(nc_out):
        outs.append(segmentation_head(p, f"{prefix}segmentation_head_{h}.", f, training, new_stats))
    out = torch.cat(outs, dim=1)
    return (out, mids) if return_mids else out
