"""Generator, loss and training-step restatement (test infrastructure).

Reference: ``ViTMatte.forward`` ``/root/reference/src/generators/mipheivit.py:106-110``;
``WeightedMSELoss`` ``/root/reference/src/loss.py:47-57`` (weights built at
``/root/reference/src/train.py:137-142``); ``pix2pix_lr_scheduler``
``/root/reference/src/utils.py:217-230``; ``ModelModule.training_step``
non-GAN branch ``/root/reference/src/models.py:87-143`` with the optimiser of
``configure_optimizers`` ``:348-371`` (Adam betas (0.5, 0.999), eps 1e-7,
LambdaLR stepped every optimiser step, clip-norm 1.0).
"""

from __future__ import annotations

import math

import numpy as np
import torch

from .decoder import decoder_forward, decoder_state_shapes, encoder_regrid
from .detgen import det_normal, det_uniform
from .vit import ViTConfig, vit_forward, vit_state_shapes

# train.py:137-140 on channel_stats.json for the 16 ORION markers
ORION_MARKER_WEIGHTS = [1.0, 6.9687, 1.4698, 3.5986, 2.4121, 10.5982, 4.4980, 2.7238, 4.5266, 3.0473, 2.8660,
                        3.5367, 1.7173, 3.6613, 1.5315, 2.5265]
# dataset.py:599-601 (H-Optimus-0 mean/std) and channel_stats.json "RGB"
HOPT_MEAN = (0.707223, 0.578729, 0.703617)
HOPT_STD = (0.211883, 0.230117, 0.177517)
RGB_MEAN = (211.1, 194.7, 213.8)
RGB_STD = (30.1, 36.4, 26.4)


def orion_marker_weights(nc: int = 16) -> torch.Tensor:
    return torch.tensor(ORION_MARKER_WEIGHTS[:nc], dtype=torch.float32)


def generator_state_shapes(cfg: ViTConfig, img: int, nc_out: int) -> dict:
    s = vit_state_shapes(cfg, img, "encoder.vit.", lora=True)
    s.update(decoder_state_shapes(cfg.dim, nc_out, 3, "decoder."))
    return s


def generator_forward(p: dict, x: torch.Tensor, cfg: ViTConfig, nc_out: int, training: bool = False,
                      new_stats: dict | None = None, return_mids: bool = False):
    """ViTMatte.forward: decoder(encoder(x), x)."""
    img = x.shape[-1]
    tok = vit_forward(p, x, cfg, "encoder.vit.", lora=True)
    feat = encoder_regrid(tok, cfg.num_prefix, cfg.grid(img), img, cfg.patch)
    out = decoder_forward(p, feat, x, nc_out, training, "decoder.", new_stats, return_mids)
    if return_mids:
        out, mids = out
        mids["tokens"] = tok
        mids["features"] = feat
        return out, mids
    return out


def weighted_mse_loss(y_true: torch.Tensor, y_pred: torch.Tensor, w: torch.Tensor, lambda_factor: float = 50.0):
    """loss.py:54-57: mean_c(mean_{b,h,w}((y_pred-y_true)^2)_c * w_c) * lambda."""
    l = (y_pred - y_true) ** 2
    l = l.mean(dim=(0, 2, 3)) * w
    return l.mean() * lambda_factor


def pix2pix_lr_lambda(step: int, total_iters: int, warmup_iters: int = 400, decay_start_iter: int | None = None):
    """utils.py:217-230."""
    if decay_start_iter is None:
        decay_start_iter = total_iters // 2
    if step < warmup_iters:
        return step / warmup_iters
    if step < decay_start_iter:
        return 1.0
    return max(0.0, 1.0 - (step - decay_start_iter) / (total_iters - decay_start_iter))


def trainable_keys(p: dict) -> list:
    """apply_lora freeze (lora.py:68-83): LoRA A/B + every decoder parameter."""
    keys = []
    for k in p:
        leaf = k.rsplit(".", 1)[-1]
        if k.startswith("encoder."):
            if ".lora_" in k:
                keys.append(k)
        elif leaf in ("weight", "bias"):
            keys.append(k)
    return keys


class OracleTrainer:
    """training_step restated: fwd -> WeightedMSE -> backward -> clip 1.0 -> Adam -> LambdaLR."""

    def __init__(self, p: dict, cfg: ViTConfig, nc_out: int, batch_size: int, total_iters: int,
                 weights: torch.Tensor | None = None, lr_g: float = 2e-4, lambda_factor: float = 50.0):
        self.p = {k: (v.clone() if isinstance(v, torch.Tensor) else torch.as_tensor(v)) for k, v in p.items()}
        self.cfg, self.nc = cfg, nc_out
        self.w = weights if weights is not None else orion_marker_weights(nc_out)
        self.lam = lambda_factor
        self.base_lr = lr_g * math.sqrt(batch_size)  # train.py:163
        self.total = total_iters
        self.keys = trainable_keys(self.p)
        self.m = {k: torch.zeros_like(self.p[k]) for k in self.keys}
        self.v = {k: torch.zeros_like(self.p[k]) for k in self.keys}
        self.step_idx = 0
        self.betas, self.eps = (0.5, 0.999), 1e-7

    def loss_and_grads(self, x, y, new_stats=None):
        leaves = {k: self.p[k].detach().clone().requires_grad_(True) for k in self.keys}
        q = dict(self.p)
        q.update(leaves)
        out = generator_forward(q, x, self.cfg, self.nc, training=True, new_stats=new_stats)
        loss = weighted_mse_loss(y, out, self.w, self.lam)
        grads = torch.autograd.grad(loss, [leaves[k] for k in self.keys])
        return out.detach(), loss.detach(), dict(zip(self.keys, grads))

    def step(self, x, y):
        new_stats = {}
        out, loss, g = self.loss_and_grads(x, y, new_stats)
        total = torch.sqrt(sum((gi.double() ** 2).sum() for gi in g.values())).float()
        coef = torch.clamp(1.0 / (total + 1e-6), max=1.0)  # clip_grad_norm_(max_norm=1.0)
        lr = self.base_lr * pix2pix_lr_lambda(self.step_idx, self.total)
        t = self.step_idx + 1
        b1, b2 = self.betas
        for k in self.keys:
            gi = g[k] * coef
            self.m[k].mul_(b1).add_(gi, alpha=1 - b1)
            self.v[k].mul_(b2).addcmul_(gi, gi, value=1 - b2)
            bc1, bc2 = 1 - b1 ** t, 1 - b2 ** t
            denom = (self.v[k].sqrt() / math.sqrt(bc2)).add_(self.eps)
            self.p[k] = self.p[k] - (lr / bc1) * self.m[k] / denom
        for k, val in new_stats.items():
            self.p[k] = val
        self.step_idx += 1
        return {"loss": float(loss), "grad_norm": float(total), "lr": lr, "out": out, "grads": g}


def synth_batch(seed: int, batch: int, img: int, nc_out: int):
    """Synthetic H&E-like image / mIF-like target batch.

    image: uint8 RGB ~ N(mu, sigma) per channel, clipped, then H-Optimus-0
    normalisation (dataset.py:599-601).  target: uint8 min(255, Exp(20)) ->
    x/255*1.8-0.9 (dataset.py:573)."""
    z = det_normal(seed, "image", (batch, 3, img, img))
    rgb = np.empty_like(z)
    for c in range(3):
        rgb[:, c] = np.clip(np.rint(RGB_MEAN[c] + RGB_STD[c] * z[:, c]), 0, 255)
    mean = np.array(HOPT_MEAN, dtype=np.float32)[None, :, None, None] * 255.0
    std = np.array(HOPT_STD, dtype=np.float32)[None, :, None, None] * 255.0
    image = ((rgb - mean) / std).astype(np.float32)
    u = det_uniform(seed, "target", (batch, nc_out, img, img))
    t8 = np.minimum(255.0, np.floor(-20.0 * np.log(u)))
    target = (t8 / 255.0 * 1.8 - 0.9).astype(np.float32)
    return torch.from_numpy(image), torch.from_numpy(target)
