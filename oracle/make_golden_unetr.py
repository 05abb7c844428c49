"""UNETR-baseline fixtures from the REFERENCE's own modules (build container only; import machinery of make_golden.py):
ViTFeatureUpsampler + Decoder + SegmentationHead (src/generators/unet.py:174-236,288-404,407-438) driven by ViT intermediates.
The timm ViT is not on disk: the four intermediate feature maps come from the oracle ViT (cross-checked against Hugging Face in
tests/test_oracle_golden.py) and are fed to the reference modules, whose outputs / BatchNorm statistics / gradients are stored.
Usage:  python oracle/make_golden_unetr.py
"""
from __future__ import annotations

import os
import sys

import numpy as np
import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle.detgen import det_state_dict  # noqa: E402
from oracle.make_golden import load_reference  # noqa: E402
from oracle.model import synth_batch  # noqa: E402
from oracle.unetr import unetr_state_shapes, vit_intermediates  # noqa: E402
from oracle.vit import VIT_CONFIGS  # noqa: E402


def main():
    VT, refgen, refsrc = load_reference()
    U = sys.modules["refgen.unet"]
    out_dir = os.path.join(ROOT, "tests", "golden")
    for name, cname, img, nc, B, seed in [("tiny4_gelu_p16_128", "tiny4", 128, 3, 2, 41),
                                          ("tiny4_swiglu_p14_128", "tiny4_swiglu", 128, 5, 2, 42)]:
        cfg = VIT_CONFIGS[cname]
        sd = det_state_dict(unetr_state_shapes(cfg, img, nc), seed=seed, layerscale=0.5)
        p = {k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}
        sf = None if cfg.patch == 16 else int(img / 16) / int(img / cfg.patch)
        up = U.ViTFeatureUpsampler(cfg.dim, drop_rate=0.0, scale_factor=sf)
        dec = U.Decoder(up.out_channels, out_channels=32, drop_rate=0.0)
        heads = nn.ModuleList([U.SegmentationHead(32, 1, kernel_size=3, activation=nn.Tanh(), use_attention=True) for _ in range(nc)])
        missing = up.load_state_dict({k[len("encoder.feature_upsampler."):]: v for k, v in p.items()
                                      if k.startswith("encoder.feature_upsampler.")}, strict=True)
        dec.load_state_dict({k[len("decoder."):]: v for k, v in p.items() if k.startswith("decoder.")}, strict=True)
        for h in range(nc):
            heads[h].load_state_dict({k[len(f"segmentation_head_{h}."):]: v for k, v in p.items()
                                      if k.startswith(f"segmentation_head_{h}.")}, strict=True)
        x, _ = synth_batch(seed, B, img, nc)
        with torch.no_grad():
            feats = vit_intermediates(p, x, cfg, "encoder.model.", True)
        rec = {"cfg": cname, "img": img, "nc": nc, "batch": B, "seed": seed,
               "keys": np.array(sorted(list(p.keys())))}
        for mode in ("eval", "train"):
            for m in (up, dec, heads):
                m.train(mode == "train")
            with torch.no_grad():
                z = up(x, [f.clone() for f in feats])
                f = dec(z)
                out = torch.cat([hd(f) for hd in heads], 1)
            rec[f"out_{mode}"] = out.numpy()
            rec[f"dec_{mode}"] = f.numpy()[:, :, ::4, ::4]
            rec[f"z1_{mode}"] = z[1].numpy()[:, ::8, ::4, ::4]
        rec["bn_rm_after"] = dec.decoder0_header[1].block[1].running_mean.numpy()
        rec["bn_rv_after"] = dec.decoder0_header[1].block[1].running_var.numpy()
        rec["up_bn_rv_after"] = up.upsampler0[1].block[2].running_var.numpy()
        np.savez_compressed(os.path.join(out_dir, f"unetr_{name}.npz"), **rec)
        print("wrote", name, float(out.abs().mean()))


if __name__ == "__main__":
    main()
