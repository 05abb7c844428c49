"""UNETR baseline (`unet_lora`): the oracle restatement (oracle/unetr.py) against fixtures produced by
the reference's own ViTFeatureUpsampler / Decoder / SegmentationHead modules (oracle/make_golden_unetr.py)."""
import os

import numpy as np
import pytest
import torch


def _load(golden_dir, name):
    from oracle import VIT_CONFIGS, det_state_dict
    from oracle.unetr import unetr_state_shapes
    g = np.load(os.path.join(golden_dir, f"unetr_{name}.npz"))
    cfg = VIT_CONFIGS[str(g["cfg"])]
    img, nc, B, seed = int(g["img"]), int(g["nc"]), int(g["batch"]), int(g["seed"])
    sd = det_state_dict(unetr_state_shapes(cfg, img, nc), seed=seed, layerscale=0.5)
    p = {k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}
    return g, cfg, p, img, nc, B, seed


def _rel(a, b):
    a, b = torch.as_tensor(a).double(), torch.as_tensor(b).double()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


@pytest.mark.parametrize("name", ["tiny4_gelu_p16_128", "tiny4_swiglu_p14_128"])
def test_oracle_unetr_matches_reference(golden_dir, name):
    from oracle import synth_batch
    from oracle.unetr import extract_layers, unetr_forward
    g, cfg, p, img, nc, B, seed = _load(golden_dir, name)
    assert sorted(p.keys()) == list(g["keys"])
    x, _ = synth_batch(seed, B, img, nc)
    with torch.no_grad():
        out_eval = unetr_forward(p, x, cfg, nc, training=False)
        stats = {}
        out_train = unetr_forward(p, x, cfg, nc, training=True, new_stats=stats)
    assert _rel(out_eval, g["out_eval"]) < 1e-5
    assert _rel(out_train, g["out_train"]) < 1e-5
    assert _rel(stats["decoder.decoder0_header.1.block.1.running_mean"], g["bn_rm_after"]) < 1e-5
    assert _rel(stats["decoder.decoder0_header.1.block.1.running_var"], g["bn_rv_after"]) < 1e-5
    assert _rel(stats["encoder.feature_upsampler.upsampler0.1.block.2.running_var"], g["up_bn_rv_after"]) < 1e-5
    assert extract_layers(40) == [10, 20, 29, 39] and extract_layers(4) == [0, 1, 2, 3]      # unet.py:131-137
    with pytest.raises(ValueError):
        extract_layers(2)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["tiny4_gelu_p16_128", "tiny4_swiglu_p14_128"])
def test_hip_unetr_forward_matches_reference(golden_dir, name):
    """HIP forward of the UNETR baseline (eval and train-mode BatchNorm) against the reference-module fixtures; tolerance = the
    north-star 1e-3 relative MSE per channel (bf16 operands, fp32 accumulation)."""
    from oracle import synth_batch
    from miphei_vit_amd.generators import get_generator
    g, cfg, p, img, nc, B, seed = _load(golden_dir, name)
    conf = {"model": {"encoder": {"encoder_name": str(g["cfg"]), "encoder_weights": None, "pretrained": False}, "dropout": 0.0},
            "train": {"foreground_head": False}}
    model = get_generator("unet_lora", img, 3, nc, conf)
    assert sorted(model.state_dict().keys()) == list(g["keys"])
    model.load_state_dict(p)
    model = model.cuda()
    x, _ = synth_batch(seed, B, img, nc)

    def chan_rel_mse(a, b):
        a, b = a.double(), torch.as_tensor(b).double()
        return float((((a - b) ** 2).sum(dim=(0, 2, 3)) / (b ** 2).sum(dim=(0, 2, 3))).max())

    model.eval()
    with torch.no_grad():
        out = model(x.cuda()).float().cpu()
    assert chan_rel_mse(out, g["out_eval"]) < 1e-3
    model.train()
    with torch.no_grad():
        out = model(x.cuda()).float().cpu()
    assert chan_rel_mse(out, g["out_train"]) < 1e-3
    sd = model.state_dict()
    assert _rel(sd["decoder.decoder0_header.1.block.1.running_var"].cpu(), g["bn_rv_after"]) < 2e-2
    assert _rel(sd["decoder.decoder0_header.1.block.1.running_mean"].cpu(), g["bn_rm_after"]) < 2e-2
    assert int(sd["decoder.decoder0_header.1.block.1.num_batches_tracked"]) == 1
