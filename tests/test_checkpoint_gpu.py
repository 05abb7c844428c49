"""GPU: checkpoint ingestion end to end: a trained generator is pruned to model.safetensors (LoRA + decoder
keys, scripts/ckpt_remove_foundation_model.py), loaded into a fresh generator that already holds the frozen encoder
(inference.py:135-153 + validate_load_info), and the HIP eval forward -- LoRA merged into the packed qkv weights, BatchNorm
running statistics -- matches the fp32 oracle on the source weights."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("cfgname,img", [("tiny_swiglu", 128), ("tiny", 256)])
def test_pruned_safetensors_roundtrip_then_hip_forward_matches_oracle(tmp_path, cfgname, img):
    from oracle import VIT_CONFIGS, det_state_dict, synth_batch
    from oracle.model import generator_forward, generator_state_shapes
    from miphei_vit_amd.checkpoint import load_generator_checkpoint, save_pruned_safetensors
    from miphei_vit_amd.generators import get_vitmatte
    nc, B = 3, 2
    cfg = VIT_CONFIGS[cfgname]
    shapes = generator_state_shapes(cfg, img, nc)
    src_sd = {k: torch.from_numpy(np.asarray(v)) for k, v in det_state_dict(shapes, seed=11, layerscale=0.5).items()}
    g = torch.Generator().manual_seed(0)
    for k in src_sd:                        # running statistics a trained model would carry
        if k.endswith("running_mean"):
            src_sd[k] = torch.randn(src_sd[k].shape, generator=g) * 0.1
        elif k.endswith("running_var"):
            src_sd[k] = torch.rand(src_sd[k].shape, generator=g) + 0.5
    src = get_vitmatte(cfgname, img, nc, use_lora=True, pretrained=False)
    src.load_state_dict(src_sd)
    keys = save_pruned_safetensors(src, tmp_path / "model.safetensors")
    assert keys and not any(k.startswith("encoder.vit.") and ".lora" not in k for k in keys)
    # the deployment side: frozen encoder from its own source, everything trainable still at another initialisation
    other = {k: torch.from_numpy(np.asarray(v)) for k, v in det_state_dict(shapes, seed=12, layerscale=0.5).items()}
    dst_sd = {k: (src_sd[k] if (k.startswith("encoder.vit.") and ".lora" not in k) else other[k]) for k in src_sd}
    dst = get_vitmatte(cfgname, img, nc, use_lora=True, pretrained=False)
    dst.load_state_dict(dst_sd)
    dst.cuda().eval()
    x, _ = synth_batch(5, B, img, nc)
    with torch.no_grad():
        before = dst(x.cuda()).float().cpu()
    info = load_generator_checkpoint(dst, tmp_path)          # on a model that has already run: caches must be dropped
    assert all(k.startswith("encoder.vit.") for k in info.missing_keys)
    with torch.no_grad():
        out = dst(x.cuda()).float().cpu()
        ref = generator_forward(src_sd, x, cfg, nc, training=False)
    rel = ((out - ref) ** 2).sum(dim=(0, 2, 3)) / (ref ** 2).sum(dim=(0, 2, 3))
    assert float(rel.max()) < 1e-3, rel
    rel_before = ((before - ref) ** 2).sum(dim=(0, 2, 3)) / (ref ** 2).sum(dim=(0, 2, 3))
    assert float(rel_before.min()) > 1e-2               # the load really changed what the engine computes
    a, b = src.state_dict(), dst.state_dict()
    assert all(torch.equal(a[k], b[k].cpu()) for k in keys)
