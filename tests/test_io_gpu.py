"""GPU: on-device input / output stage against the numpy arithmetic of the reference's NormalizationLayer and export."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_input_and_export_stage():
    from miphei_vit_amd.io_stage import HOPTIMUS_MEAN, HOPTIMUS_STD, InputStage, export_uint8
    rng = np.random.default_rng(0)
    rgb = rng.integers(0, 256, size=(3, 64, 128, 3), dtype=np.uint8)
    mif = rng.integers(0, 256, size=(3, 64, 128, 16), dtype=np.uint8)
    st = InputStage("cuda")
    x = st.image(torch.from_numpy(rgb).cuda()).cpu().numpy()
    mean = np.float32(np.array(HOPTIMUS_MEAN).reshape(1, 1, -1)); std = np.float32(np.array(HOPTIMUS_STD).reshape(1, 1, -1))
    ref = np.stack([((im - mean) / std).transpose(2, 0, 1) for im in rgb.astype(np.float32)])   # dataset.py:570 + ToTensor
    assert np.allclose(x, ref, rtol=1e-6, atol=1e-5)
    y = st.target(torch.from_numpy(mif).cuda()).cpu().numpy()
    ref_y = np.stack([(np.float32(im) / 255 * 1.8 - 0.9).transpose(2, 0, 1) for im in mif])      # dataset.py:573
    assert np.allclose(y, ref_y, rtol=1e-6, atol=1e-6)
    # export: bit-exact uint8 (callbacks.py:345-346), including out-of-range predictions
    pred = torch.from_numpy(ref_y).cuda() * 1.2
    got = export_uint8(pred).cpu()
    want = ((pred + 0.9) / 1.8).clamp(0, 1)
    want = (want * 255).to(torch.uint8).cpu()
    assert torch.equal(got, want)


def test_normalisation_matches_the_reference_fixture(golden_dir):
    """NormalizationLayer('he') / ('if') of the reference itself (oracle/make_golden_io.py), bit for bit."""
    import os
    from miphei_vit_amd.io_stage import InputStage, TrainAugmenter
    g = np.load(os.path.join(golden_dir, "comp_io.npz"))
    rgb, mif = torch.from_numpy(g["rgb"])[None].cuda(), torch.from_numpy(g["mif"])[None].cuda()
    st = InputStage("cuda", mean=g["mean"], std=g["std"])
    assert np.array_equal(st.image(rgb)[0].cpu().numpy(), g["he"].transpose(2, 0, 1))
    assert np.array_equal(st.target(mif)[0].cpu().numpy(), g["mif_norm"].transpose(2, 0, 1))
    # the augmenter with every probability at zero and crop = tile is the same map (+ the bf16 NHWC image buffer)
    aug = TrainAugmenter("cuda", (40, 52), seed=3, mean=g["mean"], std=g["std"], p_hflip=0.0, p_vflip=0.0, p_drop=0.0)
    b = aug(rgb, mif, 0)
    assert np.array_equal(b["image"][0].cpu().numpy(), g["he"].transpose(2, 0, 1))
    assert np.array_equal(b["target"][0].cpu().numpy(), g["mif_norm"].transpose(2, 0, 1))
    n8 = b["image_nhwc8"][0].float().cpu()
    assert torch.equal(n8[..., :3], torch.from_numpy(g["he"]).bfloat16().float()) and float(n8[..., 3:].abs().max()) == 0.0


@pytest.mark.parametrize("Hs,Ws,crop,C", [(300, 320, (256, 256), 16), (128, 128, (128, 128), 3), (70, 96, (64, 64), 5)])
def test_spatial_augmentation_bit_exact(Hs, Ws, crop, C):
    """RandomCrop / HorizontalFlip / VerticalFlip / CoarseDropout jointly on image and target (dataset.py:458-468), draws
    recomputed on the host: every output element equals the numpy pipeline (oracle/io.py) bit for bit."""
    from oracle.io import normalize_he, normalize_if, spatial_augment
    from miphei_vit_amd.io_stage import HOPTIMUS_MEAN, HOPTIMUS_STD, TrainAugmenter
    rng = np.random.default_rng(Hs)
    B = 12
    rgb = rng.integers(0, 256, size=(B, Hs, Ws, 3), dtype=np.uint8)
    mif = rng.integers(0, 256, size=(B, Hs, Ws, C), dtype=np.uint8)
    aug = TrainAugmenter("cuda", crop, seed=99, p_drop=0.6, rank_offset=1000)
    sample0 = 48
    out = aug(torch.from_numpy(rgb).cuda(), torch.from_numpy(mif).cuda(), sample0)
    seen = {"hflip": 0, "vflip": 0, "drop": 0}
    for b in range(B):
        d = aug.params(sample0 + b, (Hs, Ws))
        assert 0 <= d["oy"] <= Hs - crop[0] and 0 <= d["ox"] <= Ws - crop[1]
        assert d["hh"] <= int(0.3 * crop[0]) and d["hw"] <= int(0.3 * crop[1])
        for k in seen:
            seen[k] += d[k]
        xi = normalize_he(spatial_augment(rgb[b], d, crop), HOPTIMUS_MEAN, HOPTIMUS_STD).transpose(2, 0, 1)
        yi = normalize_if(spatial_augment(mif[b], d, crop)).transpose(2, 0, 1)
        assert np.array_equal(out["image"][b].cpu().numpy(), xi), (b, d)
        assert np.array_equal(out["target"][b].cpu().numpy(), yi), (b, d)
        assert torch.equal(out["image_nhwc8"][b, ..., :3].float().cpu(), torch.from_numpy(xi.transpose(1, 2, 0).copy()).bfloat16().float())
    assert all(0 < v < B for v in seen.values()), seen           # every branch was taken and not taken
    # counter-based: the same samples in a different batch split give the same tiles
    again = aug(torch.from_numpy(rgb[4:9]).cuda(), torch.from_numpy(mif[4:9]).cuda(), sample0 + 4)
    assert torch.equal(again["image"], out["image"][4:9]) and torch.equal(again["target"], out["target"][4:9])


def test_training_step_from_uint8_tiles_through_the_augmenter():
    """uint8 tiles -> TrainAugmenter -> ModelModule.training_step: the bf16 NHWC image the augmenter writes is consumed in place
    (no NCHW -> NHWC pass) and gives exactly the step the f32 image alone gives."""
    from oracle import VIT_CONFIGS, det_state_dict
    from oracle.model import generator_state_shapes, orion_marker_weights
    from miphei_vit_amd.generators import get_vitmatte
    from miphei_vit_amd.io_stage import TrainAugmenter
    from miphei_vit_amd.loss import WeightedMSELoss
    from miphei_vit_amd.models import ModelModule
    cfgname, img, nc, B = "tiny_swiglu", 128, 3, 2
    sd = det_state_dict(generator_state_shapes(VIT_CONFIGS[cfgname], img, nc), seed=4, layerscale=0.5)
    rng = np.random.default_rng(1)
    rgb = torch.from_numpy(rng.integers(0, 256, size=(B, 160, 144, 3), dtype=np.uint8)).cuda()
    mif = torch.from_numpy(rng.integers(0, 256, size=(B, 160, 144, nc), dtype=np.uint8)).cuda()
    aug = TrainAugmenter("cuda", (img, img), seed=11)
    losses, params = [], []
    for with_n8 in (True, False):
        model = get_vitmatte(cfgname, img, nc, use_lora=True, pretrained=False)
        model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
        model.cuda()
        mod = ModelModule(model, None, 1e-3, 0., WeightedMSELoss(50.0, orion_marker_weights(nc)))
        mod.total_iters = 100
        mod.global_step_ = 500                                     # past the warm-up: the step moves the weights
        mod.total_iters = 2000
        batch = aug(rgb, mif, 0, nhwc8=with_n8)
        assert ("image_nhwc8" in batch) == with_n8
        losses.append(float(mod.training_step(batch, 0)))
        params.append(model._engine._flat.flat.clone())
    assert losses[0] == losses[1] or abs(losses[0] - losses[1]) < 1e-4 * abs(losses[1])
    assert float((params[0] - params[1]).abs().max()) < 1e-4 * float(params[1].abs().max())
