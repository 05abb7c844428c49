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
