"""On-device PSNR / SSIM state updates (csrc/metrics.hip behind miphei_vit_amd.metrics.PixMetrics) against the CPU
restatement of torchmetrics 1.6.2 in oracle/metrics.py (reference call sites: src/models.py:35-52,140-143)."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("shape", [(3, 5, 64, 48), (2, 16, 256, 256), (1, 1, 12, 33)])
def test_pix_metrics_match_oracle(shape):
    from oracle.metrics import psnr_compute, psnr_update, ssim_update
    from miphei_vit_amd.metrics import PixMetrics
    g = torch.Generator().manual_seed(sum(shape))
    m = PixMetrics((-0.9, 0.9), prefix="val_")
    sse, n, ssim_sum, ni = 0.0, 0, 0.0, 0
    for it in range(2):                                  # two updates accumulate into one state
        t = (torch.rand(shape, generator=g) * 2.2 - 1.1)                          # beyond the clamp range on both sides
        p = (t + 0.25 * torch.randn(shape, generator=g)).clamp(-1.2, 1.2)
        m.update(p.cuda(), t.cuda())
        a, b = psnr_update(p, t)
        sse, n = sse + float(a), n + b
        s = ssim_update(p, t)
        ssim_sum, ni = ssim_sum + float(s.sum()), ni + shape[0]
    got = m.compute()
    assert set(got) == {"val_psnr_metric", "val_ssim_metric"}
    assert abs(got["val_psnr_metric"] - psnr_compute(sse, n)) < 1e-4
    assert abs(got["val_ssim_metric"] - ssim_sum / ni) < 2e-5
    m.reset()
    assert m.compute() == {}


def test_training_step_updates_metrics():
    from oracle import synth_batch
    from oracle.metrics import psnr_compute, psnr_update, ssim_update
    from oracle.model import orion_marker_weights
    from miphei_vit_amd.generators import get_vitmatte
    from miphei_vit_amd.loss import WeightedMSELoss
    from miphei_vit_amd.models import ModelModule
    nc, B, img = 3, 2, 128
    model = get_vitmatte("tiny", img, nc, use_lora=True, pretrained=False).cuda()
    mod = ModelModule(model, None, 1e-3, 0.0, WeightedMSELoss(50.0, orion_marker_weights(16)[:nc])).cuda()
    mod.total_iters = 100
    x, y = synth_batch(3, B, img, nc)
    model.train()
    with torch.no_grad():
        out = model(x.cuda()).float().cpu()              # same weights, same batch statistics as the step below
    mod.training_step({"image": x.cuda(), "target": y.cuda()}, 0)
    vals = mod.on_train_epoch_end()
    a, b = psnr_update(out, y)
    assert abs(vals["psnr_metric"] - psnr_compute(a, b)) < 1e-3
    assert abs(vals["ssim_metric"] - float(ssim_update(out, y).mean())) < 1e-4
    assert mod.train_pix_metrics.compute() == {}          # reset at epoch end
    assert math.isfinite(vals["psnr_metric"])
