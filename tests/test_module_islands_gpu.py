"""GPU: the (f)-row kernels reached through the module surface, as the reference reaches them:
``ModelModule(cell_metrics=...)`` -> ``validation_step`` -> ``cell_metrics.update(fake, batch["nuclei"], batch["slide_name"])``
(``/root/reference/src/models.py:54-61, 233-241``) and ``run.py ++data.uint8_tiles=...`` -> ``TrainAugmenter`` -> ``training_step``
with the in-place bf16 NHWC image (``/root/reference/src/train.py:80-90``, ``src/dataset.py:244-311, 458-483``)."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


class _FixedGenerator(nn.Module):
    """returns a preset prediction: lets validation_step be checked against the fixture captured from the reference's CellMetrics"""

    def __init__(self, pred):
        super().__init__()
        self.pred = pred

    def forward(self, x):
        return self.pred[: x.shape[0]]


def test_validation_step_updates_cell_metrics_as_reference_fixture(golden_dir):
    from miphei_vit_amd.cells import CellMetrics
    from miphei_vit_amd.loss import WeightedMSELoss
    from miphei_vit_amd.models import ModelModule
    from tests.test_cells_gpu import _fixture_inputs
    g = np.load(os.path.join(golden_dir, "comp_cells.npz"))
    pred, target, nuclei = _fixture_inputs(g)
    names = ["Hoechst", "CD31", "CD45", "CD68", "CD4"]
    cm = CellMetrics(["slideA", "slideB"], names)
    loss = WeightedMSELoss(0.5, torch.ones(pred.shape[1]))
    module = ModelModule(_FixedGenerator(pred.cuda()), None, 1e-4, 1e-4, loss, cell_metrics=cm).cuda()
    assert module.use_cell_metrics and module.logreg_layer.in_features == 4            # Hoechst excluded (metrics.py:16-22)
    batch = {"image": torch.zeros(pred.shape[0], 3, 8, 8, device="cuda"), "target": target.cuda(), "nuclei": nuclei,
             "slide_name": ["slideA", "slideB", "slideA", "slideB"]}
    val_loss = module.validation_step(batch, 0)
    ref_loss = loss.cpu()(target, pred)
    assert abs(float(val_loss) - float(ref_loss)) < 1e-5 * max(1.0, abs(float(ref_loss)))
    for s in ("slideA", "slideB"):
        assert len(cm.state[s]["cell_id"]) == int(g[f"cm_{s}_n"])
        for i in range(int(g[f"cm_{s}_n"])):
            assert np.array_equal(cm.state[s]["cell_id"][i].numpy(), g[f"cm_{s}_id{i}"])
            assert np.array_equal(cm.state[s]["area"][i].numpy(), g[f"cm_{s}_area{i}"])
            assert np.abs(cm.state[s]["sum"][i].numpy() - g[f"cm_{s}_sum{i}"]).max() <= 1
    with pytest.raises(NotImplementedError):
        ModelModule(_FixedGenerator(pred), None, 1e-4, 1e-4, loss, cell_loss=nn.MSELoss())
    with pytest.raises(TypeError):
        ModelModule(_FixedGenerator(pred), None, 1e-4, 1e-4, loss, cell_metrics=object())


def test_run_py_trains_on_resident_uint8_tiles_and_validates_with_cell_metrics(tmp_path):
    rng = np.random.default_rng(11)
    S, C = 128, 3
    img = rng.integers(0, 256, (6, S + 16, S + 8, 3), dtype=np.uint8)
    tgt = rng.integers(0, 256, (6, S + 16, S + 8, C), dtype=np.uint8)
    np.savez(tmp_path / "train.npz", image=img, target=tgt)
    lab = np.zeros((4, S, S), dtype=np.int32)
    lab[0, 10:20, 10:30] = 7
    lab[0, 50:60, 40:44] = 9
    lab[1, 0:5, 0:5] = 123456
    lab[3, 100:128, 100:128] = 7          # same id on another slide: slides are separate tables
    np.savez(tmp_path / "val.npz", image=img[:4, :S, :S], target=tgt[:4, :S, :S], nuclei=lab,
             slide_name=np.array(["s1", "s1", "s2", "s2"]))
    p = subprocess.run([sys.executable, "run.py", "+default_configs=tiny", f"++data.tile_size={S}", "++train.batch_size=2",
                        "++train.max_steps=4", "++train.use_cell_metrics=true", f"++data.uint8_tiles={tmp_path / 'train.npz'}",
                        f"++data.val_uint8_tiles={tmp_path / 'val.npz'}"], cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, (p.stdout[-1500:], p.stderr[-3000:])
    losses = [float(l.split("loss")[1].split()[0]) for l in p.stdout.splitlines() if l.startswith("step")]
    assert len(losses) >= 2 and all(np.isfinite(losses))
    val = [l for l in p.stdout.splitlines() if l.startswith("validation:")]
    assert val and "4 tiles" in val[0] and "cell_metrics: 4 nuclei over 2 slides" in val[0], p.stdout[-800:]


def test_train_augmenter_batch_equals_plain_batch_through_training_step():
    """the bf16 NHWC image written by the augmenter is what the engine would have converted itself: same weights, one step each --
    once with the augmenter's batch dict (image_nhwc8 consumed in place), once with its f32 tensors only"""
    from oracle import VIT_CONFIGS, det_state_dict
    from oracle.model import generator_state_shapes
    from miphei_vit_amd.generators import get_vitmatte
    from miphei_vit_amd.io_stage import TrainAugmenter
    from miphei_vit_amd.loss import WeightedMSELoss
    from miphei_vit_amd.models import ModelModule
    rng = np.random.default_rng(3)
    cfgname, S, C, B = "tiny_swiglu", 128, 3, 2        # patch 14 on 128 px: the patch-embed gather reads the NHWC image in place
    img = torch.from_numpy(rng.integers(0, 256, (B, S + 8, S + 8, 3), dtype=np.uint8)).cuda()
    tgt = torch.from_numpy(rng.integers(0, 256, (B, S + 8, S + 8, C), dtype=np.uint8)).cuda()
    sd = det_state_dict(generator_state_shapes(VIT_CONFIGS[cfgname], S, C), seed=4, layerscale=0.5)
    p0 = {k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}
    res = []
    for with_n8 in (True, False):
        model = get_vitmatte(cfgname, S, C, use_lora=True, pretrained=False)
        model.load_state_dict(p0)
        model.cuda()
        m = ModelModule(model, None, 2e-4, 0., WeightedMSELoss(0.5, torch.ones(C)))
        m.total_iters = 10
        m.update_pix_metrics = False
        batch = TrainAugmenter("cuda", (S, S), seed=5)(img, tgt, 0)
        assert batch["image_nhwc8"].dtype == torch.bfloat16 and tuple(batch["image_nhwc8"].shape) == (B, S, S, 8)
        if not with_n8:
            batch.pop("image_nhwc8")
        loss = float(m.training_step(batch, 0))
        res.append((loss, float(torch.sqrt(model._engine._saved.w.sqn[0]))))
    assert np.isfinite(res[0][0]) and abs(res[0][0] - res[1][0]) <= 2e-5 * max(1.0, abs(res[1][0]))
    assert abs(res[0][1] - res[1][1]) <= 2e-2 * res[1][1]      # (train-mode BN statistics meet in f32 atomics: run-to-run noise)
