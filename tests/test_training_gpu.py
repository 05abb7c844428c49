"""GPU: fused ModelModule.training_step against the golden training fixtures captured from the reference's own
ModelModule.training_step (oracle/make_golden.py) and against the oracle trainer."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _rel(a, b):
    a, b = torch.as_tensor(a).double().cpu(), torch.as_tensor(b).double().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


@pytest.mark.parametrize("name", ["tiny_gelu_p16_128", "tiny_swiglu_p14_128"])
def test_training_steps_match_reference(golden_dir, name):
    from oracle import VIT_CONFIGS, det_state_dict, synth_batch
    from oracle.model import OracleTrainer, generator_state_shapes, orion_marker_weights
    from miphei_vit_amd.generators import get_vitmatte
    from miphei_vit_amd.loss import WeightedMSELoss
    from miphei_vit_amd.models import ModelModule
    g = np.load(os.path.join(golden_dir, f"train_{name}.npz"))
    cfgname, img, nc, B, seed = str(g["cfg"]), int(g["img"]), int(g["nc"]), int(g["batch"]), int(g["seed"])
    cfg = VIT_CONFIGS[cfgname]
    sd = det_state_dict(generator_state_shapes(cfg, img, nc), seed=seed, layerscale=0.5)
    p = {k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}
    model = get_vitmatte(cfgname, img, nc, use_lora=True, pretrained=False)
    model.load_state_dict(p)
    model.cuda()
    mod = ModelModule(model, None, float(g["lr_g"]), 0., WeightedMSELoss(50.0, orion_marker_weights(nc)))
    mod.total_iters = int(g["total_iters"])
    tr = OracleTrainer(p, cfg, nc, batch_size=B, total_iters=int(g["total_iters"]))
    tr.base_lr = float(g["lr_g"])
    for it in range(3):
        x, y = synth_batch(seed * 100 + it, B, img, nc)
        assert abs(mod.current_lr() - g["lrs"][it]) < 1e-12
        loss = float(mod.training_step({"image": x.cuda(), "target": y.cuda()}, it))
        r = tr.step(x, y)
        # reference fixture: loss of the reference's own training_step
        assert abs(loss - g["losses"][it]) < 3e-3 * g["losses"][it]
        assert abs(loss - r["loss"]) < 3e-3 * r["loss"]
        gn = float(torch.sqrt(model._engine._saved.w.sqn[0]))
        assert abs(gn - g["grad_norms"][it]) < 3e-2 * g["grad_norms"][it]
    named = dict(model.named_parameters())
    for k in g["watch"]:
        k = str(k)
        ref = torch.from_numpy(g["after3::" + k])
        mine = named[k].detach().cpu()
        if mine.numel() > 20000:
            mine = mine.reshape(-1)[::37]
        # parameters moved by 3 Adam steps: compare the displacement as well as the value
        assert _rel(mine, ref) < 2e-3, k  # Adam turns bf16 gradient noise into +-lr moves
    for k, v in tr.p.items():
        if k.endswith("psi.0.bias"):
            continue  # bias in front of a train-mode BatchNorm: exact gradient is 0 (the reference moves it by fp noise only)
        if k in named and named[k].requires_grad:
            d_ref = v - p[k]
            d_got = named[k].detach().cpu() - p[k]
            if float(d_ref.norm()) > 0:
                assert _rel(d_got, d_ref) < 0.35, k  # Adam normalises gradients: bf16 noise shows up in tiny moves


def test_lora_group_size_does_not_change_gradients():
    """The LoRA weight-gradient products run batched over groups of ViT blocks (HipEngine.lora_group); any grouping - one block per
    launch, a partial group, all blocks - must give the same LoRA gradients."""
    from oracle import VIT_CONFIGS, det_state_dict, synth_batch
    from oracle.model import generator_state_shapes, orion_marker_weights
    from miphei_vit_amd.generators import get_vitmatte
    from miphei_vit_amd.loss import WeightedMSELoss
    from miphei_vit_amd.models import ModelModule
    cfgname, img, nc, B = "tiny4_swiglu", 128, 3, 2
    sd = det_state_dict(generator_state_shapes(VIT_CONFIGS[cfgname], img, nc), seed=5, layerscale=0.5)
    p = {k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}
    x, y = synth_batch(77, B, img, nc)
    grads = []
    for group in (1, 1, 3, 10):
        model = get_vitmatte(cfgname, img, nc, use_lora=True, pretrained=False)
        model.load_state_dict(p)
        model.cuda()
        model._engine.lora_group = group
        mod = ModelModule(model, None, 1e-4, 0., WeightedMSELoss(50.0, orion_marker_weights(nc)))
        mod.total_iters = 100
        mod.training_step({"image": x.cuda(), "target": y.cuda()}, 0)
        c = model._engine._config()
        grads.append(model._engine._flat.gflat[:c.L * 4 * c.rank * c.D].clone())     # the LoRA region of the flat gradient
    assert float(grads[0].abs().max()) > 0
    # two runs of the SAME grouping differ (f32 atomics upstream flip bf16 roundings): that is the yardstick; a wrong stride is O(1)
    noise = _rel(grads[1], grads[0])
    assert noise < 0.05
    for g in grads[2:]:
        assert _rel(g, grads[0]) < max(3 * noise, 1e-3)


@pytest.mark.parametrize("n", [1, 3, 4, 1023, 262144 * 4 + 5, 6_697_712, 6_697_713])
def test_squared_norm_kernel_any_length(n):
    """mvit_sqnorm over lengths around its unrolled trips (four 16-byte groups, single groups, ragged end): f64 sum of squares"""
    import miphei_vit_amd.ops as ops
    g = torch.Generator(device="cuda").manual_seed(n % 1000)
    x = torch.randn(n + 3, device="cuda", generator=g)[:n]          # (a view: same alignment as the allocation)
    out = torch.full((1,), 2.5, device="cuda", dtype=torch.float64)
    ops.sqnorm(x, out)
    ref = 2.5 + float((x.double() ** 2).sum())
    assert abs(float(out) - ref) < 2e-6 * max(1.0, abs(ref))      # (pairs of squares are added in f32 before the f64 accumulation)
