"""GPU: device-gated NaN guard (reference src/models.py:102-105) and optimiser-state persistence / checkpoint resume of the
fused training step (the state the reference's Lightning checkpoints carry)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _module(seed=3, B=2, img=128, nc=3, cfgname="tiny_swiglu"):
    from oracle import VIT_CONFIGS, det_state_dict, synth_batch
    from oracle.model import generator_state_shapes, orion_marker_weights
    from miphei_vit_amd.generators import get_vitmatte
    from miphei_vit_amd.loss import WeightedMSELoss
    from miphei_vit_amd.models import ModelModule
    sd = det_state_dict(generator_state_shapes(VIT_CONFIGS[cfgname], img, nc), seed=seed, layerscale=0.5)
    p = {k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}
    model = get_vitmatte(cfgname, img, nc, use_lora=True, pretrained=False)
    model.load_state_dict(p)
    model.cuda()
    mod = ModelModule(model, None, 2e-3, 0., WeightedMSELoss(50.0, orion_marker_weights(nc)))
    mod.total_iters = 1000
    mod.global_step_ = 400          # past the warm-up: lr = lr_g, parameters really move
    batches = [tuple(t.cuda() for t in synth_batch(seed * 100 + i, B, img, nc)) for i in range(6)]
    return mod, model, batches


def _trainable(model):
    return {k: p.detach().clone() for k, p in model.named_parameters() if p.requires_grad}


def test_nan_step_is_refused_on_the_device_and_reported(tmp_path, monkeypatch):
    monkeypatch.chdir(tmp_path)
    mod, model, batches = _module()
    for i in range(2):
        mod.training_step({"image": batches[i][0], "target": batches[i][1]}, i)
    torch.cuda.synchronize()
    before = _trainable(model)
    m_before = model._engine._flat.m.clone()
    bad = batches[2][0].clone()
    bad[0, 1, 5, 7] = float("nan")
    with pytest.raises(ValueError, match="Nan found"):
        mod.training_step({"image": bad, "target": batches[2][1]}, 2)
        # the host check is asynchronous: later (finite) steps must not move the weights either, the flag is sticky
        for i in range(3, 6):
            mod.training_step({"image": batches[i][0], "target": batches[i][1]}, i)
        mod.on_train_end()
    torch.cuda.synchronize()
    after = _trainable(model)
    assert all(torch.equal(before[k], after[k]) for k in before)          # last finite weights, bit for bit
    assert all(bool(torch.isfinite(v).all()) for v in after.values())
    assert torch.equal(m_before, model._engine._flat.m)                   # Adam moments untouched as well
    ck = torch.load(tmp_path / "weights_nan.ckpt", weights_only=False)
    # every PARAMETER in the dump is finite (BatchNorm running statistics have seen the NaN batch in the forward pass, exactly
    # as in the reference, whose check also sits behind the generator call)
    pnames = {"generator." + k for k, _ in model.named_parameters()}
    assert all(bool(torch.isfinite(v).all()) for k, v in ck["state_dict"].items() if k in pnames)
    k0 = next(iter(before))
    assert torch.equal(ck["state_dict"]["generator." + k0], before[k0].cpu())


def test_optimizer_state_survives_reflatten_and_checkpoint_resume():
    mod, model, batches = _module()
    for i in range(3):
        mod.training_step({"image": batches[i][0], "target": batches[i][1]}, i)
    eng = model._engine
    m3, v3, step3 = eng._flat.m.clone(), eng._flat.v.clone(), eng._flat.step
    assert step3 == 3 and float(m3.abs().sum()) > 0
    ckpt = mod.checkpoint_state()
    p3 = _trainable(model)
    assert ckpt["global_step"] == 403 and ckpt["optimizer_state"]["step"] == 3
    assert all(k.startswith(("generator.", "loss_reconstruct.")) for k in ckpt["state_dict"])
    # (a) .cuda() / load_state_dict re-flatten the parameters: moments and step count carry over
    model.cuda()
    model.load_state_dict(model.state_dict())
    assert eng._flat is None
    mod.training_step({"image": batches[3][0], "target": batches[3][1]}, 3)
    assert eng._flat.step == 4
    ref4 = _trainable(model)
    # (b) a fresh module resumed from the checkpoint takes the same 4th step
    mod2, model2, _ = _module(seed=5)            # different initial weights: everything must come from the checkpoint
    mod2.load_checkpoint_state(ckpt)
    assert mod2.global_step_ == 403
    mod2.training_step({"image": batches[3][0], "target": batches[3][1]}, 3)
    assert model2._engine._flat.step == 4
    got4 = _trainable(model2)
    for k in ref4:
        d, moved = float((got4[k] - ref4[k]).abs().max()), float((ref4[k] - p3[k]).abs().max())
        # same kernels, same inputs, same Adam state: the two 4th steps agree to a few % of the step itself (the gradients'
        # f32 atomics are summed in a different order run to run, Adam's normalisation turns that into ~1e-5 moves); a lost or
        # zeroed moment estimate would change the step by its own size
        if moved == 0.0:          # bias in front of a train-mode BatchNorm: its gradient is exactly zero
            assert d == 0.0, k
            continue
        # (single elements with a near-zero gradient sit closest to the noise: the bound on the maximum is looser than the one
        # on the norm of the whole tensor)
        dn, mn = float((got4[k] - ref4[k]).norm()), float((ref4[k] - p3[k]).norm())
        assert moved > 1e-3 and dn < 0.1 * mn and d < 0.5 * moved, (k, d, moved, dn, mn)
    # (c) a state of another architecture is rejected
    bad = dict(ckpt["optimizer_state"], layout=ckpt["optimizer_state"]["layout"][:-1])
    with pytest.raises(RuntimeError, match="optimizer state does not match"):
        model2._engine.load_optimizer_state_dict(bad)
