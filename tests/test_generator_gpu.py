"""GPU parity of the full generator (HIP engine) against the CPU oracle and the committed golden fixtures.

Tolerance (BASELINE.json north_star): per-channel relative MSE <= 1e-3 against the fp32 CPU path; the HIP path
computes in bf16 with f32 accumulation.  Gradients are compared by relative L2 error.
"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

REL_MSE = 1e-3


def _load(cfgname, img, nc, seed, layerscale=0.5):
    from oracle import VIT_CONFIGS, det_state_dict
    from oracle.model import generator_state_shapes
    from miphei_vit_amd.generators import get_vitmatte
    cfg = VIT_CONFIGS[cfgname]
    sd = det_state_dict(generator_state_shapes(cfg, img, nc), seed=seed, layerscale=layerscale)
    p = {k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}
    model = get_vitmatte(cfgname, img, nc, use_lora=True, pretrained=False)
    model.load_state_dict(p)
    return cfg, p, model.cuda()


def _chan_rel_mse(a, b):
    a, b = a.double(), b.double()
    return ((a - b) ** 2).sum(dim=(0, 2, 3)) / (b ** 2).sum(dim=(0, 2, 3))


def _rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float(((a - b) ** 2).sum().sqrt() / (b ** 2).sum().sqrt().clamp_min(1e-30))


@pytest.mark.parametrize("name", ["tiny_gelu_p16_128", "tiny_swiglu_p14_128", "tiny_gelu_p16_256_cfg1",
                                  "tiny_swiglu_p14_256_16ch"])
def test_forward_matches_golden_and_oracle(golden_dir, name):
    from oracle import generator_forward, synth_batch
    g = np.load(os.path.join(golden_dir, f"fwd_{name}.npz"))
    cfgname, img, nc, B, seed = str(g["cfg"]), int(g["img"]), int(g["nc"]), int(g["batch"]), int(g["seed"])
    cfg, p, model = _load(cfgname, img, nc, seed)
    x, _ = synth_batch(seed, B, img, nc)
    st = int(g["out_stride"])
    model.eval()
    with torch.no_grad():
        out = model(x.cuda()).cpu()
        ref, mids = generator_forward(p, x, cfg, nc, training=False, return_mids=True)
    assert out.shape == ref.shape
    assert float(_chan_rel_mse(out, ref).max()) < REL_MSE
    # reference fixture (strided subsample of the reference's own output)
    gold = torch.from_numpy(g["out_eval"])
    sub = out[..., ::st, ::st]
    assert float((((sub - gold) ** 2).sum(dim=(0, 2, 3)) / (gold ** 2).sum(dim=(0, 2, 3))).max()) < REL_MSE
    # train-mode BatchNorm (batch statistics + running-stat update)
    model.train()
    with torch.no_grad():
        out_t = model(x.cuda()).cpu()
    gold_t = torch.from_numpy(g["out_train"])
    sub = out_t[..., ::st, ::st]
    assert float((((sub - gold_t) ** 2).sum(dim=(0, 2, 3)) / (gold_t ** 2).sum(dim=(0, 2, 3))).max()) < REL_MSE
    sd = model.state_dict()
    assert _rel(sd["decoder.fusion_blks.3.conv.bn.running_var"], torch.from_numpy(g["bn_rv_after"])) < 2e-2
    assert _rel(sd["decoder.fusion_blks.3.conv.bn.running_mean"], torch.from_numpy(g["bn_rm_after"])) < 2e-2
    assert _rel(sd["decoder.segmentation_head_0.0.psi.1.running_var"], torch.from_numpy(g["head_bn_rv_after"])) < 2e-2
    assert int(sd["decoder.fusion_blks.0.conv.bn.num_batches_tracked"]) == 1


@pytest.mark.parametrize("cfgname,img,nc,B", [("tiny_swiglu", 128, 3, 2), ("tiny", 128, 16, 2)])
def test_backward_matches_oracle_autograd(cfgname, img, nc, B):
    """loss.backward() through the autograd bridge vs torch autograd on the CPU oracle (same weights / inputs)."""
    from oracle import synth_batch, weighted_mse_loss
    from oracle.model import OracleTrainer, orion_marker_weights
    seed = 11
    cfg, p, model = _load(cfgname, img, nc, seed)
    x, y = synth_batch(seed, B, img, nc)
    w = orion_marker_weights(nc)
    tr = OracleTrainer(p, cfg, nc, batch_size=B, total_iters=100, weights=w)
    out_ref, loss_ref, gref = tr.loss_and_grads(x, y)
    model.train()
    out = model(x.cuda())
    loss = weighted_mse_loss(y.cuda(), out, w.cuda())
    loss.backward()
    assert abs(float(loss) - float(loss_ref)) < 2e-3 * abs(float(loss_ref))
    # yardstick: the same arithmetic under bf16 autocast (the reference's mixed-precision mode) vs fp32 on CPU.
    # The HIP path (bf16 operands, f32 accumulate) must not be noisier than that, per parameter.
    with torch.autocast("cpu", dtype=torch.bfloat16):
        _, _, gac = tr.loss_and_grads(x, y)
    named = dict(model.named_parameters())
    gnorm = float(torch.cat([g.flatten().double() for g in gref.values()]).norm())
    bad = {}
    for k, gr in gref.items():
        got = named[k].grad
        assert got is not None, k
        if float(gr.double().norm()) < 1e-5 * gnorm:   # analytically zero (bias in front of a train-mode BatchNorm)
            assert float(got.double().norm()) < 1e-5 * gnorm, k
            continue
        e_hip, e_ac = _rel(got, gr), _rel(gac[k], gr)
        if e_hip > max(1.25 * e_ac, 0.02):
            bad[k] = (e_hip, e_ac)
    assert not bad, sorted(bad.items(), key=lambda kv: -kv[1][0])[:12]
    keys = [k for k in gref if float(gref[k].double().norm()) >= 1e-5 * gnorm]
    gv = torch.cat([named[k].grad.flatten().cpu().double() for k in keys])
    rv = torch.cat([gref[k].flatten().double() for k in keys])
    av = torch.cat([gac[k].flatten().double() for k in keys])
    cos = float((gv * rv).sum() / (gv.norm() * rv.norm()))
    cos_ac = float((av * rv).sum() / (av.norm() * rv.norm()))
    assert cos > min(0.9995, cos_ac), (cos, cos_ac)
    assert abs(float(gv.norm() / rv.norm()) - 1) < 2e-2


def test_graph_captured_inference_matches_eager():
    """hipGraph replay of the eval forward (BASELINE config 5) gives the eager result, also for a second batch."""
    from oracle import synth_batch
    cfg, p, model = _load("tiny_swiglu", 128, 16, 5)
    model.eval()
    eng = model._engine
    run, x_static, out_static = eng.capture_inference(3)
    for seed in (1, 2):
        x, _ = synth_batch(seed, 3, 128, 16)
        x_static.copy_(x.cuda())
        run()
        torch.cuda.synchronize()
        got = out_static.clone()
        with torch.no_grad():
            ref = model(x.cuda())
        assert torch.equal(got, ref)


def test_eval_batchnorm_fold_matches_the_unfolded_path_and_follows_the_running_statistics():
    """in eval mode the ConvStream's BatchNorm is folded into its convolutions (scaled weights, shift
    as bias, ReLU in the epilogue) and the fusion blocks' scale / shift are constants.  Same result as the unfolded path up to the bf16
    rounding of the pre-activation it no longer stores; the fold is rebuilt when the running statistics change (a training forward)."""
    cfg, p, model = _load("tiny", 128, 3, seed=11)
    from oracle import synth_batch
    x, _ = synth_batch(11, 3, 128, 3)
    x = x.cuda()
    eng = model._engine
    model.eval()
    with torch.no_grad():
        eng.bn_fold = False
        ref = model(x).clone()
        eng.bn_fold = True
        out = model(x).clone()
        assert float(_chan_rel_mse(out, ref).max()) < 2e-4
        assert torch.equal(model(x), out)                     # cached fold: same bits again
        model.train()
        model(x)                                              # train-mode BatchNorm: running statistics move
        model.eval()
        out2 = model(x).clone()
        eng.bn_fold = False
        ref2 = model(x).clone()
    assert float(_chan_rel_mse(out2, ref2).max()) < 2e-4
    assert float(_chan_rel_mse(out2, out).max()) > 1e-6       # the statistics did change the eval output: the fold followed them


def test_eval_batchnorm_fold_is_rebuilt_after_an_in_place_parameter_update():
    """Adam (and a rank-0 broadcast) rewrite the parameters through the flat buffer, which tensor._version does not see: the sequence
    train forward -> eval forward (fold cached) -> adam_step -> eval forward must not serve the old ConvStream fold."""
    cfg, p, model = _load("tiny", 128, 3, seed=12)
    from oracle import synth_batch
    x, y = synth_batch(12, 3, 128, 3)
    x, y = x.cuda(), y.cuda()
    eng = model._engine
    model.train()
    out = model(x)
    (out.float() - y).pow(2).mean().backward()
    model.eval()
    with torch.no_grad():
        before = model(x).clone()                             # builds and caches the fold
    eng.adam_step(lr=5e-2)
    with torch.no_grad():
        after = model(x).clone()
        eng.bn_fold = False
        ref = model(x).clone()
        eng.bn_fold = True
    assert float(_chan_rel_mse(after, ref).max()) < 2e-4
    assert float(_chan_rel_mse(after, before).max()) > 1e-6   # the step did move the output


def test_512_tiles_ragged_tokens():
    """512x512 tiles (N = 36*36+5 = 1301 tokens, regrid 36->32): forward parity vs the oracle (BASELINE config 4 shape)."""
    from oracle import VIT_CONFIGS, det_state_dict, generator_forward, synth_batch
    from oracle.model import generator_state_shapes
    from miphei_vit_amd.generators import get_vitmatte
    cfg = VIT_CONFIGS["tiny_swiglu"]
    sd = det_state_dict(generator_state_shapes(cfg, 512, 3), seed=4, layerscale=0.5)
    p = {k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}
    model = get_vitmatte("tiny_swiglu", 512, 3, use_lora=True, pretrained=False)
    model.load_state_dict(p)
    model.cuda().eval()
    x, _ = synth_batch(4, 1, 512, 3)
    with torch.no_grad():
        out = model(x.cuda()).cpu()
        ref = generator_forward(p, x, cfg, 3, training=False)
    assert float(_chan_rel_mse(out, ref).max()) < REL_MSE


def test_half_precision_eval_convention():
    """evaluation scripts of the reference call generator.eval().cuda().half() and feed x.half() (eval_orion.py:191,214):
    such a model runs on the fp16-operand library (libmiphei_hip_f16.so: v_mfma_f32_*_f16, the weights exactly as the module holds
    them; round 6 -- before, they were re-rounded to bf16 and this test needed 2e-3) and returns the input dtype, within the
    north-star tolerance of the fp32 oracle."""
    from oracle import generator_forward, synth_batch
    from miphei_vit_amd import _lib
    cfg, p, model = _load("tiny_swiglu", 128, 3, 6)
    x, _ = synth_batch(6, 2, 128, 3)
    with torch.no_grad():
        out_bf = model.eval()(x.cuda())                      # same module, fp32 parameters: the bf16-operand library
    model = model.eval().half()
    assert model._engine.operand_mode() == "f16"
    with torch.no_grad():
        out = model(x.cuda().half())
        ref = generator_forward(p, x, cfg, 3, training=False)
    assert "f16" in _lib._libs and _lib.operand_mode() == "bf16"     # the fp16 library was loaded and the mode is per forward
    assert out.dtype == torch.float16
    e16 = float(_chan_rel_mse(out.float().cpu(), ref).max())
    ebf = float(_chan_rel_mse(out_bf.float().cpu(), ref).max())
    assert e16 < REL_MSE, e16                                # fp16-rounded parameters + fp16 operands vs fp32: 1e-3
    assert e16 < ebf, (e16, ebf)                             # 11 mantissa bits against 8
    model = model.float()                                    # and back (parameters now fp16-rounded): the caches follow the dtype
    assert model._engine.operand_mode() == "bf16"
    with torch.no_grad():
        back = model(x.cuda())
    assert back.dtype == torch.float32 and float(_chan_rel_mse(back.cpu(), ref).max()) < REL_MSE


@pytest.mark.parametrize("cfgname,img,pool", [("tiny_swiglu", 126, "token"), ("tiny", 128, "avg"), ("tiny", 128, "")])
def test_registry_model_embeddings(cfgname, img, pool):
    """encoder-only embedding extraction: FOUNDATION_MODEL_REGISTRY[name](img, global_pool=...).eval().cuda().half() called
    on half inputs (reference preprocessings/artifacts_detection/extract_embeddings.py:41-42,78) vs the fp32 oracle ViT."""
    from oracle import VIT_CONFIGS, det_state_dict
    from oracle.vit import vit_forward, vit_state_shapes
    from miphei_vit_amd.generators.foundation_models import FOUNDATION_MODEL_REGISTRY
    cfg = VIT_CONFIGS[cfgname]
    sd = det_state_dict(vit_state_shapes(cfg, img, prefix="", lora=False), seed=5, layerscale=0.5)
    p = {k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}
    model = FOUNDATION_MODEL_REGISTRY[cfgname](img, pretrained=False, global_pool=pool)
    model.load_state_dict(p)
    model = model.eval().cuda().half()
    x = torch.from_numpy(np.random.default_rng(3).standard_normal((3, 3, img, img)).astype(np.float32))
    with torch.no_grad():
        got = model(x.cuda().half())
        tok = vit_forward(p, x.half().float(), cfg, prefix="", lora=False)
    ref = tok[:, 0] if pool == "token" else tok[:, 5:].mean(1) if pool == "avg" else tok
    assert got.dtype == torch.float16 and got.shape == ref.shape
    assert _rel(got.float(), ref) < 5e-3      # fp16 operands (libmiphei_hip_f16.so) / fp16-rounded parameters vs fp32 (bf16 operands: 2e-2)
    with pytest.raises(ValueError):
        model(torch.zeros(1, 3, img + 14, img + 14).cuda().half())


@pytest.mark.parametrize("cfgname,img,nc,B", [("tiny", 128, 1, 1), ("tiny_swiglu", 126 + 2, 5, 3), ("tiny", 256, 16, 1)])
def test_edge_shapes_forward_backward(cfgname, img, nc, B):
    """single-tile batches (train-mode BatchNorm over one sample), a single marker head, odd batch / head counts:
    forward vs the fp32 oracle and a full backward that must stay finite and close on the head / decoder parameters."""
    from oracle import generator_forward, synth_batch, weighted_mse_loss
    from oracle.model import OracleTrainer, orion_marker_weights
    cfg, p, model = _load(cfgname, img, nc, 21)
    x, y = synth_batch(21, B, img, nc)
    w = orion_marker_weights(16)[:nc]
    model.train()
    out = model(x.cuda())
    ref = generator_forward(p, x, cfg, nc, training=True)
    assert float(_chan_rel_mse(out.detach().float().cpu(), ref).max()) < REL_MSE
    loss = weighted_mse_loss(y.cuda(), out, w.cuda())
    loss.backward()
    tr = OracleTrainer(p, cfg, nc, batch_size=B, total_iters=100, weights=w)
    _, loss_ref, gref = tr.loss_and_grads(x, y)
    assert abs(float(loss) - float(loss_ref)) < 2e-3 * abs(float(loss_ref))
    named = dict(model.named_parameters())
    gnorm = float(torch.cat([g.flatten().double() for g in gref.values()]).norm())
    for k, gr in gref.items():
        got = named[k].grad
        assert got is not None and bool(torch.isfinite(got).all()), k
        if k.startswith("decoder.") and float(gr.double().norm()) > 1e-3 * gnorm:
            # bf16-operand noise grows with depth (5-19 % on the first ConvStream conv, the same as the reference's own
            # bf16 autocast mode - see test_backward_matches_oracle_autograd for the yardstick); heads are shallow
            tol = 0.1 if k.startswith("decoder.segmentation_head") else 0.35
            assert _rel(got, gr) < tol, (k, _rel(got, gr))


def test_decoder_only_training_with_a_frozen_encoder():
    """get_vitmatte(use_lora=False): the reference then fine-tunes the whole encoder (outside the LoRA hot path: refused with a message that
    says what to do); with the encoder frozen the decoder trains on fixed features -- its gradients must equal those of the LoRA model whose
    adapters are still at their initial B = 0 (same forward), and one optimiser step must move the decoder only."""
    from oracle import synth_batch
    from oracle.model import orion_marker_weights
    from miphei_vit_amd.generators import get_vitmatte
    from miphei_vit_amd.loss import WeightedMSELoss
    from miphei_vit_amd.models import ModelModule
    cfgname, img, nc, B = "tiny_swiglu", 128, 16, 3
    cfg, p, lora_model = _load(cfgname, img, nc, seed=4)
    for k, v in lora_model.named_parameters():                 # adapters at their initial state: B = 0 -> no contribution to the forward
        if ".lora_" in k and k.endswith(".B"):
            v.data.zero_()
    plain = get_vitmatte(cfgname, img, nc, use_lora=False, pretrained=False)
    sd = {k.replace(".qkv.qkv.", ".qkv."): v for k, v in lora_model.state_dict().items() if ".lora_" not in k}
    plain.load_state_dict(sd)
    plain.cuda()
    x, y = synth_batch(11, B, img, nc)
    batch = {"image": x.cuda(), "target": y.cuda()}
    mk = lambda m: ModelModule(m, None, 1e-3, 0., WeightedMSELoss(50.0, orion_marker_weights(nc)))
    mod_p = mk(plain)
    with pytest.raises(NotImplementedError, match="freeze"):
        mod_p.training_step(batch, 0)                          # every encoder weight trainable: full fine-tuning is refused
    plain.encoder.requires_grad_(False)
    dec0 = {k: v.detach().clone() for k, v in plain.decoder.named_parameters()}
    enc0 = {k: v.detach().clone() for k, v in plain.encoder.named_parameters()}
    mod_p.total_iters, mod_p.global_step_ = 4000, 1000
    loss_p = float(mod_p.training_step(batch, 0))
    mod_l = mk(lora_model)
    mod_l.total_iters, mod_l.global_step_ = 4000, 1000
    loss_l = float(mod_l.training_step(batch, 0))
    assert abs(loss_p - loss_l) < 1e-4 * abs(loss_l)
    gl = dict(lora_model.decoder.named_parameters())
    for k, v in plain.decoder.named_parameters():
        assert v.grad is not None and _rel(v.grad, gl[k].grad) < 1e-3, k
        if float(gl[k].grad.abs().max()) > 1e-12:               # (a bias in front of a BatchNorm has no gradient)
            assert not torch.equal(v.detach(), dec0[k]), k      # the step moved it
    for k, v in plain.encoder.named_parameters():
        assert torch.equal(v.detach(), enc0[k]), k
