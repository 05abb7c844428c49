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


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["tiny4_gelu_p16_128", "tiny4_swiglu_p14_128"])
def test_hip_unetr_backward_matches_oracle_autograd(golden_dir, name):
    """loss.backward() through the UNETR autograd bridge vs torch autograd on the CPU oracle (same weights / inputs); yardstick as in
    test_generator_gpu: per parameter no noisier than 1.25x the same arithmetic under bf16 autocast (floor 2 %)."""
    from oracle import synth_batch, weighted_mse_loss
    from oracle.model import orion_marker_weights
    from oracle.unetr import unetr_forward
    from miphei_vit_amd.generators.unet import Unet
    g, cfg, p, img, nc, B, seed = _load(golden_dir, name)
    model = Unet(img, str(g["cfg"]), use_lora=True, classes=nc, pretrained=False)
    model.load_state_dict(p)
    model = model.cuda().train()
    x, y = synth_batch(seed, B, img, nc)
    w = orion_marker_weights(16)[:nc]
    out = model(x.cuda())
    loss = weighted_mse_loss(y.cuda(), out, w.cuda())
    loss.backward()
    named = dict(model.named_parameters())
    train_keys = [k for k, v in named.items() if v.requires_grad]
    assert any("lora" in k for k in train_keys) and not any(k.endswith("attn.proj.weight") for k in train_keys)

    def ref_grads(autocast):
        q = {k: (v.clone().requires_grad_(True) if k in train_keys else v.clone()) for k, v in p.items()}
        with torch.autocast("cpu", dtype=torch.bfloat16, enabled=autocast):
            o = unetr_forward(q, x, cfg, nc, training=True)
        l = weighted_mse_loss(y, o.float(), w)
        l.backward()
        return float(l), {k: q[k].grad for k in train_keys}

    loss_ref, gref = ref_grads(False)
    _, gac = ref_grads(True)
    assert abs(float(loss) - loss_ref) < 2e-3 * abs(loss_ref)
    gnorm = float(torch.cat([v.flatten().double() for v in gref.values() if v is not None]).norm())
    bad = {}
    for k in train_keys:
        got, gr = named[k].grad, gref[k]
        assert got is not None, k
        if gr is None or float(gr.double().norm()) < 1e-5 * gnorm:       # conv biases in front of train-mode BatchNorm
            assert float(got.double().norm()) < 1e-4 * gnorm, k
            continue
        e_hip, e_ac = _rel(got.cpu(), gr), _rel(gac[k], gr)
        if e_hip > max(1.25 * e_ac, 0.02):
            bad[k] = (round(e_hip, 4), round(e_ac, 4))
    assert not bad, sorted(bad.items(), key=lambda kv: -kv[1][0])[:12]


@pytest.mark.gpu
def test_unetr_training_steps_track_oracle(golden_dir):
    """three ModelModule.training_step iterations of the UNETR baseline (autograd bridge + torch clip / Adam / pix2pix LR, the
    reference's own sequence models.py:87-143) against the same loop on the CPU oracle"""
    from oracle import synth_batch, weighted_mse_loss
    from oracle.model import orion_marker_weights, pix2pix_lr_lambda
    from oracle.unetr import unetr_forward
    from miphei_vit_amd.generators.unet import Unet
    from miphei_vit_amd.loss import WeightedMSELoss
    from miphei_vit_amd.models import ModelModule
    g, cfg, p, img, nc, B, seed = _load(golden_dir, "tiny4_swiglu_p14_128")
    model = Unet(img, str(g["cfg"]), use_lora=True, classes=nc, pretrained=False)
    model.load_state_dict(p)
    model = model.cuda()
    w = orion_marker_weights(16)[:nc]
    total, lr = 1000, 0.05            # warm-up scales the step: lr * step / 400
    mod = ModelModule(model, None, lr, 0.0, WeightedMSELoss(50.0, w)).cuda()
    mod.total_iters = total
    train_keys = [k for k, v in model.named_parameters() if v.requires_grad]
    q = {k: (v.clone().requires_grad_(True) if k in train_keys else v.clone()) for k, v in p.items()}
    opt = torch.optim.Adam([q[k] for k in train_keys], lr=lr, betas=(0.5, 0.999), eps=1e-7)
    sched = torch.optim.lr_scheduler.LambdaLR(opt, lr_lambda=lambda s: pix2pix_lr_lambda(s, total, 400, total // 2))
    losses_hip, losses_ref = [], []
    for step in range(3):
        x, y = synth_batch(seed + step, B, img, nc)
        losses_hip.append(float(mod.training_step({"image": x.cuda(), "target": y.cuda()}, step)))
        stats = {}
        out = unetr_forward(q, x, cfg, nc, training=True, new_stats=stats)
        loss = weighted_mse_loss(y, out, w)
        opt.zero_grad()
        loss.backward()
        torch.nn.utils.clip_grad_norm_([q[k] for k in train_keys], 1.0)
        opt.step()
        sched.step()
        with torch.no_grad():
            for k, v in stats.items():
                q[k] = v
        losses_ref.append(float(loss))
    for a, b in zip(losses_hip, losses_ref):
        assert abs(a - b) < 5e-3 * abs(b), (losses_hip, losses_ref)
    sd = model.state_dict()
    for k in ("decoder.decoder0_header.2.weight", "encoder.feature_upsampler.upsampler1.2.block.0.weight",
              "encoder.model.blocks.1.attn.qkv.lora_v.B", "segmentation_head_1.1.weight"):
        d_ref = q[k].detach() - p[k]
        d_hip = sd[k].cpu() - p[k]
        assert float(d_ref.norm()) > 0
        # Adam's first steps are sign-like (lr * m / sqrt(v)): bf16 gradient noise on near-zero components flips whole steps,
        # so the displacement is compared by direction, the gradients themselves are compared in the test above
        cos = float((d_hip.double() * d_ref.double()).sum() / (d_hip.double().norm() * d_ref.double().norm()))
        assert cos > 0.8, (k, cos)
    assert _rel(sd["decoder.decoder0_header.1.block.1.running_var"].cpu(), q["decoder.decoder0_header.1.block.1.running_var"]) < 2e-2


@pytest.mark.gpu
def test_unetr_half_eval_and_errors(golden_dir):
    """evaluation convention of the reference scripts (generator.eval().cuda().half() on half inputs) and the error surface"""
    from oracle import synth_batch
    from miphei_vit_amd.generators import get_generator
    from miphei_vit_amd.generators.unet import Unet
    g, cfg, p, img, nc, B, seed = _load(golden_dir, "tiny4_gelu_p16_128")
    model = Unet(img, str(g["cfg"]), use_lora=True, classes=nc, pretrained=False)
    model.load_state_dict(p)
    model = model.eval().cuda().half()
    x, _ = synth_batch(seed, B, img, nc)
    with torch.no_grad():
        out = model(x.cuda().half())
    assert out.dtype == torch.float16
    ref = torch.as_tensor(g["out_eval"])
    assert float((((out.float().cpu() - ref) ** 2).sum(dim=(0, 2, 3)) / (ref ** 2).sum(dim=(0, 2, 3))).max()) < 2e-3
    with pytest.raises(ValueError):
        model(torch.zeros(1, 3, img * 2, img * 2, device="cuda").half())
    conf = {"model": {"encoder": {"encoder_name": "tiny", "encoder_weights": None, "pretrained": False}, "dropout": 0.0},
            "train": {"foreground_head": False}}
    with pytest.raises(ValueError):                        # depth-2 ViT: "Vit Should have a depth higher than 3" (unet.py:137)
        get_generator("unet_lora", 128, 3, 3, conf)
    with pytest.raises(NotImplementedError):
        Unet(128, "restnet50_lunit_swav", classes=3, pretrained=False)


def _oracle_prefix(layer):
    """engine layer name -> block prefix of the reference state dict"""
    up = "encoder.feature_upsampler."
    if layer in ("s0", "s1"):
        return f"{up}convsteam.{layer[1]}."
    if layer.startswith("u"):                      # "u0.2.c": upsampler0, Deconv2DBlock 2 (its conv part)
        a, k, _ = layer[1:].split(".")
        return f"{up}upsampler{a}.{k}."
    d, k = layer[1:].split(".")                    # "d3.1": decoder3_upsampler.1 / "d0.k": decoder0_header.k
    return f"decoder.decoder{d}_" + ("header" if d == "0" else "upsampler") + f".{k}."


@pytest.mark.gpu
def test_unetr_train_mode_dropout_and_drop_path_match_oracle_with_the_same_masks(golden_dir):
    """The reference trains the UNETR baseline with model.dropout = 0.1 (configs/model/unet.yaml:2): nn.Dropout behind every block's
    ReLU, timm DropPath in the ViT blocks.  The HIP masks are counter-based (seed, layer, element) and the DropPath factors are kept
    with the saved activations, so the oracle can be run with exactly the masks the kernels applied: outputs, loss and gradients
    must then agree as in the deterministic tests.  Eval mode is unaffected by the rate."""
    from oracle import synth_batch, weighted_mse_loss
    from oracle.model import orion_marker_weights
    from oracle.unetr import unetr_forward
    from miphei_vit_amd import ops
    from miphei_vit_amd.generators.unet import Unet
    g, cfg, p, img, nc, B, seed = _load(golden_dir, "tiny4_swiglu_p14_128")
    rate = 0.3
    torch.manual_seed(1234)
    model = Unet(img, str(g["cfg"]), use_lora=True, classes=nc, pretrained=False, drop_rate=rate)
    model.load_state_dict(p)
    model = model.cuda().train()
    x, y = synth_batch(seed, B, img, nc)
    w = orion_marker_weights(16)[:nc]
    out = model(x.cuda())
    loss = weighted_mse_loss(y.cuda(), out, w.cuda())
    loss.backward()
    sv = model._engine._saved
    dpath = sv.we.dpath                                        # [L, 2, M] per-row factors of this step
    assert dpath is not None and dpath.shape[:2] == (cfg.depth, 2)
    dp = dpath[:, :, ::cfg.tokens(img)].cpu()                  # one factor per sample
    assert torch.equal(dpath.view(cfg.depth, 2, B, -1)[..., :1].expand(-1, -1, -1, cfg.tokens(img)).reshape(dpath.shape), dpath)
    keep_last = 1.0 - rate
    assert float(dp[0].min()) == 1.0                           # block 0: rate 0 (linspace(0, rate, depth))
    assert all(v == 0.0 or abs(v - 1.0 / keep_last) < 1e-6 for v in dp[-1].flatten().tolist())
    seeds = {_oracle_prefix(k): (v.drop_seed, v.drop_p) for k, v in sv.st.items() if hasattr(v, "drop_seed")}
    assert len(seeds) == 17 and len({s for s, _ in seeds.values()}) == 17 and all(abs(pp - rate) < 1e-9 for _, pp in seeds.values())
    zeros = []

    def drop(prefix, act):
        s, pp = seeds[prefix]
        Bc, C, H, W = act.shape
        m = ops.dropout_keep_mask(s, Bc * H * W * C, pp).view(Bc, H, W, C).permute(0, 3, 1, 2)
        zeros.append(float((m == 0).float().mean()))
        return m

    train_keys = [k for k, v in model.named_parameters() if v.requires_grad]

    def ref(autocast):
        q = {k: (v.clone().requires_grad_(True) if k in train_keys else v.clone()) for k, v in p.items()}
        with torch.autocast("cpu", dtype=torch.bfloat16, enabled=autocast):
            o = unetr_forward(q, x, cfg, nc, training=True, drop=drop, drop_path=dp)
        l = weighted_mse_loss(y, o.float(), w)
        l.backward()
        return o.detach().float(), float(l), {k: q[k].grad for k in train_keys}

    o_ref, l_ref, gref = ref(False)
    _, _, gac = ref(True)                                        # the same arithmetic in the reference's bf16-mixed mode
    assert all(abs(z - rate) < 0.02 for z in zeros), zeros       # the masks drop ~rate of the elements
    rel = ((out.detach().float().cpu() - o_ref) ** 2).sum(dim=(0, 2, 3)) / (o_ref ** 2).sum(dim=(0, 2, 3))
    assert float(rel.max()) < 2e-3, rel
    assert abs(float(loss) - l_ref) < 3e-3 * abs(l_ref)
    named = dict(model.named_parameters())
    gnorm = float(torch.cat([v.flatten().double() for v in gref.values() if v is not None]).norm())
    worst = {}
    for k in train_keys:
        gr = gref[k]
        if gr is None or float(gr.double().norm()) < 1e-3 * gnorm:
            continue
        e_hip, e_ac = _rel(named[k].grad.cpu(), gr), _rel(gac[k], gr)
        # yardstick of the deterministic test: no noisier than 1.25x bf16 autocast; a wrong mask in a backward kernel would show
        # as an error of the order of the rate (0.3), far above either
        if e_hip > max(1.25 * e_ac, 0.02):
            worst[k] = (round(e_hip, 4), round(e_ac, 4))
    assert not worst, sorted(worst.items(), key=lambda kv: -kv[1][0])[:10]
    # a second step draws different masks; eval mode ignores the rate
    out2 = model(x.cuda())
    assert float((out2 - out).abs().max()) > 1e-3
    model.eval()
    ref_model = Unet(img, str(g["cfg"]), use_lora=True, classes=nc, pretrained=False, drop_rate=0.0)
    ref_model.load_state_dict(model.state_dict())
    ref_model = ref_model.cuda().eval()
    with torch.no_grad():
        assert torch.equal(model(x.cuda()), ref_model(x.cuda()))


def test_get_generator_unet_lora_accepts_the_shipped_dropout():
    """`run.py +default_configs=unetr` composes model.dropout = 0.1 (as the reference's model/unet.yaml): the generator must build."""
    import os
    from miphei_vit_amd.config import compose
    from miphei_vit_amd.generators import get_generator
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = compose(os.path.join(root, "configs"), ["+default_configs=unetr", "++model.encoder.encoder_name=tiny4",
                                                  "++model.encoder.pretrained=false"])
    assert cfg.model.dropout == 0.1 and cfg.model.model_name == "unet_lora"
    gen = get_generator(cfg.model.model_name, 128, 3, 3, cfg)
    assert gen.decoder.drop_rate == 0.1 and gen.encoder.model.drop_path_rate == 0.1
    assert any(isinstance(m, torch.nn.Dropout) and m.p == 0.1 for m in gen.modules())
    with pytest.raises(ValueError):
        from miphei_vit_amd.generators.unet import Unet
        Unet(128, "tiny4", classes=3, pretrained=False, drop_rate=1.0)


@pytest.mark.gpu
def test_unetr_fused_step_matches_autograd_bridge_and_hipgraph_inference(golden_dir):
    """ModelModule.training_step on the UNETR baseline runs the fused sequence (flat gradient buffers, one global-norm clip + Adam
    kernel pair): same parameters after a step as the reference's own sequence through the autograd bridge + torch clip / Adam;
    the optimiser state survives a re-flatten; the eval forward replays from a hipGraph."""
    from oracle import synth_batch
    from oracle.model import orion_marker_weights
    from miphei_vit_amd.generators.unet import Unet
    from miphei_vit_amd.loss import WeightedMSELoss
    from miphei_vit_amd.models import ModelModule
    g, cfg, p, img, nc, B, seed = _load(golden_dir, "tiny4_swiglu_p14_128")
    w = orion_marker_weights(16)[:nc]

    def make():
        model = Unet(img, str(g["cfg"]), use_lora=True, classes=nc, pretrained=False)
        model.load_state_dict(p)
        mod = ModelModule(model.cuda(), None, 0.05, 0.0, WeightedMSELoss(50.0, w)).cuda()
        mod.total_iters = 1000             # warm-up: lr = 0.05 * step / 400 (the torch LambdaLR of the bridge path starts at step 0 too)
        return mod, model

    x, y = synth_batch(seed, B, img, nc)
    batch = {"image": x.cuda(), "target": y.cuda()}
    fused, mf = make()
    ref, mr = make()
    for it in range(2):                    # step 0 runs at lr 0, step 1 moves the parameters
        l1 = float(fused.training_step(batch, it))
        l2 = float(ref._training_step_autograd(batch["image"], batch["target"]))   # autograd bridge + torch clip + torch Adam
        assert abs(l1 - l2) < 1e-4 * abs(l2)
    assert mf._engine._flat is not None and mf._engine._flat.step == 2          # the fused path ran
    a, b = dict(mf.named_parameters()), dict(mr.named_parameters())
    for k in a:
        if not a[k].requires_grad:
            continue
        moved = float((b[k].detach().cpu() - p[k]).double().norm())
        d = float((a[k].detach() - b[k].detach()).double().norm())
        # same gradients, same Adam rule; Adam's sign-like first steps turn the f32-atomics noise of near-zero gradient
        # components into +-lr flips of single elements, so the displacement is compared in norm
        assert d <= 0.2 * moved + 1e-7, (k, d, moved)
    assert float((b["decoder.decoder0_header.2.weight"].detach().cpu() - p["decoder.decoder0_header.2.weight"]).abs().max()) > 1e-5
    # state survives .cuda() / load_state_dict, second step continues with step count 2
    mf.cuda()
    assert mf._engine._flat is None
    fused.training_step(batch, 2)
    assert mf._engine._flat.step == 3 and mf._engine._encoder_engine()._flat.step == 3
    sd = mf._engine.optimizer_state_dict()
    assert sd["step"] == 3 and sd["exp_avg"].numel() == sum(n for _, n in sd["layout"])
    # hipGraph inference
    mf.eval()
    run, xs, out_static = mf._engine.capture_inference(B)
    xs.copy_(batch["image"])
    run()
    with torch.no_grad():
        eager = mf(batch["image"])
    assert torch.equal(out_static, eager)
