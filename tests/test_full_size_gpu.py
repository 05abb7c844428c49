"""Parity at BASELINE.json's full model size (configs[1]: H-Optimus-0 ViT-g/14 + LoRA + ViTMatte decoder, 256x256, 16 markers):
the 1.14 B-parameter HIP path against the fp32 CPU oracle on one synthetic tile pair - outputs within the north-star tolerance
(1e-3 relative MSE per channel) and the training loss / gradient norm of the 6.7 M trainable parameters."""
import pytest
import torch

pytestmark = pytest.mark.gpu


# worst per-parameter relative gradient error vs the fp32 oracle, per class: measured on the round-5 build (see the printout of
# the test) + 25 %.  Round 6 (one-pass attention backward), three repeats of this test on one box (tools/abl/ceil_reps.sh): the same
# values to four digits every time, B = 2 and B = 16 -- the worst tensor of a class does not move with the f32-atomic orderings
# (B = 2, round-5 build: decoder 0.158 -- convstream.convs.0.conv.weight, autocast 0.196 --, LoRA block 0 / 20 / 39: 0.056 / 0.053 / 0.054)
ABS_CEIL = {2: {"decoder": 0.20, "lora0": 0.070, "lora20": 0.067, "lora39": 0.067},
            # (B = 16: decoder 0.110, LoRA block 0 / 20 / 39: 0.025 / 0.064 / 0.038; autocast on the same tensors 0.244 / 0.026 / 0.069 / 0.048)
            16: {"decoder": 0.14, "lora0": 0.031, "lora20": 0.080, "lora39": 0.047}}


# (16, 256) = BASELINE.json configs[1] itself: 256-row GEMM tiles, M = 5264 (about 40 s of CPU oracle);
# (2, 512) = the shape of configs[3]: 1301 tokens, 36 -> 32 regrid, decoder at 512 x 512
@pytest.mark.parametrize("B,img", [(2, 256), (16, 256), (2, 512)])
def test_hoptimus0_forward_loss_gradnorm_vs_oracle(B, img):
    import bench
    from oracle import VIT_CONFIGS
    from oracle.model import OracleTrainer, orion_marker_weights
    from miphei_vit_amd.generators import get_vitmatte
    from miphei_vit_amd.loss import WeightedMSELoss
    nc = 16
    dev = torch.device("cuda:0")
    with torch.device(dev):
        model = get_vitmatte("hoptimus0", img, nc, use_lora=True, pretrained=False)
    bench.synthetic_init_(model, seed=3)
    if img == 256:
        assert sum(p.numel() for p in model.parameters()) == 1_141_576_432       # SURVEY.md section 8 a1 [probe]
    assert sum(p.numel() for p in model.parameters() if p.requires_grad) == 6_697_712
    x, y = bench.synthetic_batch(77, B, img, nc, dev)
    w = orion_marker_weights(nc)
    torch.set_num_threads(min(32, torch.get_num_threads()))
    p = {k: v.detach().to("cpu", torch.float32) for k, v in model.state_dict().items()}
    tr = OracleTrainer(p, VIT_CONFIGS["hoptimus0"], nc, batch_size=B, total_iters=1000, weights=w)
    tr.p = p
    out_ref, loss_ref, gref = tr.loss_and_grads(x.cpu(), y.cpu())
    model.train()
    out = model(x)
    loss = WeightedMSELoss(50.0, w).to(dev)(y_true=y, y_pred=out)
    loss.backward()
    o = out.detach().float().cpu()
    rel = ((o - out_ref) ** 2).sum(dim=(0, 2, 3)) / (out_ref ** 2).sum(dim=(0, 2, 3))
    assert float(rel.max()) < 1e-3, rel
    assert abs(float(loss.detach()) - float(loss_ref)) < 2e-3 * abs(float(loss_ref))
    named = dict(model.named_parameters())
    gn_hip = torch.cat([named[k].grad.flatten().double().cpu() for k in gref]).norm()
    gn_ref = torch.cat([g.flatten().double() for g in gref.values()]).norm()
    assert abs(float(gn_hip) - float(gn_ref)) < 0.05 * float(gn_ref), (float(gn_hip), float(gn_ref))
    # the shallow end of the backward pass (heads, last fusion block) is only a few bf16 roundings away from fp32
    for k in ("decoder.segmentation_head_3.1.weight", "decoder.segmentation_head_0.0.psi.3.weight", "decoder.fusion_blks.3.conv.bn.weight"):
        g1, g0 = named[k].grad.double().cpu(), gref[k].double()
        assert float((g1 - g0).norm() / g0.norm()) < 0.05, k
    if img != 256:
        return
    # (round 4: also at B = 16, the tile path the benchmark runs -- 256-row dgrad tiles, three attention row blocks per pair, the
    # batched LoRA weight-gradient products over groups of 10 blocks)
    # Per-parameter gradients at full depth: LoRA A / B of blocks 0, 20, 39 (the far end, the middle and the near end of the
    # 40-block backward chain: a wrong stride or a dropped term in ANY block's dgrad chain shows up at block 0) and EVERY decoder
    # gradient.  Yardstick: the same arithmetic under bf16 autocast on the CPU (the reference's mixed-precision mode,
    # train.precision) against fp32 -- the HIP path (bf16 operands, f32 accumulate) must be no noisier than that per parameter
    # and point the same way (/root/reference/src/generators/lora.py:16-33, src/models.py:128-138).
    with torch.autocast("cpu", dtype=torch.bfloat16):
        _, _, gac = tr.loss_and_grads(x.cpu(), y.cpu())
    rel = lambda a, b: float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30))
    cosf = lambda a, b: float((a.double().flatten() @ b.double().flatten()) / (a.double().norm() * b.double().norm()).clamp_min(1e-30))
    gnorm = float(gn_ref)
    keys = [k for k in gref if k.startswith("decoder.")]
    for l in (0, 20, 39):
        keys += [f"encoder.vit.blocks.{l}.attn.qkv.lora_{qv}.{ab}" for qv in "qv" for ab in "AB"]
    bad, rows = {}, []
    for k in keys:
        gr, got = gref[k], named[k].grad.detach().cpu()
        if ".lora_" not in k and float(gr.double().norm()) < 1e-5 * gnorm:   # analytically zero (conv bias in front of a train-mode BatchNorm)
            assert float(got.double().norm()) < 1e-5 * gnorm, k
            continue
        # (LoRA tensors are never skipped: every criterion below is relative to the tensor's own norm, however small a share of the
        # global norm a deep adapter's gradient is)
        e_hip, e_ac, c_hip, c_ac = rel(got, gr), rel(gac[k], gr), cosf(got, gr), cosf(gac[k], gr)
        rows.append((k, e_hip, e_ac, c_hip, c_ac))
        if ".lora_" in k:
            # yardstick of an adapter tensor: the noisiest autocast gradient among the four tensors of the same block's adapters --
            # they hang off the same d(qkv), whose bf16 noise is what both implementations carry (at B = 16 block 0's dA_q came
            # out at 1.9 % / 2.2 % in two runs -- BatchNorm statistics meet in f32 atomics -- against 1.2 % for its own autocast
            # gradient and 2.4-2.6 % for dB_q / dA_v / dB_v of the same block)
            blk = k.rsplit(".attn.qkv.", 1)[0]
            e_ac = max(rel(gac[kk], gref[kk]) for kk in gref if kk.startswith(blk + ".attn.qkv.lora_"))
        if e_hip > max(1.25 * e_ac, 0.02) or c_hip < min(0.999, c_ac - 0.002):
            bad[k] = (round(e_hip, 4), round(e_ac, 4), round(c_hip, 5), round(c_ac, 5))
    lora = [r for r in rows if ".lora_" in r[0]]
    print("LoRA gradients (rel err HIP, rel err autocast, cos HIP, cos autocast):")
    for r in lora:
        print("  %-48s %.4f %.4f %.5f %.5f" % r)
    worst = max(rows, key=lambda r: r[1] / max(r[2], 0.016))
    print("worst decoder/LoRA ratio: %s %.4f vs %.4f" % worst[:3])
    assert len(lora) == 12 and not bad, sorted(bad.items(), key=lambda kv: -kv[1][0])[:12]
    # Absolute ceilings next to the autocast-relative bound (round 5): the relative bound lets any tensor sit at 2 % without a
    # word, so a drift from e.g. 1.2 % to 1.9 % would pass silently.  The worst relative error per parameter class is printed and
    # bounded by the values measured on the round-5 build + 25 % (ABS_CEIL below, per batch size: the B = 16 step meets its
    # BatchNorm statistics in f32 atomics and runs the batched LoRA products, so its run-to-run spread is wider).
    classes = {"decoder": [r for r in rows if r[0].startswith("decoder.")]}
    for l in (0, 20, 39):
        classes[f"lora{l}"] = [r for r in rows if f".blocks.{l}.attn.qkv.lora_" in r[0]]
    worst_abs = {c: max(rs, key=lambda r: r[1]) for c, rs in classes.items()}
    print("worst absolute relative error per class (B = %d):" % B)
    for c, r in worst_abs.items():
        print("  %-8s %-52s %.4f (autocast %.4f, ceiling %.4f)" % (c, r[0], r[1], r[2], ABS_CEIL[B][c]))
    over = {c: (r[0], round(r[1], 4)) for c, r in worst_abs.items() if r[1] > ABS_CEIL[B][c]}
    assert not over, over


@pytest.mark.parametrize("B", [16, 64])   # 64 = BASELINE configs[4] (inference): fc1 runs the 256x256 tile there
def test_hoptimus0_batch16_tiles_agree_with_batch2_chunks(B):
    """The batch-16 step runs the 256-row GEMM tiles (M = 5264) and three attention row blocks per pair; the oracle-checked
    batch-2 run above runs the 128-row tiles (M = 658).  In eval mode (running BatchNorm statistics) a tile's prediction does
    not depend on its batch, so the two paths must agree tile by tile, to bf16 rounding."""
    import bench
    from miphei_vit_amd.generators import get_vitmatte
    nc, img = 16, 256
    dev = torch.device("cuda:0")
    with torch.device(dev):
        model = get_vitmatte("hoptimus0", img, nc, use_lora=True, pretrained=False)
    bench.synthetic_init_(model, seed=5)
    model.eval()
    x, _ = bench.synthetic_batch(11, B, img, nc, dev)
    with torch.no_grad():
        full = model(x).float()
        parts = torch.cat([model(x[i:i + 2]).float() for i in range(0, B, 2)])
    rel = ((full - parts) ** 2).sum(dim=(0, 2, 3)) / (parts ** 2).sum(dim=(0, 2, 3))
    assert float(rel.max()) < 2e-4, rel     # both are bf16 paths: different tile shapes = different summation order only


def test_hoptimus0_half_precision_eval_forward_vs_oracle():
    """The reference's evaluation convention at the full model size: generator.eval().cuda().half() on x.half()
    (/root/reference/evaluation/eval_orion.py:191, 214-215) runs on the fp16-operand library (v_mfma_f32_*_f16); output vs the fp32 CPU
    oracle on the fp32 parameters within the north-star tolerance, and closer to it than the bf16-operand forward of the same module."""
    import bench
    from oracle import VIT_CONFIGS, generator_forward
    from miphei_vit_amd.generators import get_vitmatte
    nc, img, B = 16, 256, 2
    dev = torch.device("cuda:0")
    with torch.device(dev):
        model = get_vitmatte("hoptimus0", img, nc, use_lora=True, pretrained=False)
    bench.synthetic_init_(model, seed=9)
    model.eval()
    x, _ = bench.synthetic_batch(31, B, img, nc, dev)
    torch.set_num_threads(min(32, torch.get_num_threads()))
    p = {k: v.detach().to("cpu", torch.float32) for k, v in model.state_dict().items()}
    with torch.no_grad():
        out_bf = model(x).float().cpu()
        ref = generator_forward(p, x.cpu(), VIT_CONFIGS["hoptimus0"], nc, training=False)
        model.half()
        assert model._engine.operand_mode() == "f16"
        out = model(x.half())
    assert out.dtype == torch.float16 and torch.isfinite(out).all()
    rel = lambda a: float((((a - ref) ** 2).sum(dim=(0, 2, 3)) / (ref ** 2).sum(dim=(0, 2, 3))).max())
    e16, ebf = rel(out.float().cpu()), rel(out_bf)
    print(f"worst-channel relative MSE vs the fp32 oracle: fp16 operands {e16:.3e}, bf16 operands {ebf:.3e}")
    assert e16 < 1e-3 and e16 < ebf, (e16, ebf)


def test_hoptimus0_batch64_hipgraph_replay_equals_eager():
    """BASELINE configs[4] at size: the hipGraph-captured batch-64 forward of H-Optimus-0 at 256 x 256 (merged LoRA, eval-mode
    BatchNorm, the 256 x 256 / wave-specialised GEMM tiles of the inference path) returns exactly the eager result, for two
    different batches through the same captured graph (reference call: /root/reference/src/models.py:75-79, predict_step)."""
    import bench
    from miphei_vit_amd.generators import get_vitmatte
    nc, img, B = 16, 256, 64
    dev = torch.device("cuda:0")
    with torch.device(dev):
        model = get_vitmatte("hoptimus0", img, nc, use_lora=True, pretrained=False)
    bench.synthetic_init_(model, seed=6)
    model.eval()
    run, x_static, out_static = model._engine.capture_inference(B)
    outs = []
    for seed in (21, 22):
        x, _ = bench.synthetic_batch(seed, B, img, nc, dev)
        x_static.copy_(x)
        run()
        torch.cuda.synchronize()
        got = out_static.clone()
        with torch.no_grad():
            ref = model(x)
        assert got.shape == (B, nc, img, img) and torch.isfinite(got).all()
        assert torch.equal(got, ref), float((got.float() - ref.float()).abs().max())
        outs.append(got)
    # and the two batches were different inputs giving different outputs (the graph did not replay a stale buffer)
    assert not torch.equal(outs[0], outs[1])


def test_hoptimus0_three_fused_training_steps_vs_oracle():
    """ModelModule.training_step (fused HIP step: loss, clip-norm 1.0, Adam(0.5, 0.999, 1e-7), LambdaLR warm-up) on the full
    H-Optimus-0 configuration against the oracle's restatement of models.py:87-143, three steps: loss, gradient norm, learning
    rate per step, BatchNorm running statistics and the direction the decoder weights moved."""
    import bench
    from oracle import VIT_CONFIGS
    from oracle.model import OracleTrainer, orion_marker_weights
    from miphei_vit_amd.generators import get_vitmatte
    from miphei_vit_amd.loss import WeightedMSELoss
    from miphei_vit_amd.models import ModelModule
    nc, B, img, lr_g = 16, 2, 256, 2e-4 * 2 ** 0.5
    dev = torch.device("cuda:0")
    with torch.device(dev):
        model = get_vitmatte("hoptimus0", img, nc, use_lora=True, pretrained=False)
    bench.synthetic_init_(model, seed=9)
    w = orion_marker_weights(nc)
    p0 = {k: v.detach().to("cpu", torch.float32).clone() for k, v in model.state_dict().items()}
    mod = ModelModule(model, None, lr_g, 0., WeightedMSELoss(50.0, w)).to(dev)
    mod.total_iters = 1000
    torch.set_num_threads(min(32, torch.get_num_threads()))
    tr = OracleTrainer(p0, VIT_CONFIGS["hoptimus0"], nc, batch_size=B, total_iters=1000, weights=w)
    tr.base_lr = lr_g
    for it in range(3):
        x, y = bench.synthetic_batch(500 + it, B, img, nc, dev)
        assert abs(mod.current_lr() - tr.base_lr * (it / 400)) < 1e-12          # linear warm-up, utils.py:217-230
        loss = float(mod.training_step({"image": x, "target": y}, it))
        r = tr.step(x.cpu(), y.cpu())
        assert abs(loss - r["loss"]) < 3e-3 * r["loss"], (it, loss, r["loss"])
        gn = float(torch.sqrt(model._engine._saved.w.sqn[0]))
        assert abs(gn - r["grad_norm"]) < 0.05 * r["grad_norm"], (it, gn, r["grad_norm"])
    sd = {k: v.detach().float().cpu() for k, v in model.state_dict().items() if k.startswith("decoder.")}
    for k in ("decoder.fusion_blks.3.conv.bn.running_mean", "decoder.fusion_blks.0.conv.bn.running_var", "decoder.convstream.convs.0.bn.running_mean"):
        assert float((sd[k] - tr.p[k]).norm() / tr.p[k].norm()) < 2e-2, k
    # steps 1 and 2 moved the weights (step 0 has lr 0): same direction as the oracle for the shallow, low-noise parameters
    for k in ("decoder.fusion_blks.3.conv.conv.weight", "decoder.segmentation_head_5.1.weight", "decoder.fusion_blks.2.conv.bn.weight"):
        d_ref, d_got = (tr.p[k] - p0[k]).flatten().double(), (sd[k] - p0[k]).flatten().double()
        cos = float((d_ref @ d_got) / (d_ref.norm() * d_got.norm()))
        assert cos > 0.8 and 0.7 < float(d_got.norm() / d_ref.norm()) < 1.4, (k, cos)
