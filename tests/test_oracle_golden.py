"""CPU: pin the oracle against fixtures captured from the reference's own modules
(oracle/make_golden.py) and the timm arithmetic against Hugging Face Dinov2WithRegisters."""
import os

import numpy as np
import pytest
import torch

from oracle import VIT_CONFIGS, det_state_dict, generator_forward, synth_batch, weighted_mse_loss
from oracle.model import OracleTrainer, generator_state_shapes, orion_marker_weights, pix2pix_lr_lambda
from oracle.vit import ViTConfig, vit_forward, vit_state_shapes

REL = 1e-5  # fp32 CPU restatement vs fp32 CPU reference (different op order -> ~1e-6)


def _rel(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.sqrt(((a - b) ** 2).sum() / max((b ** 2).sum(), 1e-30)))


def _load_sd(cfgname, img, nc, seed):
    cfg = VIT_CONFIGS[cfgname]
    sd = det_state_dict(generator_state_shapes(cfg, img, nc), seed=seed, layerscale=0.5)
    return cfg, {k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}


FWD = ["tiny_gelu_p16_128", "tiny_swiglu_p14_128", "tiny_gelu_p16_256_cfg1", "tiny_swiglu_p14_256_16ch"]


@pytest.mark.parametrize("name", FWD)
def test_forward_matches_reference(golden_dir, name):
    g = np.load(os.path.join(golden_dir, f"fwd_{name}.npz"))
    cfgname, img, nc, B, seed = str(g["cfg"]), int(g["img"]), int(g["nc"]), int(g["batch"]), int(g["seed"])
    cfg, p = _load_sd(cfgname, img, nc, seed)
    assert sorted(p.keys()) == list(g["keys"])
    x, y = synth_batch(seed, B, img, nc)
    st = int(g["out_stride"])
    with torch.no_grad():
        out, mids = generator_forward(p, x, cfg, nc, training=False, return_mids=True)
        tok = mids["tokens"]
        assert _rel(tok.numpy()[:, :: max(1, tok.shape[1] // 16)], g["tokens_eval"]) < REL
        assert _rel(out.numpy()[..., ::st, ::st], g["out_eval"]) < REL
        assert abs(float(weighted_mse_loss(y, out, orion_marker_weights(nc))) - float(g["loss_eval"])) < 1e-4 * float(g["loss_eval"])
        ns = {}
        out_t = generator_forward(p, x, cfg, nc, training=True, new_stats=ns)
        assert _rel(out_t.numpy()[..., ::st, ::st], g["out_train"]) < REL
        assert _rel(ns["decoder.fusion_blks.3.conv.bn.running_var"].numpy(), g["bn_rv_after"]) < REL
        assert _rel(ns["decoder.fusion_blks.3.conv.bn.running_mean"].numpy(), g["bn_rm_after"]) < REL
        assert _rel(ns["decoder.segmentation_head_0.0.psi.1.running_var"].numpy(), g["head_bn_rv_after"]) < REL
    m = out.double().mean(dim=(0, 2, 3)).numpy()
    assert np.allclose(m, g["out_eval_mean"], atol=1e-5)


@pytest.mark.parametrize("name", ["tiny_gelu_p16_128", "tiny_swiglu_p14_128"])
def test_training_steps_match_reference(golden_dir, name):
    g = np.load(os.path.join(golden_dir, f"train_{name}.npz"))
    cfgname, img, nc, B, seed = str(g["cfg"]), int(g["img"]), int(g["nc"]), int(g["batch"]), int(g["seed"])
    cfg, p = _load_sd(cfgname, img, nc, seed)
    tr = OracleTrainer(p, cfg, nc, batch_size=B, total_iters=int(g["total_iters"]))
    tr.base_lr = float(g["lr_g"])
    assert int(g["n_grads"]) == len(tr.keys)
    for it in range(3):
        x, y = synth_batch(seed * 100 + it, B, img, nc)
        r = tr.step(x, y)
        assert abs(r["loss"] - g["losses"][it]) < 2e-4 * g["losses"][it]
        assert abs(r["grad_norm"] - g["grad_norms"][it]) < 1e-3 * g["grad_norms"][it]
        assert abs(r["lr"] - g["lrs"][it]) < 1e-12
        if it == 0:
            coef = min(1.0, 1.0 / (r["grad_norm"] + 1e-6))
            for k in g["watch"]:
                ref = g["gradclip0::" + str(k)]
                mine = (r["grads"][str(k)] * coef).numpy()
                if mine.size > 20000:
                    mine = mine.reshape(-1)[::37]
                assert _rel(mine, ref) < 1e-3, k
    for k in g["watch"]:
        ref = g["after3::" + str(k)]
        mine = tr.p[str(k)].numpy()
        if mine.size > 20000:
            mine = mine.reshape(-1)[::37]
        assert _rel(mine, ref) < 1e-4, k


def test_state_dict_contract(golden_dir):
    g = np.load(os.path.join(golden_dir, "keys_f256.npz"))
    shapes = generator_state_shapes(VIT_CONFIGS["hoptimus0"], 256, 16)
    assert len(shapes) == int(g["n_tensors"]) == 945
    assert sorted(shapes.keys()) == list(g["keys"])
    n_total = sum(int(np.prod(v)) if len(v) else 1 for k, v in shapes.items()
                  if not k.endswith(("running_mean", "running_var", "num_batches_tracked")))
    assert n_total == 1141576432  # SURVEY.md §0.5 [probe]
    n_train = sum(int(np.prod(v)) for k, v in shapes.items()
                  if (".lora_" in k) or (k.startswith("decoder.") and k.endswith(("weight", "bias"))))
    assert n_train == 6697712


def test_lr_schedule():
    assert pix2pix_lr_lambda(0, 1000) == 0.0
    assert pix2pix_lr_lambda(200, 1000) == 0.5
    assert pix2pix_lr_lambda(450, 1000) == 1.0
    assert pix2pix_lr_lambda(750, 1000) == 0.5
    assert pix2pix_lr_lambda(1000, 1000) == 0.0


@pytest.mark.parametrize("swiglu", [True, False])
def test_vit_matches_hf_dinov2_with_registers(swiglu):
    """timm arithmetic vs an independent implementation on shared weights."""
    tr = pytest.importorskip("transformers")
    from transformers import Dinov2WithRegistersConfig, Dinov2WithRegistersModel

    D, L, Hh, img, patch = 96, 2, 3, 56, 14
    hidden = 512 if swiglu else 384
    cfg = ViTConfig(patch=patch, dim=D, depth=L, heads=Hh, mlp="swiglu" if swiglu else "gelu", hidden=hidden)
    shapes = vit_state_shapes(cfg, img, prefix="", lora=False)
    p = {k: torch.from_numpy(v) for k, v in det_state_dict(shapes, seed=5, layerscale=0.7).items()}
    hc = Dinov2WithRegistersConfig(hidden_size=D, num_hidden_layers=L, num_attention_heads=Hh, image_size=img,
                                   patch_size=patch, num_register_tokens=4, use_swiglu_ffn=swiglu,
                                   mlp_ratio=int(hidden // D) if not swiglu else 4, layer_norm_eps=1e-6,
                                   layerscale_value=1.0, attn_implementation="eager")
    m = Dinov2WithRegistersModel(hc).eval()
    sd = m.state_dict()
    new = {}
    g = img // patch
    for k, v in sd.items():
        new[k] = v.clone()
    new["embeddings.cls_token"] = p["cls_token"]
    new["embeddings.register_tokens"] = p["reg_token"]
    new["embeddings.mask_token"] = sd["embeddings.mask_token"]
    # HF adds a position embedding to the cls token too; timm (no_embed_class) does not -> zero it
    new["embeddings.position_embeddings"] = torch.cat([torch.zeros(1, 1, D), p["pos_embed"]], 1)
    new["embeddings.patch_embeddings.projection.weight"] = p["patch_embed.proj.weight"]
    new["embeddings.patch_embeddings.projection.bias"] = p["patch_embed.proj.bias"]
    for i in range(L):
        a, b = f"encoder.layer.{i}.", f"blocks.{i}."
        new[a + "norm1.weight"], new[a + "norm1.bias"] = p[b + "norm1.weight"], p[b + "norm1.bias"]
        new[a + "norm2.weight"], new[a + "norm2.bias"] = p[b + "norm2.weight"], p[b + "norm2.bias"]
        W, bb = p[b + "attn.qkv.weight"], p[b + "attn.qkv.bias"]
        for j, n in enumerate(("query", "key", "value")):
            new[a + f"attention.attention.{n}.weight"] = W[j * D:(j + 1) * D]
            new[a + f"attention.attention.{n}.bias"] = bb[j * D:(j + 1) * D]
        new[a + "attention.output.dense.weight"] = p[b + "attn.proj.weight"]
        new[a + "attention.output.dense.bias"] = p[b + "attn.proj.bias"]
        new[a + "layer_scale1.lambda1"] = p[b + "ls1.gamma"]
        new[a + "layer_scale2.lambda1"] = p[b + "ls2.gamma"]
        if swiglu:
            if sd[a + "mlp.weights_in.weight"].shape != p[b + "mlp.fc1.weight"].shape:
                pytest.skip("HF SwiGLU hidden size differs for this D")
            new[a + "mlp.weights_in.weight"], new[a + "mlp.weights_in.bias"] = p[b + "mlp.fc1.weight"], p[b + "mlp.fc1.bias"]
            new[a + "mlp.weights_out.weight"], new[a + "mlp.weights_out.bias"] = p[b + "mlp.fc2.weight"], p[b + "mlp.fc2.bias"]
        else:
            new[a + "mlp.fc1.weight"], new[a + "mlp.fc1.bias"] = p[b + "mlp.fc1.weight"], p[b + "mlp.fc1.bias"]
            new[a + "mlp.fc2.weight"], new[a + "mlp.fc2.bias"] = p[b + "mlp.fc2.weight"], p[b + "mlp.fc2.bias"]
    new["layernorm.weight"], new["layernorm.bias"] = p["norm.weight"], p["norm.bias"]
    m.load_state_dict(new)
    x = torch.from_numpy(np.asarray(det_state_dict({"x.weight": (2, 3, img, img)}, seed=9)["x.weight"])) * 20
    with torch.no_grad():
        ref = m(pixel_values=x).last_hidden_state
        mine = vit_forward(p, x, cfg, prefix="", lora=False)
    assert _rel(mine.numpy(), ref.numpy()) < 1e-5


def test_oracle_cells_match_reference_fixture(golden_dir):
    """MeanCellExtrator (scale 1 / 0.5 / 0.25, target=None) and CellMetrics.update state, captured from the reference's classes."""
    from oracle.cells import cell_metrics_update, extract_means
    from oracle.detgen import det_normal
    g = np.load(os.path.join(golden_dir, "comp_cells.npz"))
    seed, B, C, H, W = (int(g[k]) for k in ("seed", "B", "C", "H", "W"))
    T = lambda name, shape, std=1.0: np.asarray(det_normal(seed, name, shape, 0.0, std), dtype=np.float32)
    pred, target = np.tanh(T("pred", (B, C, H, W))), np.clip(T("target", (B, C, H, W), 0.5), -0.9, 0.9)
    pred = torch.tanh(torch.from_numpy(T("pred", (B, C, H, W)))).numpy()    # the same tanh the fixture script applied
    nuclei = g["nuclei"]
    for tag, sf in (("s1", 1.0), ("s05", 0.5), ("s025", 0.25)):
        pm, tm, ids, _ = extract_means(pred, target, nuclei, sf)
        assert np.array_equal(ids, g[f"ids_{tag}"])                          # labels and their order: exact
        assert np.allclose(pm, g[f"pm_{tag}"], atol=2e-6) and np.allclose(tm, g[f"tm_{tag}"], atol=2e-6)
    pm, tm, _, _ = extract_means(pred, None, nuclei[:, None], 1.0)
    assert np.allclose(pm, g["pm_notarget"], atol=2e-6) and not tm.any() and not g["tm_notarget"].any()
    st = cell_metrics_update(pred, nuclei, [1, 2, 3, 4])
    slides = ["slideA", "slideB", "slideA", "slideB"]
    seen = {"slideA": 0, "slideB": 0}
    for b, rec in enumerate(st):
        if rec is None:
            continue
        s, i = slides[b], seen[slides[b]]
        seen[s] += 1
        assert np.array_equal(rec[0], g[f"cm_{s}_id{i}"]) and np.array_equal(rec[2], g[f"cm_{s}_area{i}"])
        assert np.abs(rec[1] - g[f"cm_{s}_sum{i}"]).max() <= 1               # truncation of sums*255 to uint32: f32 summation order
    assert seen == {"slideA": int(g["cm_slideA_n"]), "slideB": int(g["cm_slideB_n"])}


def test_input_stage_restatement_matches_reference_fixture(golden_dir):
    """oracle/io.py against NormalizationLayer of the reference (fixture from oracle/make_golden_io.py): bit for bit."""
    from oracle.io import normalize_he, normalize_if, spatial_augment
    g = np.load(os.path.join(golden_dir, "comp_io.npz"))
    assert np.array_equal(normalize_he(g["rgb"], g["mean"], g["std"]), g["he"])
    assert np.array_equal(normalize_if(g["mif"]), g["mif_norm"])
    assert np.allclose(g["he_unorm"], g["rgb"], atol=2e-5) and np.allclose(g["if_unorm"], g["mif"], atol=1e-4)
    d = dict(oy=3, ox=5, hflip=1, vflip=1, drop=1, y1=2, x1=4, hh=6, hw=7)
    a = spatial_augment(g["rgb"], d, (32, 40))
    assert a.shape == (32, 40, 3) and (a[2:8, 4:11] == 0).all()
    assert np.array_equal(a[0, 0], g["rgb"][3 + 31, 5 + 39]) and np.array_equal(a[31, 39], g["rgb"][3, 5])
