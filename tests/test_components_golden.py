"""Component fixtures captured from the reference's own classes (oracle/make_golden_components.py):
QkvWithLoRA forward + gradients (src/generators/lora.py:8-33) and WeightedMSELoss value + gradient (src/loss.py:47-57).
CPU: the oracle restatement must reproduce them; GPU: the HIP kernels the engine uses for the same step must."""
import os

import numpy as np
import pytest
import torch


def _T(seed, name, shape, std=1.0):
    from oracle.detgen import det_normal
    return torch.from_numpy(np.asarray(det_normal(seed, name, shape, 0.0, std), dtype=np.float32))


def _lora_inputs(g):
    seed, B, N, D, r = int(g["seed"]), int(g["B"]), int(g["N"]), int(g["D"]), int(g["rank"])
    return dict(w=_T(seed, "w", (3 * D, D), D ** -0.5), b=_T(seed, "b", (3 * D,), 0.02), Aq=_T(seed, "Aq", (D, r), r ** -0.5),
                Bq=_T(seed, "Bq", (r, D), 0.05), Av=_T(seed, "Av", (D, r), r ** -0.5), Bv=_T(seed, "Bv", (r, D), 0.05),
                x=_T(seed, "x", (B, N, D)), up=_T(seed, "up", (B, N, 3 * D)))


def _rel(a, b):
    a, b = torch.as_tensor(a).double().cpu(), torch.as_tensor(b).double().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def test_oracle_lora_qkv_matches_reference(golden_dir):
    from oracle.vit import lora_qkv
    g = np.load(os.path.join(golden_dir, "comp_lora_qkv.npz"))
    t = {k: v.clone().requires_grad_(k in ("x", "Aq", "Bq", "Av", "Bv")) for k, v in _lora_inputs(g).items()}
    out = lora_qkv(t["x"], t["w"], t["b"], t["Aq"], t["Bq"], t["Av"], t["Bv"], float(g["alpha"]))
    (out * t["up"]).sum().backward()
    assert _rel(out.detach(), g["out"]) < 1e-6
    for k, f in (("x", "dx"), ("Aq", "dAq"), ("Bq", "dBq"), ("Av", "dAv"), ("Bv", "dBv")):
        assert _rel(t[k].grad, g[f]) < 1e-5, k


def test_oracle_wmse_matches_reference(golden_dir):
    from oracle.model import orion_marker_weights, weighted_mse_loss
    g = np.load(os.path.join(golden_dir, "comp_wmse.npz"))
    seed, shape = int(g["seed"]), (int(g["B"]), int(g["C"]), int(g["H"]), int(g["W"]))
    pred = torch.tanh(_T(seed, "pred", shape)).requires_grad_(True)
    target = _T(seed, "target", shape, 0.5).clamp(-0.9, 0.9)
    loss = weighted_mse_loss(target, pred, orion_marker_weights(shape[1]), float(g["lambda_factor"]))
    loss.backward()
    assert abs(float(loss) - float(g["loss"])) < 1e-5 * float(g["loss"])
    assert _rel(pred.grad, g["dpred"]) < 1e-6


@pytest.mark.gpu
def test_hip_lora_qkv_matches_reference(golden_dir):
    """the engine's LoRA path: t = h @ [A_q | A_v] (skinny MFMA), qkv GEMM with the rank-2r K extension, and the backward
    (paired dt, dual-output TN GEMMs for dA / dB, dgrad GEMM with the A extension) on bf16 operands"""
    import miphei_vit_amd.ops as ops
    g = np.load(os.path.join(golden_dir, "comp_lora_qkv.npz"))
    B, N, D, r, alpha = int(g["B"]), int(g["N"]), int(g["D"]), int(g["rank"]), float(g["alpha"])
    t = {k: v.cuda() for k, v in _lora_inputs(g).items()}
    M, bf = B * N, torch.bfloat16
    h = t["x"].reshape(M, D).to(bf).contiguous()
    AcatT = torch.cat([t["Aq"], t["Av"]], 1).t().to(bf).contiguous()          # [2r, D]
    B2 = torch.zeros(3 * D, 2 * r, device="cuda", dtype=bf)
    B2[:D, :r] = (alpha * t["Bq"]).t()
    B2[2 * D:, r:] = (alpha * t["Bv"]).t()
    tt = torch.empty(M, 2 * r, device="cuda", dtype=bf)
    ops.skinny_xw(h, AcatT, tt)
    qkv = torch.empty(M, 3 * D, device="cuda", dtype=bf)
    ops.gemm(h, t["w"].to(bf).contiguous(), qkv, bias=t["b"].contiguous(), a2=tt, b2=B2, K2=2 * r)
    assert _rel(qkv.float().view(B, N, 3 * D), g["out"]) < 1e-2
    # backward
    dqkv = t["up"].reshape(M, 3 * D).to(bf).contiguous()
    Bq16, Bv16 = (alpha * t["Bq"]).to(bf).contiguous(), (alpha * t["Bv"]).to(bf).contiguous()
    dt = torch.empty(M, 2 * r, device="cuda", dtype=bf)
    ops.skinny_xw2(dqkv, Bq16, dt, dqkv.view(-1)[2 * D:], Bv16, dt.view(-1)[r:], ldx=3 * D, ldw=D, ldo=2 * r, M=M, K=D, R=r)
    dAq, dAv = torch.zeros(D, r, device="cuda"), torch.zeros(D, r, device="cuda")
    dBq, dBv = torch.zeros(r, D, device="cuda"), torch.zeros(r, D, device="cuda")
    ops.gemm_tn(tt, dqkv, dBq, M=M, I=2 * r, J=3 * D, lda=2 * r, ldb=3 * D, ldci=D, ldcj=1, msplit=2, c2=dBv, isplit=r, j1=D,
                jlo2=2 * D)
    ops.gemm_tn(dt, h, dAq, M=M, I=2 * r, J=D, lda=2 * r, ldb=D, ldci=1, ldcj=r, msplit=2, c2=dAv, isplit=r)
    dx = torch.empty(M, D, device="cuda", dtype=bf)
    ops.gemm(dqkv, t["w"].t().to(bf).contiguous(), dx, a2=dt, b2=torch.cat([t["Aq"], t["Av"]], 1).to(bf).contiguous(), K2=2 * r)
    torch.cuda.synchronize()
    assert _rel(dx.float().view(B, N, D), g["dx"]) < 1.5e-2
    for got, key in ((alpha * dBq, "dBq"), (alpha * dBv, "dBv"), (dAq, "dAq"), (dAv, "dAv")):
        assert _rel(got, g[key]) < 1.5e-2, key


@pytest.mark.gpu
def test_hip_wmse_matches_reference(golden_dir):
    import miphei_vit_amd.ops as ops
    from oracle.model import orion_marker_weights
    g = np.load(os.path.join(golden_dir, "comp_wmse.npz"))
    seed, shape, lam = int(g["seed"]), (int(g["B"]), int(g["C"]), int(g["H"]), int(g["W"])), float(g["lambda_factor"])
    pred = torch.tanh(_T(seed, "pred", shape)).cuda().contiguous()
    target = _T(seed, "target", shape, 0.5).clamp(-0.9, 0.9).cuda().contiguous()
    acc = torch.zeros(1, device="cuda", dtype=torch.float64)
    dY = torch.empty_like(pred)
    ops.wmse_fwd_bwd(pred, target, orion_marker_weights(shape[1]).cuda(), acc, dY, lam)
    loss = float(acc) * lam / (shape[1] * shape[0] * shape[2] * shape[3])
    assert abs(loss - float(g["loss"])) < 1e-5 * float(g["loss"])
    assert _rel(dY, g["dpred"]) < 1e-5


def _head_params(g):
    seed, NH = int(g["seed"]), int(g["NH"])
    p = {}
    for i in range(NH):
        pre = f"h{i}."
        p[pre + "0.psi.0.weight"] = _T(seed, f"W1_{i}", (16, 32, 1, 1), 0.3)
        p[pre + "0.psi.0.bias"] = _T(seed, f"b1_{i}", (16,), 0.2)
        p[pre + "0.psi.1.weight"] = 1 + _T(seed, f"g_{i}", (16,), 0.3)
        p[pre + "0.psi.1.bias"] = _T(seed, f"be_{i}", (16,), 0.3)
        p[pre + "0.psi.1.running_mean"] = torch.zeros(16)
        p[pre + "0.psi.1.running_var"] = torch.ones(16)
        p[pre + "0.psi.1.num_batches_tracked"] = torch.zeros((), dtype=torch.long)
        p[pre + "0.psi.3.weight"] = _T(seed, f"W2_{i}", (1, 16, 1, 1), 0.5)
        p[pre + "0.psi.3.bias"] = _T(seed, f"b2_{i}", (1,), 0.2)
        p[pre + "1.weight"] = _T(seed, f"W3_{i}", (1, 32, 3, 3), 0.1)
        p[pre + "1.bias"] = _T(seed, f"b3_{i}", (1,), 0.1)
    shape = (int(g["B"]), 32, int(g["H"]), int(g["W"]))
    x = _T(seed, "x", shape).to(torch.bfloat16).float()
    up = _T(seed, "up", (shape[0], NH, shape[2], shape[3]))
    return p, x, up


_HEAD_GRADS = (("dW1", "0.psi.0.weight"), ("db1", "0.psi.0.bias"), ("dg", "0.psi.1.weight"), ("dbe", "0.psi.1.bias"),
               ("dW2", "0.psi.3.weight"), ("db2", "0.psi.3.bias"), ("dW3", "1.weight"), ("db3", "1.bias"))


def test_oracle_heads_match_reference(golden_dir):
    from oracle.decoder import segmentation_head
    g = np.load(os.path.join(golden_dir, "comp_heads.npz"))
    p, x, up = _head_params(g)
    NH = int(g["NH"])
    for k, v in p.items():
        if v.is_floating_point() and "running" not in k:
            v.requires_grad_(True)
    x.requires_grad_(True)
    stats = {}
    out = torch.cat([segmentation_head(p, f"h{i}.", x, True, stats) for i in range(NH)], 1)
    (out * up).sum().backward()
    assert _rel(out.detach(), g["out"]) < 1e-6
    assert _rel(x.grad, g["dx"]) < 1e-5
    for i in range(NH):
        assert _rel(stats[f"h{i}.0.psi.1.running_mean"], g[f"rm_{i}"]) < 1e-5
        assert _rel(stats[f"h{i}.0.psi.1.running_var"], g[f"rv_{i}"]) < 1e-5
        for f, key in _HEAD_GRADS:
            ref = torch.from_numpy(g[f"{f}_{i}"])
            if f == "db1":     # bias in front of a train-mode BatchNorm: analytically zero, numerically noise
                assert float(p[f"h{i}.{key}"].grad.abs().max()) < 1e-4 and float(ref.abs().max()) < 1e-4
            else:
                assert _rel(p[f"h{i}.{key}"].grad, ref) < 1e-4, (i, f)


@pytest.mark.gpu
def test_hip_heads_match_reference(golden_dir):
    """the fused heads sequence of the engine (moments -> BN statistics -> gate -> gated conv, and its backward) against the
    reference's own SegmentationHead modules in train mode: outputs, running statistics and every gradient"""
    import miphei_vit_amd.ops as ops
    g = np.load(os.path.join(golden_dir, "comp_heads.npz"))
    p, x, up = _head_params(g)
    NH, B, H, W = int(g["NH"]), int(g["B"]), int(g["H"]), int(g["W"])
    M, nch, dev = B * H * W, NH * 16, "cuda"
    st = lambda key, shape: torch.stack([p[f"h{i}.{key}"].reshape(-1) for i in range(NH)]).reshape(shape).contiguous().to(dev)
    W1, b1 = st("0.psi.0.weight", (nch, 32)), st("0.psi.0.bias", (nch,))
    gam, bet = st("0.psi.1.weight", (nch,)), st("0.psi.1.bias", (nch,))
    W2, b2 = st("0.psi.3.weight", (nch,)), st("0.psi.3.bias", (NH,))
    W3k = st("1.weight", (NH, 32, 9)).transpose(1, 2).contiguous()               # [NH, 9, 32]
    b3 = st("1.bias", (NH,))
    xh = x.permute(0, 2, 3, 1).reshape(M, 32).to(torch.bfloat16).contiguous().to(dev)
    nslots = 32
    mom = torch.zeros(nslots * 1056, device=dev, dtype=torch.float64)
    mom_sum = torch.zeros(1056, device=dev, dtype=torch.float64)
    rm, rv = torch.zeros(nch, device=dev), torch.ones(nch, device=dev)
    scale, shift, mean, rstd = (torch.empty(nch, device=dev) for _ in range(4))
    G = torch.empty(M, 16, device=dev, dtype=torch.bfloat16)
    out = torch.empty(B, NH, H, W, device=dev)
    ops.heads_moments(xh, mom, M, nslots)
    ops.heads_bn_from_moments(mom, W1, b1, gam, bet, rm, rv, scale, shift, mean, rstd, mom_sum, NH, nslots, M, 1e-5, 0.1, True)
    ops.heads_gate_fwd(xh, W1, b1, scale, shift, W2, b2, G, M, NH)
    ops.heads_conv_fwd(xh, G, W3k, b3, out, B, H, W, NH)
    assert float((out.cpu() - torch.from_numpy(g["out"])).abs().max()) < 2e-2      # bf16 gate between the two stages
    assert _rel(out, g["out"]) < 5e-3
    for i in range(NH):
        assert _rel(rm[16 * i:16 * i + 16], g[f"rm_{i}"]) < 1e-3
        assert _rel(rv[16 * i:16 * i + 16], g[f"rv_{i}"]) < 1e-3
    # backward
    dY = up.to(dev).contiguous()
    cs = torch.empty(ops.heads_conv_bwd_scratch_bytes(M) // 4 + 1, device=dev)
    gs = torch.empty(ops.heads_gate_bwd_scratch_bytes() // 4, device=dev)
    dG, dXc = torch.empty(M, 16, device=dev), torch.empty(M, 32, device=dev)
    dW3, db3 = torch.empty(NH * 9, 32, device=dev), torch.zeros(64, 32, device=dev)
    dW1, dgam, dbet, dW2, db2 = (torch.zeros(n, device=dev) for n in (nch * 32, nch, nch, nch, NH))
    dF = torch.empty(M, 32, device=dev, dtype=torch.bfloat16)
    ops.heads_conv_bwd(dY, out, xh, G, W3k, cs, dG, dXc, dW3, db3, B, H, W, NH)
    ops.heads_gate_bwd(xh, G, dG, dXc, W1, b1, scale, shift, mean, rstd, gam, W2, mom_sum, gs, dW1, dgam, dbet, dW2, db2, dF, M, NH)
    torch.cuda.synchronize()
    tol = 3e-2
    assert _rel(dF.float().view(B, H, W, 32).permute(0, 3, 1, 2), g["dx"]) < tol
    cat = lambda f: torch.cat([torch.from_numpy(g[f"{f}_{i}"]).reshape(-1) for i in range(NH)])
    assert _rel(dW1, cat("dW1")) < tol
    assert _rel(dgam, cat("dg")) < tol
    assert _rel(dbet, cat("dbe")) < tol
    assert _rel(dW2, cat("dW2")) < tol
    assert _rel(db2, cat("db2")) < tol
    assert _rel(db3.sum(0)[:NH], cat("db3")) < tol
    ref_dW3 = torch.stack([torch.from_numpy(g[f"dW3_{i}"]).reshape(32, 9).t() for i in range(NH)]).reshape(NH * 9, 32)
    assert _rel(dW3, ref_dW3) < tol
