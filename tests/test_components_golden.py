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
