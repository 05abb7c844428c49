"""Fused output-head gate kernels (moments -> BN statistics -> gate forward / backward on MFMA) against a plain fp32
torch restatement of AttentionBlock (reference: src/generators/unet.py:407-427, one block per marker at
src/generators/mipheivit.py:198-205).  Floating-point path: bf16 operands, fp32 accumulation -> tolerances below."""
import pytest
import torch

pytestmark = pytest.mark.gpu

XC, HC = 32, 16


def _ref(x, W1, b1, gamma, beta, W2, b2, NH, dG=None):
    x = x.clone().requires_grad_(True)
    ps = [t.clone().requires_grad_(True) for t in (W1, b1, gamma, beta, W2, b2)]
    W1, b1, gamma, beta, W2, b2 = ps
    u = x @ W1.t() + b1
    mean, var = u.mean(0), u.var(0, unbiased=False)
    a = (u - mean) * torch.rsqrt(var + 1e-5) * gamma + beta
    r = torch.relu(a).view(-1, NH, HC)
    g = torch.sigmoid((r * W2.view(NH, HC)).sum(-1) + b2)
    if dG is None:
        return g.detach()
    (g * dG[:, :NH]).sum().backward()
    return g.detach(), x.grad, [p.grad for p in ps]


@pytest.mark.parametrize("NH,M", [(16, 4096 + 37), (3, 1000), (5, 33)])
def test_gate_forward_backward(NH, M):
    import miphei_vit_amd.ops as ops
    torch.manual_seed(NH * 1000 + M)
    dev = "cuda"
    nch = NH * HC
    x = (torch.randn(M, XC, device=dev) * 1.5 + 0.3).bfloat16()
    W1 = torch.randn(nch, XC, device=dev) * 0.3
    b1 = torch.randn(nch, device=dev) * 0.2
    gamma = 1 + 0.3 * torch.randn(nch, device=dev)
    beta = 0.3 * torch.randn(nch, device=dev)
    W2 = torch.randn(nch, device=dev) * 0.5
    b2 = torch.randn(NH, device=dev) * 0.2
    dG = torch.zeros(M, 16, device=dev)
    dG[:, :NH] = torch.randn(M, NH, device=dev)
    dXc = torch.randn(M, XC, device=dev) * 0.1

    nslots = 32
    mom = torch.zeros(nslots * (32 + 1024), device=dev, dtype=torch.float64)
    mom_sum = torch.zeros(32 + 1024, device=dev, dtype=torch.float64)
    rm, rv = torch.zeros(nch, device=dev), torch.ones(nch, device=dev)
    scale, shift, mean, rstd = (torch.empty(nch, device=dev) for _ in range(4))
    ops.heads_moments(x, mom, M, nslots)
    ops.heads_bn_from_moments(mom, W1, b1, gamma, beta, rm, rv, scale, shift, mean, rstd, mom_sum, NH, nslots, M, 1e-5, 0.1, True)
    xf = x.float()
    ms = mom_sum.float()
    assert torch.allclose(ms[:32], xf.sum(0), rtol=1e-4, atol=1e-2)
    assert torch.allclose(ms[32:].view(32, 32), xf.t() @ xf, rtol=1e-4, atol=1e-1)
    u = xf @ W1.t() + b1
    assert torch.allclose(mean, u.mean(0), rtol=1e-4, atol=1e-4)
    assert torch.allclose(rstd, torch.rsqrt(u.var(0, unbiased=False) + 1e-5), rtol=1e-3)
    assert torch.allclose(rm, 0.1 * u.mean(0), rtol=1e-3, atol=1e-5)   # running statistics, momentum 0.1

    G = torch.empty(M, 16, device=dev, dtype=torch.bfloat16)
    ops.heads_gate_fwd(x, W1, b1, scale, shift, W2, b2, G, M, NH)
    g_ref, dx_ref, grads = _ref(xf, W1, b1, gamma, beta, W2, b2, NH, dG)
    assert float((G[:, :NH].float() - g_ref).abs().max()) < 2e-2      # bf16 output + bf16 operands
    assert float((G[:, :NH].float() - g_ref).abs().mean()) < 2e-3
    assert NH == 16 or float(G[:, NH:].float().abs().max()) == 0.0

    scratch = torch.empty(ops.heads_gate_bwd_scratch_bytes() // 4, device=dev)
    dW1, dgam, dbet, dW2 = torch.zeros(nch, XC, device=dev), torch.zeros(nch, device=dev), torch.zeros(nch, device=dev), torch.zeros(nch, device=dev)
    db2 = torch.zeros(NH, device=dev)
    dF = torch.empty(M, XC, device=dev, dtype=torch.bfloat16)
    # the kernels differentiate through the stored bf16 gate; feed the reference's dG unchanged
    ops.heads_gate_bwd(x, G, dG, dXc, W1, b1, scale, shift, mean, rstd, gamma, W2, mom_sum, scratch, dW1, dgam, dbet, dW2, db2,
                       dF, M, NH)
    torch.cuda.synchronize()

    def rel(a, b):
        return float((a - b).norm() / (b.norm() + 1e-12))

    dW1_ref, db1_ref, dgam_ref, dbet_ref, dW2_ref, db2_ref = grads
    tol = 3e-2 if M > 500 else 6e-2   # bf16 operand noise; tiny batches have fewer pixels to average it over
    assert rel(dF.float(), dx_ref + dXc) < tol
    assert rel(dW1, dW1_ref) < tol
    assert rel(dgam, dgam_ref) < tol
    assert rel(dbet, dbet_ref) < tol
    assert rel(dW2, dW2_ref) < tol
    assert rel(db2, db2_ref) < tol
    assert float(db1_ref.abs().max()) < 1e-2 * float(dbet_ref.abs().max() + 1e-6)  # conv bias before train-mode BN: zero gradient


@pytest.mark.parametrize("NH,B,H,W", [(16, 2, 40, 40), (3, 1, 24, 31), (5, 3, 9, 7)])
def test_gated_conv_backward(NH, B, H, W):
    """backward of y = tanh(conv3x3(x * g_h) + b3) per head (SegmentationHead, src/generators/unet.py:430-438) against
    torch autograd in fp32: dG (gate gradient), dXc (feature gradient through the conv), dW3, db3."""
    import torch.nn.functional as F
    import miphei_vit_amd.ops as ops
    torch.manual_seed(NH + B * 7 + H)
    dev = "cuda"
    M = B * H * W
    x = torch.randn(M, XC, device=dev).bfloat16()
    G = torch.zeros(M, 16, device=dev)
    G[:, :NH] = torch.rand(M, NH, device=dev)
    G = G.bfloat16()
    W3 = torch.randn(NH, 9, XC, device=dev) * 0.1          # kernel layout [head][ky*3+kx][channel]
    b3 = torch.randn(NH, device=dev) * 0.1
    dY = torch.randn(B, NH, H, W, device=dev)

    xr = x.float().view(B, H, W, XC).permute(0, 3, 1, 2).clone().requires_grad_(True)
    gr = G.float().view(B, H, W, 16).permute(0, 3, 1, 2)[:, :NH].clone().requires_grad_(True)
    w = W3.view(NH, 3, 3, XC).permute(0, 3, 1, 2).clone().requires_grad_(True)     # [head, c, ky, kx]
    bb = b3.clone().requires_grad_(True)
    ys = [torch.tanh(F.conv2d(xr * gr[:, h:h + 1], w[h:h + 1], bb[h:h + 1], padding=1)) for h in range(NH)]
    Y = torch.cat(ys, 1)
    (Y * dY).sum().backward()

    out = torch.empty(B, NH, H, W, device=dev)
    ops.heads_conv_fwd(x, G, W3, b3, out, B, H, W, NH)
    assert float((out - Y.detach()).abs().max()) < 1e-3

    scratch = torch.empty(ops.heads_conv_bwd_scratch_bytes(M) // 4 + 1, device=dev)
    dG = torch.empty(M, 16, device=dev)
    dXc = torch.empty(M, XC, device=dev)
    dW3 = torch.empty(NH * 9, XC, device=dev)
    db3 = torch.zeros(64, 32, device=dev)
    ops.heads_conv_bwd(dY, out, x, G, W3, scratch, dG, dXc, dW3, db3, B, H, W, NH)
    torch.cuda.synchronize()

    def rel(a, b):
        return float((a - b).norm() / (b.norm() + 1e-12))

    dG_ref = gr.grad.permute(0, 2, 3, 1).reshape(M, NH)
    dX_ref = xr.grad.permute(0, 2, 3, 1).reshape(M, XC)
    dW_ref = w.grad.permute(0, 2, 3, 1).reshape(NH * 9, XC)
    assert rel(dG[:, :NH], dG_ref) < 1e-2       # bf16 dz / W3 operands, fp32 accumulation
    assert rel(dXc, dX_ref) < 1e-2
    assert rel(dW3, dW_ref) < 1e-2
    assert rel(db3.sum(0)[:NH], bb.grad) < 1e-3
