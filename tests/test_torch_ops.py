"""torch.library registration of the per-kernel entry points.  CPU: the ops exist, carry fake (meta)
implementations and refuse CPU tensors.  GPU: a transformer block written the timm way calls them op by op and matches plain
PyTorch fp32 math, forward and backward."""
import pytest
import torch
import torch.nn.functional as F


def test_ops_registered_with_fake_impls_and_no_cpu_kernel():
    import miphei_vit_amd.torch_ops as T
    from torch._subclasses.fake_tensor import FakeTensorMode
    for name in ("linear", "layer_norm", "attention_forward", "weighted_mse_forward", "linear_backward", "layer_norm_backward",
                 "attention_backward"):
        assert hasattr(torch.ops.mvit, name), name
    with FakeTensorMode():
        x = torch.empty(2, 329, 96, device="cuda", dtype=torch.bfloat16)
        w = torch.empty(288, 96, device="cuda", dtype=torch.bfloat16)
        qkv = torch.ops.mvit.linear(x, w, None)
        assert qkv.shape == (2, 329, 288) and qkv.dtype == torch.bfloat16
        assert T.attention(qkv, 3).shape == (2, 329, 96)
        xf = torch.empty(2, 329, 96, device="cuda")
        assert torch.ops.mvit.linear(xf, w, None).dtype == torch.float32
        h = torch.ops.mvit.layer_norm(xf, torch.empty(96, device="cuda"), torch.empty(96, device="cuda"), 1e-6)
        assert h.shape == xf.shape and h.dtype == torch.bfloat16
        assert T.weighted_mse(torch.empty(2, 3, 8, 8, device="cuda"), torch.empty(2, 3, 8, 8, device="cuda"),
                              torch.empty(3, device="cuda"), 50.0).shape == ()
    with pytest.raises(NotImplementedError):          # no CPU fallback: the dispatcher has no CPU kernel for these ops
        torch.ops.mvit.linear(torch.zeros(2, 8), torch.zeros(4, 8), None)


def _rel(a, b):
    a, b = a.double(), b.double()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


@pytest.mark.gpu
def test_reference_style_block_through_custom_ops_matches_fp32_autograd():
    import miphei_vit_amd.torch_ops as T
    torch.manual_seed(0)
    B, N, D, H, Hid = 2, 329, 96, 3, 256
    dev = "cuda"
    p = {k: torch.randn(*s, device=dev) * sc for k, (s, sc) in dict(
        n1w=((D,), 0.1), n1b=((D,), 0.1), wqkv=((3 * D, D), D ** -0.5), bqkv=((3 * D,), 0.1), wproj=((D, D), D ** -0.5),
        bproj=((D,), 0.1), n2w=((D,), 0.1), n2b=((D,), 0.1), w1=((Hid, D), D ** -0.5), b1=((Hid,), 0.1),
        w2=((D, Hid), Hid ** -0.5), b2=((D,), 0.1)).items()}
    p["n1w"] += 1
    p["n2w"] += 1
    x0 = torch.randn(B, N, D, device=dev)

    def block(x, q, lin, ln, attn):
        h = ln(x, q["n1w"], q["n1b"])
        a = attn(lin(h, q["wqkv"], q["bqkv"]))
        x = x + lin(a, q["wproj"], q["bproj"]).float()
        h = ln(x, q["n2w"], q["n2b"])
        return x + lin(F.gelu(lin(h, q["w1"], q["b1"]).float()).to(h.dtype), q["w2"], q["b2"]).float()

    def ref_attn(qkv):
        q, k, v = qkv.view(B, N, 3, H, D // H).permute(2, 0, 3, 1, 4)
        return F.scaled_dot_product_attention(q, k, v).transpose(1, 2).reshape(B, N, D)

    pr = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    xr = x0.clone().requires_grad_(True)
    yr = block(xr, pr, F.linear, lambda t, w, b: F.layer_norm(t, (D,), w, b, 1e-6), ref_attn)
    tgt = torch.randn_like(yr)
    ((yr - tgt) ** 2).mean().backward()

    ph = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    xh = x0.clone().requires_grad_(True)
    yh = block(xh, ph, lambda t, w, b: torch.ops.mvit.linear(t, w.to(torch.bfloat16), b),
               lambda t, w, b: torch.ops.mvit.layer_norm(t, w, b, 1e-6), lambda t: T.attention(t, H))
    ((yh - tgt) ** 2).mean().backward()
    assert _rel(yh, yr) < 1e-2
    assert _rel(xh.grad, xr.grad) < 3e-2
    for k in p:
        assert ph[k].grad is not None and _rel(ph[k].grad, pr[k].grad) < 5e-2, k


@pytest.mark.gpu
def test_weighted_mse_op_matches_reference_loss_and_opcheck():
    import miphei_vit_amd.torch_ops as T
    from miphei_vit_amd.loss import WeightedMSELoss
    torch.manual_seed(1)
    pred = torch.randn(2, 5, 32, 32, device="cuda", requires_grad=True)
    tgt = torch.randn(2, 5, 32, 32, device="cuda")
    w = torch.rand(5, device="cuda") + 0.5
    ref = WeightedMSELoss(50.0, w)(tgt, pred)
    g_ref, = torch.autograd.grad(ref, pred)
    p2 = pred.detach().clone().requires_grad_(True)
    got = T.weighted_mse(p2, tgt, w, 50.0)
    g_got, = torch.autograd.grad(got, p2)
    assert abs(float(got) - float(ref)) < 1e-5 * abs(float(ref)) and _rel(g_got, g_ref) < 1e-5
    torch.library.opcheck(torch.ops.mvit.layer_norm.default,
                          (torch.randn(4, 64, device="cuda"), torch.ones(64, device="cuda"), torch.zeros(64, device="cuda"), 1e-6),
                          test_utils=("test_schema", "test_faketensor"))
