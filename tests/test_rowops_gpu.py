"""GPU parity of the row-wise encoder kernels against plain PyTorch fp32 math."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _ops():
    import miphei_vit_amd.ops as ops
    return ops


def _rel(a, b):
    a, b = a.double(), b.double()
    return float(((a - b) ** 2).sum().sqrt() / b.pow(2).sum().sqrt().clamp_min(1e-30))


def _rand(*shape, scale=1.0, seed=0):
    g = torch.Generator(device="cuda").manual_seed(seed)
    return torch.randn(*shape, generator=g, device="cuda") * scale


@pytest.mark.parametrize("M,D", [(7, 64), (329 * 2, 96), (1000, 1536)])
def test_layernorm_fwd_bwd(M, D):
    ops = _ops()
    x = _rand(M, D, seed=1) * 3 + 0.5
    w, b = _rand(D, seed=2) * 0.2 + 1, _rand(D, seed=3) * 0.1
    out = torch.empty(M, D, device="cuda", dtype=torch.bfloat16)
    ops.layernorm_fwd(x, w, b, out, 1e-6)
    ref = F.layer_norm(x, (D,), w, b, 1e-6)
    assert _rel(out.float(), ref) < 4e-3
    dh = _rand(M, D, seed=4).bfloat16()
    xg = x.clone().requires_grad_(True)
    F.layer_norm(xg, (D,), w, b, 1e-6).backward(dh.float())
    dx0 = _rand(M, D, seed=5)
    dx = dx0.clone()
    gam = _rand(D, seed=6)
    dy = torch.empty(M, D, device="cuda", dtype=torch.bfloat16)
    ops.layernorm_bwd(dh, x, w, dx, gam, dy, 1e-6, True)
    assert _rel(dx, dx0 + xg.grad) < 1e-5
    assert _rel(dy.float(), gam * (dx0 + xg.grad)) < 4e-3
    dx2 = torch.empty_like(dx)
    ops.layernorm_bwd(dh, x, w, dx2, None, None, 1e-6, False)
    assert _rel(dx2, xg.grad) < 1e-5


def test_skinny():
    ops = _ops()
    M, K, R = 700, 1536, 16
    X = _rand(M, K, seed=1).bfloat16()
    W = _rand(R, K, seed=2, scale=0.1).bfloat16()
    out = torch.empty(M, R, device="cuda", dtype=torch.bfloat16)
    ops.skinny_xw(X, W, out)
    assert _rel(out.float(), X.float() @ W.float().t()) < 4e-3
    # R = 8 into a column slice, X a column window of a wider matrix, ragged K
    Xw = _rand(M, 3 * 104, seed=7).bfloat16()
    Bm = _rand(8, 104, seed=3, scale=0.1).bfloat16()
    out2 = torch.zeros(M, 16, device="cuda", dtype=torch.bfloat16)
    ops.skinny_xw(Xw.view(-1)[2 * 104:], Bm, out2.view(-1)[8:], ldx=3 * 104, ldo=16, M=M)
    assert _rel(out2[:, 8:].float(), Xw[:, 208:].float() @ Bm.float().t()) < 4e-3 and float(out2[:, :8].abs().max()) == 0
    # the paired launch (both LoRA adapters): q columns -> out[:, :8], v columns -> out[:, 8:]
    Bq = _rand(8, 104, seed=4, scale=0.1).bfloat16()
    out3 = torch.zeros(M, 16, device="cuda", dtype=torch.bfloat16)
    ops.skinny_xw2(Xw, Bq, out3, Xw.view(-1)[2 * 104:], Bm, out3.view(-1)[8:], ldx=3 * 104, ldw=104, ldo=16, M=M, K=104, R=8)
    assert _rel(out3[:, :8].float(), Xw[:, :104].float() @ Bq.float().t()) < 4e-3
    assert _rel(out3[:, 8:].float(), Xw[:, 208:].float() @ Bm.float().t()) < 4e-3


def test_patch_prefix_cast():
    ops = _ops()
    B, S, p = 2, 128, 14
    g = S // p
    img = _rand(B, 3, S, S, seed=1)
    Kp = (3 * p * p + 7) // 8 * 8
    out = torch.empty(B * g * g, Kp, device="cuda", dtype=torch.bfloat16)
    ops.im2col_patch(img, out, p, g)
    ref = F.unfold(img[:, :, :g * p, :g * p], p, stride=p).transpose(1, 2).reshape(B * g * g, 3 * p * p)
    assert torch.equal(out[:, :3 * p * p].float(), ref.bfloat16().float()) and float(out[:, 3 * p * p:].abs().max()) == 0
    D, R, ntok = 96, 4, g * g + 5
    x = torch.zeros(B, ntok, D, device="cuda")
    cls, reg = _rand(D, seed=2), _rand(R, D, seed=3)
    ops.prefix_tokens(x, cls, reg, B, ntok, D, R)
    assert torch.equal(x[:, 0], cls.expand(B, D)) and torch.equal(x[:, 1:5], reg.expand(B, R, D)) and float(x[:, 5:].abs().max()) == 0
    src = _rand(1003, seed=4)
    dst = torch.empty(1003, device="cuda", dtype=torch.bfloat16)
    ops.cast_bf16(src, dst)
    assert torch.equal(dst, src.bfloat16())
    xx, gam = _rand(50, 96, seed=5), _rand(96, seed=6)
    o = torch.empty(50, 96, device="cuda", dtype=torch.bfloat16)
    ops.scale_cols_cast(xx, gam, o)
    assert torch.equal(o, (xx * gam).bfloat16())
