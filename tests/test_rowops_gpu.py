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


# D = 512 / 1024 / 1536 / 2048 run the instantiations without per-group bounds tests (rows of whole float4 groups), the rest the generic ones
@pytest.mark.parametrize("M,D", [(7, 64), (329 * 2, 96), (1000, 1536), (5, 512), (33, 1024), (9, 2048), (17, 1280), (3, 1540)])
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


@pytest.mark.parametrize("K,R,M", [(1536, 8, 5264), (512, 5, 333), (1024, 8, 1000), (2048, 8, 70)])
def test_skinny_rows(K, R, M):
    """the encoder's dt = dq B_q^T || dv B_v^T shapes (column windows of the packed d(qkv)), single and paired launch"""
    ops = _ops()
    Xw = _rand(M, 3 * K, seed=1).bfloat16()
    Wq = _rand(R, K, seed=2, scale=0.1).bfloat16()
    Wv = _rand(R, K, seed=3, scale=0.1).bfloat16()
    out = torch.full((M, 16), 9.0, device="cuda", dtype=torch.bfloat16)
    ops.skinny_xw2(Xw, Wq, out, Xw.view(-1)[2 * K:], Wv, out.view(-1)[8:], ldx=3 * K, ldw=K, ldo=16, M=M, K=K, R=R)
    assert _rel(out[:, :R].float(), Xw[:, :K].float() @ Wq.float().t()) < 4e-3
    assert _rel(out[:, 8:8 + R].float(), Xw[:, 2 * K:].float() @ Wv.float().t()) < 4e-3
    if R < 8:   # columns past R are not written
        assert float((out[:, R:8].float() - 9.0).abs().max()) == 0
    one = torch.empty(M, R, device="cuda", dtype=torch.bfloat16)
    ops.skinny_xw(Xw.view(-1)[K:], Wq, one, ldx=3 * K, M=M)
    assert _rel(one.float(), Xw[:, K:2 * K].float() @ Wq.float().t()) < 4e-3


def test_patch_prefix_cast():
    ops = _ops()
    B, S, p = 2, 128, 14
    g = S // p
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


@pytest.mark.parametrize("M,D,r", [(7, 64, 4), (329 * 2, 96, 8), (5264, 1536, 8), (1301, 1536, 4), (37, 512, 8), (21, 1024, 8), (19, 2048, 4), (23, 1280, 8),
                                   (21056, 1536, 8), (8300, 512, 8), (530, 1536, 8)])   # more than 32 rows per block of the balanced grid; 33 blocks
def test_layernorm_lora_fused_matches_ln_then_matmul(M, D, r):
    """LN1 + the LoRA down-projection in one pass: h identical to the plain LN kernel, t = bf16(h) @ bf16([A_q|A_v])."""
    ops = _ops()
    x = _rand(M, D, seed=1) * 3 + 0.5
    w, b = _rand(D, seed=2) * 0.2 + 1, _rand(D, seed=3) * 0.1
    AcatT = (_rand(2 * r, D, seed=7) * D ** -0.5).bfloat16()
    h0 = torch.empty(M, D, device="cuda", dtype=torch.bfloat16)
    ops.layernorm_fwd(x, w, b, h0, 1e-6)
    h = torch.empty_like(h0)
    t = torch.full((M, 2 * r), float("nan"), device="cuda", dtype=torch.bfloat16)
    ops.layernorm_lora_fwd(x, w, b, h, AcatT, t, 1e-6)
    assert torch.equal(h, h0)
    ref = h0.float() @ AcatT.float().t()
    assert bool(torch.isfinite(t.float()).all())
    assert _rel(t.float(), ref) < 4e-3                       # bf16 output rounding
    t2 = torch.empty_like(t)
    ops.skinny_xw(h0, AcatT, t2)                             # the MFMA kernel it replaces
    assert _rel(t.float(), t2.float()) < 4e-3


def test_lora_pack_and_conv_wgrad_unpack():
    ops = _ops()
    L, D, r, alpha = 3, 96, 8, 0.5
    flat = _rand(L * 4 * r * D, seed=11)
    reg = flat.view(L, 4, r * D)
    Aq, Bq, Av, Bv = reg[:, 0].view(L, D, r), reg[:, 1].view(L, r, D), reg[:, 2].view(L, D, r), reg[:, 3].view(L, r, D)
    bf = torch.bfloat16
    AcatT = torch.empty(L, 2 * r, D, device="cuda", dtype=bf)
    Acat = torch.empty(L, D, 2 * r, device="cuda", dtype=bf)
    B2 = torch.full((L, 3 * D, 2 * r), 7.0, device="cuda", dtype=bf)
    Bqv = torch.empty(L, 2, r, D, device="cuda", dtype=bf)
    ops.lora_pack(flat, AcatT, Acat, B2, Bqv, L, D, r, alpha)
    cat = torch.cat([Aq, Av], dim=2)
    assert torch.equal(Acat, cat.to(bf)) and torch.equal(AcatT, cat.transpose(1, 2).to(bf))
    ref = torch.zeros(L, 3 * D, 2 * r, device="cuda")
    ref[:, :D, :r] = (alpha * Bq).transpose(1, 2)
    ref[:, 2 * D:, r:] = (alpha * Bv).transpose(1, 2)
    assert torch.equal(B2, ref.to(bf))
    assert torch.equal(Bqv[:, 0], (alpha * Bq).to(bf)) and torch.equal(Bqv[:, 1], (alpha * Bv).to(bf))
    for cout, cin, cp, rot in [(32, 67, 72, 3), (48, 3, 8, 0), (256, 1728, 1728, 0)]:
        dWt = _rand(9 * cp, cout, seed=cout)
        dW = torch.empty(cout, cin, 3, 3, device="cuda")
        ops.unpack_conv3x3_wgrad(dWt, dW, cp, rot=rot)
        g = dWt.view(3, 3, cp, cout)[:, :, :cin].permute(3, 2, 0, 1)
        want = torch.empty_like(dW)
        perm = (torch.arange(cin, device="cuda") + rot) % cin
        want[:, perm] = g
        assert torch.equal(dW, want)
        ops.unpack_conv3x3_wgrad(dWt, dW, cp, rot=rot, accumulate=True)
        assert torch.allclose(dW, 2 * want)
        dwn = dWt.view(9, cp, cout).permute(2, 0, 1).contiguous()       # the direct kernel's output-channel-major layout
        ops.unpack_conv3x3_wgrad(dwn, dW, cp, rot=rot, n_major=True)
        assert torch.equal(dW, want)


def test_conv_weight_pack_unpack_multi():
    """the *_multi entry points (all decoder conv weights per launch) give exactly what the single-weight calls give; more than
    8 descriptors are split over launches"""
    ops = _ops()
    shapes = [(48, 3, 8, 0), (96, 48, 48, 0), (192, 96, 96, 0), (256, 1728, 1728, 0), (128, 352, 352, 0), (64, 176, 176, 0),
              (32, 67, 72, 3), (16, 8, 8, 0), (24, 16, 16, 0)]
    bf = torch.bfloat16
    Ws = [_rand(co, ci, 3, 3, seed=co + ci, scale=0.05) for co, ci, _, _ in shapes]
    single, multi = [], []
    for W, (co, ci, cp, rot) in zip(Ws, shapes):
        wk, wd = torch.empty(co, 9 * cp, device="cuda", dtype=bf), torch.empty(cp, 9 * co, device="cuda", dtype=bf)
        ops.pack_conv3x3_weights(W, wk, wd, rot=rot)
        single.append((wk, wd))
        multi.append((W, torch.zeros_like(wk), torch.zeros_like(wd) if co != 16 else None, rot))     # one entry without wd
    ops.pack_conv3x3_weights_multi(multi)
    for (wk, wd), (_, wk2, wd2, _) in zip(single, multi):
        assert torch.equal(wk, wk2) and (wd2 is None or torch.equal(wd, wd2))
    dWts = [_rand(9 * cp, co, seed=7 + co) for co, _, cp, _ in shapes]
    one = [torch.empty(co, ci, 3, 3, device="cuda") for co, ci, _, _ in shapes]
    many = [torch.zeros(co, ci, 3, 3, device="cuda") for co, ci, _, _ in shapes]
    for dWt, dW, (_, _, cp, rot) in zip(dWts, one, shapes):
        ops.unpack_conv3x3_wgrad(dWt, dW, cp, rot=rot)
    ops.unpack_conv3x3_wgrad_multi([(dWt, dW, cp, rot, False) for dWt, dW, (_, _, cp, rot) in zip(dWts, many, shapes)])
    for a, b in zip(one, many):
        assert torch.equal(a, b)
