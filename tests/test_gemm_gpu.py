"""GPU parity of mvit_gemm_bf16 (MFMA GEMM + implicit-GEMM conv) against plain PyTorch fp32 math."""
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
    return (torch.randn(*shape, generator=g, device="cuda") * scale)


@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (329, 192, 96), (5264, 1536, 1536), (700, 64, 72), (1000, 48, 432),
                                   (513, 32, 648), (260, 96, 64), (130, 288, 64),
                                   # 256-row tiles (M >= 1024): ragged M / N / K, K shorter than the DMA ring, one K tile,
                                   # a K tail behind the explicitly ordered steps, the 256x256 tile (>= 1024 tiles)
                                   (1100, 200, 200), (2000, 384, 136), (1500, 256, 1000), (1024, 128, 64), (1300, 640, 72),
                                   (8192, 8192, 136)])
def test_gemm_store(M, N, K):
    ops = _ops()
    a = _rand(M, K, seed=1).bfloat16()
    b = _rand(N, K, seed=2).bfloat16()
    bias = _rand(N, seed=3)
    ref = a.float() @ b.float().t() + bias
    c = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    ops.gemm(a, b, c, bias=bias)
    assert _rel(c.float(), ref) < 4e-3
    cf = torch.empty(M, N, device="cuda", dtype=torch.float32)
    ops.gemm(a, b, cf, bias=bias, flags=ops.OUT_F32)
    assert _rel(cf, ref) < 1e-5
    # transpose detection: asymmetric operands, exact small integers
    ai = torch.randint(-3, 4, (M, K), device="cuda").bfloat16()
    bi = torch.randint(-3, 4, (N, K), device="cuda").bfloat16()
    ops.gemm(ai, bi, cf, flags=ops.OUT_F32)
    assert torch.equal(cf, ai.float() @ bi.float().t())


@pytest.mark.parametrize("B,S,p,D", [(2, 128, 14, 96), (3, 128, 16, 64), (16, 256, 14, 1536), (5, 256, 14, 200)])
def test_patch_embed_window_gather(B, S, p, D):
    """MVIT_A_PATCH: the patch-embedding convolution (timm PatchEmbed.proj, kernel = stride = p; foundation_models.py:53-57) as an
    in-kernel window gather from the bf16 NHWC image, + bias + pos-embed + row remap past the prefix tokens, against F.conv2d on
    the same bf16-rounded operands.  S = 128 / 256 with p = 14: the last 2 / 4 pixel rows and columns belong to no patch."""
    ops = _ops()
    g, prefix = S // p, 5
    P = g * g
    img = _rand(B, 3, S, S, seed=1)
    w = _rand(D, 3, p, p, seed=2, scale=(3 * p * p) ** -0.5)
    bias, pos = _rand(D, seed=3), _rand(P, D, seed=4)
    img8 = torch.empty(B, S, S, 8, device="cuda", dtype=torch.bfloat16)
    ops.image_to_nhwc(img, img8, 8, nzero=5)
    wk = torch.zeros(D, p, p, 8, device="cuda", dtype=torch.bfloat16)
    wk[..., :3] = w.permute(0, 2, 3, 1)
    out = torch.full((B * (P + prefix), D), 7.0, device="cuda")
    ops.gemm(img8, wk.view(D, -1), out, M=B * P, amode=ops.A_PATCH, conv=(S, S, 8, 8, g, g, p), bias=bias, pos=pos,
             epi=ops.EPI_PATCH, patch=(P, P + prefix, prefix), flags=ops.OUT_F32)
    ref = F.conv2d(img.bfloat16().float(), w.bfloat16().float(), bias, stride=p)            # [B, D, g, g]
    ref = ref.flatten(2).transpose(1, 2) + pos
    got = out.view(B, P + prefix, D)
    assert _rel(got[:, prefix:], ref) < 2e-5
    assert bool((got[:, :prefix] == 7.0).all())                                             # prefix rows are not touched


@pytest.mark.parametrize("M,N,K,K2", [(700, 384, 256, 16), (2100, 384, 1536, 16), (1300, 256, 200, 16)])
def test_gemm_kext_and_splitk(M, N, K, K2):
    ops = _ops()
    a, b = _rand(M, K, seed=1).bfloat16(), _rand(N, K, seed=2).bfloat16()
    a2, b2 = _rand(M, K2, seed=3).bfloat16(), _rand(N, K2, seed=4).bfloat16()
    ref = a.float() @ b.float().t() + a2.float() @ b2.float().t()
    cf = torch.empty(M, N, device="cuda")
    ops.gemm(a, b, cf, a2=a2, b2=b2, flags=ops.OUT_F32)
    assert _rel(cf, ref) < 1e-5
    cf.zero_()
    ops.gemm(a, b, cf, a2=a2, b2=b2, flags=ops.OUT_F32 | ops.ATOMIC, ksplit=3)
    assert _rel(cf, ref) < 1e-5


@pytest.mark.parametrize("M,N,K,P,B", [(329 * 2, 256, 64, 81, 3), (329 * 7, 384, 200, 324, 6)])  # second: 256-row tiles, K tail
def test_gemm_gelu_resid_patch(M, N, K, P, B):
    ops = _ops()
    a, b, bias = _rand(M, K, seed=1).bfloat16(), _rand(N, K, seed=2, scale=0.2).bfloat16(), _rand(N, seed=3)
    u = a.float() @ b.float().t() + bias
    c = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    aux = torch.empty_like(c)
    ops.gemm(a, b, c, bias=bias, aux=aux, epi=ops.EPI_GELU)
    assert _rel(c.float(), F.gelu(u)) < 4e-3 and _rel(aux.float(), u) < 4e-3
    # dgelu
    dg = torch.empty_like(c)
    ops.gemm(a, b, dg, aux=aux, epi=ops.EPI_DGELU)
    uu = aux.float().requires_grad_(True)
    (F.gelu(uu) * (a.float() @ b.float().t())).sum().backward()
    assert _rel(dg.float(), uu.grad) < 5e-3
    # residual + layerscale
    x = _rand(M, N, seed=5)
    gam = _rand(N, seed=6)
    ref = x + gam * u
    ops.gemm(a, b, x, bias=bias, gamma=gam, epi=ops.EPI_RESID)
    assert _rel(x, ref) < 1e-5
    # patch-embed epilogue
    prefix = 5
    Mp = B * P
    ap = _rand(Mp, K, seed=7).bfloat16()
    pos = _rand(P, N, seed=8)
    out = torch.zeros(B * (P + prefix), N, device="cuda")
    ops.gemm(ap, b, out, bias=bias, pos=pos, epi=ops.EPI_PATCH, patch=(P, P + prefix, prefix), flags=ops.OUT_F32)
    ref = (ap.float() @ b.float().t() + bias).view(B, P, N) + pos
    got = out.view(B, P + prefix, N)
    assert _rel(got[:, prefix:], ref) < 1e-5 and float(got[:, :prefix].abs().max()) == 0.0


def _pack_swiglu_rows(H):
    """packed row p of fc1 -> original row: groups of 32 gate cols: [32 a | 32 b]."""
    idx = torch.empty(2 * H, dtype=torch.long)
    g = torch.arange(H)
    idx[(g // 32) * 64 + g % 32] = g
    idx[(g // 32) * 64 + 32 + g % 32] = H + g
    return idx


# second case: the 8-wave 256x128 tile of the batch-16 step; third: the 256x256 tile (>= 1024 tiles)
# last two: the backward epilogue at the batch-16 shape, where the product runs as two launches (rows [0, 4096) on 256x256 tiles, the
# remaining 1168 rows on 256x128 tiles)
@pytest.mark.parametrize("M,D,H,Do", [(400, 96, 256, 64), (2100, 200, 384, 136), (8192, 136, 4096, 64), (5264, 136, 4096, 64),
                                      (5264, 200, 4096, 328)])
def test_gemm_swiglu_fwd_bwd(M, D, H, Do):
    ops = _ops()
    x = _rand(M, D, seed=1).bfloat16()
    w = _rand(2 * H, D, seed=2, scale=0.15)
    bias = _rand(2 * H, seed=3, scale=0.1)
    idx = _pack_swiglu_rows(H).cuda()
    wp, bp = w[idx].bfloat16().contiguous(), bias[idx].contiguous()
    g = torch.empty(M, H, device="cuda", dtype=torch.bfloat16)
    u = torch.empty(M, 2 * H, device="cuda", dtype=torch.bfloat16)
    ops.gemm(x, wp, g, bias=bp, aux=u, epi=ops.EPI_SWIGLU)
    uref = x.float() @ w.bfloat16().float().t() + bias
    a, b = uref[:, :H], uref[:, H:]
    assert _rel(g.float(), F.silu(a) * b) < 5e-3
    assert _rel(u.float()[:, idx.argsort()], uref) < 4e-3
    # backward epilogue: dy[M,Dout] @ W2[Dout,H] -> dg, fused d(silu(a)*b)
    dy = _rand(M, Do, seed=4).bfloat16()
    w2t = _rand(H, Do, seed=5, scale=0.2).bfloat16()  # [N=H, K=Do]
    du = torch.empty(M, 2 * H, device="cuda", dtype=torch.bfloat16)
    ops.gemm(dy, w2t, du, aux=u, epi=ops.EPI_DSWIGLU)
    dg = dy.float() @ w2t.float().t()
    up = u.float()
    ua = up[:, idx.argsort()]
    aa, bb = ua[:, :H].clone().requires_grad_(True), ua[:, H:].clone().requires_grad_(True)
    (F.silu(aa) * bb * dg).sum().backward()
    ref = torch.cat([aa.grad, bb.grad], 1)
    assert _rel(du.float()[:, idx.argsort()], ref) < 5e-3


@pytest.mark.parametrize("B,H,W,C,Cout,stride", [(2, 16, 16, 8, 48, 2), (1, 32, 32, 48, 96, 2), (2, 16, 16, 352, 128, 1),
                                                 (1, 24, 24, 72, 32, 1), (3, 8, 8, 1728, 256, 1),
                                                 (1, 64, 64, 48, 128, 1), (2, 72, 72, 40, 96, 2),   # 256-row conv tiles
                                                 # the three stride-2 ConvStream layers at their real sizes (mipheivit.py:59-64), B = 2
                                                 (2, 256, 256, 8, 48, 2), (2, 128, 128, 48, 96, 2), (2, 64, 64, 96, 192, 2)])
def test_conv3x3_fwd_stats_and_dgrad(B, H, W, C, Cout, stride):
    ops = _ops()
    x = _rand(B, H, W, C, seed=1).bfloat16()  # NHWC
    w = _rand(Cout, C, 3, 3, seed=2, scale=0.05)
    OH, OW = (H + 2 - 3) // stride + 1, (W + 2 - 3) // stride + 1
    wk = w.permute(0, 2, 3, 1).reshape(Cout, 9 * C).bfloat16().contiguous()  # [Cout, (ky,kx,c)]
    M = B * OH * OW
    y = torch.empty(M, Cout, device="cuda", dtype=torch.bfloat16)
    nslots = 8
    stats = torch.zeros(nslots, 2, Cout, device="cuda", dtype=torch.float64)
    ops.gemm(x, wk, y, M=M, amode=ops.A_CONV3, conv=(H, W, C, C, OH, OW, stride), epi=ops.EPI_STATS, stats=stats,
             nslots=nslots)
    xr = x.float().permute(0, 3, 1, 2)
    wr = w.bfloat16().float()
    ref = F.conv2d(xr, wr, stride=stride, padding=1).permute(0, 2, 3, 1).reshape(M, Cout)
    assert _rel(y.float(), ref) < 4e-3
    st = stats.sum(0)
    assert _rel(st[0], ref.double().sum(0)) < 1e-4 and _rel(st[1], (ref.double() ** 2).sum(0)) < 1e-4
    # adjoint (dgrad): dX[b,y,x,c] from dY
    dy = _rand(B, OH, OW, Cout, seed=3).bfloat16()
    wd = w.permute(1, 2, 3, 0).reshape(C, 9 * Cout).bfloat16().contiguous()  # [Cin, (ky,kx,co)]
    dx = torch.empty(B * H * W, C, device="cuda", dtype=torch.float32)
    ops.gemm(dy, wd, dx, M=B * H * W, amode=ops.A_CONV3_T, conv=(OH, OW, Cout, Cout, H, W, stride), flags=ops.OUT_F32)
    xg = xr.clone().requires_grad_(True)
    F.conv2d(xg, wr, stride=stride, padding=1).backward(dy.float().permute(0, 3, 1, 2))
    assert _rel(dx, xg.grad.permute(0, 2, 3, 1).reshape(-1, C)) < 1e-4


@pytest.mark.parametrize("B,H,W,C,Cout,ldc", [(2, 64, 64, 8, 48, 176), (1, 32, 32, 48, 96, 352), (2, 16, 16, 96, 192, 192)])
def test_conv3x3_bias_relu_into_a_wider_buffer(B, H, W, C, Cout, ldc):
    """MVIT_RELU: the eval-mode ConvStream convolution with BatchNorm folded in (weights scaled, shift as bias) writes max(conv + bias, 0)
    straight into its slice of a concat buffer (row stride ldc > Cout); the other columns stay untouched.  The flag is refused anywhere else."""
    ops = _ops()
    x = _rand(B, H, W, C, seed=1).bfloat16()
    w = _rand(Cout, C, 3, 3, seed=2, scale=0.05)
    bias = _rand(Cout, seed=3, scale=0.3)
    OH, OW = (H + 2 - 3) // 2 + 1, (W + 2 - 3) // 2 + 1
    M = B * OH * OW
    wk = w.permute(0, 2, 3, 1).reshape(Cout, 9 * C).bfloat16().contiguous()
    y = torch.full((M, ldc), 7.0, device="cuda", dtype=torch.bfloat16)
    ops.gemm(x, wk, y, M=M, N=Cout, ldc=ldc, amode=ops.A_CONV3, conv=(H, W, C, C, OH, OW, 2), bias=bias, flags=ops.RELU)
    ref = F.conv2d(x.float().permute(0, 3, 1, 2), w.bfloat16().float(), bias=bias, stride=2, padding=1).relu().permute(0, 2, 3, 1).reshape(M, Cout)
    assert _rel(y[:, :Cout].float(), ref) < 4e-3
    assert float(y[:, :Cout].float().min()) >= 0.0 and bool((y[:, Cout:] == 7.0).all())
    assert float((ref == 0).float().mean()) > 0.05                      # the clamp is exercised
    a = _rand(1024, 64, seed=4).bfloat16()
    b = _rand(128, 64, seed=5).bfloat16()
    with pytest.raises(RuntimeError):
        ops.gemm(a, b, torch.empty(1024, 128, device="cuda", dtype=torch.bfloat16), flags=ops.RELU)                 # dense operand
    with pytest.raises(RuntimeError):
        ops.gemm(x, wk, torch.empty(M, Cout, device="cuda"), M=M, amode=ops.A_CONV3, conv=(H, W, C, C, OH, OW, 2),
                 flags=ops.RELU | ops.OUT_F32)                                                                          # f32 output


@pytest.mark.parametrize("M,I,J", [(700, 16, 200), (5264, 8, 1536), (1000, 144, 32), (333, 72, 48)])
def test_gemm_tn_dense(M, I, J):
    ops = _ops()
    a = _rand(M, I, seed=1).bfloat16()
    b = _rand(M, J, seed=2).bfloat16()
    ref = a.float().t() @ b.float()
    c = torch.zeros(I, J, device="cuda")
    ops.gemm_tn(a, b, c, M=M, I=I, J=J, msplit=1)
    assert _rel(c, ref) < 1e-5
    c2 = torch.ones(J, I, device="cuda")            # transposed output strides + accumulate onto existing values
    ops.gemm_tn(a, b, c2, M=M, I=I, J=J, ldci=1, ldcj=I, msplit=5)
    assert _rel(c2 - 1, ref.t()) < 1e-5
    # column windows of wider matrices (LoRA slices)
    aw = _rand(M, 16, seed=3).bfloat16()
    bw = _rand(M, 3 * 104, seed=4).bfloat16()
    c3 = torch.zeros(8, 104, device="cuda")
    ops.gemm_tn(aw.view(-1)[8:], bw.view(-1)[2 * 104:], c3, M=M, I=8, J=104, lda=16, ldb=3 * 104, msplit=3)
    assert _rel(c3, aw[:, 8:].float().t() @ bw[:, 208:].float()) < 1e-5
    # two outputs from one pass: row split (both adapters' dA) and row+column split (both adapters' dB)
    full = aw.float().t() @ bw.float()                      # [16, 312]
    d0, d1 = torch.zeros(104, 8, device="cuda"), torch.zeros(104, 8, device="cuda")
    ops.gemm_tn(aw, bw, d0, M=M, I=16, J=104, lda=16, ldb=3 * 104, ldci=1, ldcj=8, msplit=2, c2=d1, isplit=8)
    assert _rel(d0, full[:8, :104].t()) < 1e-5 and _rel(d1, full[8:, :104].t()) < 1e-5
    e0, e1 = torch.zeros(8, 104, device="cuda"), torch.zeros(8, 104, device="cuda")
    ops.gemm_tn(aw, bw, e0, M=M, I=16, J=3 * 104, lda=16, ldb=3 * 104, ldci=104, ldcj=1, msplit=2, c2=e1, isplit=8, j1=104,
                jlo2=2 * 104)
    assert _rel(e0, full[:8, :104]) < 1e-5 and _rel(e1, full[8:, 208:]) < 1e-5


@pytest.mark.parametrize("nb,msplit", [(3, 1), (5, 4)])
def test_gemm_tn_batched(nb, msplit):
    """batch > 1: the LoRA weight gradients of a group of ViT blocks from one launch, both adapters, strided operands/outputs"""
    ops = _ops()
    M, r, D = 650, 8, 104
    t = _rand(nb, M, 2 * r, seed=5).bfloat16()
    dqkv = _rand(nb, M, 3 * D, seed=6).bfloat16()
    h = _rand(nb, M, D, seed=7).bfloat16()
    g = torch.zeros(nb, 4, r * D, device="cuda")          # [block, (Aq, Bq, Av, Bv), r*D] as in the flat gradient buffer
    ops.gemm_tn(t, dqkv, g[0, 1], M=M, I=2 * r, J=3 * D, lda=2 * r, ldb=3 * D, ldci=D, ldcj=1, msplit=msplit, c2=g[0, 3],
                isplit=r, j1=D, jlo2=2 * D, batch=nb, stride_a=M * 2 * r, stride_b=M * 3 * D, stride_c=4 * r * D)
    ops.gemm_tn(t, h, g[0, 0], M=M, I=2 * r, J=D, lda=2 * r, ldb=D, ldci=1, ldcj=r, msplit=msplit, c2=g[0, 2], isplit=r,
                batch=nb, stride_a=M * 2 * r, stride_b=M * D, stride_c=4 * r * D)
    for b in range(nb):
        full = t[b].float().t() @ dqkv[b].float()
        assert _rel(g[b, 1].view(r, D), full[:r, :D]) < 1e-5 and _rel(g[b, 3].view(r, D), full[r:, 2 * D:]) < 1e-5
        fa = t[b].float().t() @ h[b].float()
        assert _rel(g[b, 0].view(D, r), fa[:r].t()) < 1e-5 and _rel(g[b, 2].view(D, r), fa[r:].t()) < 1e-5
    with pytest.raises(Exception):      # batched products need the dense A mode
        ops.gemm_tn(t, h, g[0, 0], M=M, I=72, J=32, conv=(5, 5, 8, 8, 5, 5, 1), batch=2)


@pytest.mark.parametrize("M,r,D,nb,msplit", [(5264, 16, 1536, 2, 10), (700, 16, 200, 1, 3), (650, 4, 104, 3, 2)])
def test_gemm_tn_i_contiguous_outputs(M, r, D, nb, msplit):
    """ldci == 1 (LoRA dA, stored [D][r]): the accumulator tiles are computed transposed so that an atomic instruction covers whole
    lines of the output; rank 16 at the training size (both adapters: lanes 0..15 / 16..31 of an instruction go to C / C2), a rank
    whose 2 r is not 32, a ragged last column block."""
    ops = _ops()
    dt = _rand(nb, M, 2 * r, seed=11).bfloat16()
    h = _rand(nb, M, D, seed=12).bfloat16()
    g = torch.zeros(nb, 4, r * D, device="cuda")
    ops.gemm_tn(dt, h, g[0, 0], M=M, I=2 * r, J=D, lda=2 * r, ldb=D, ldci=1, ldcj=r, msplit=msplit, c2=g[0, 2], isplit=r,
                batch=nb, stride_a=M * 2 * r, stride_b=M * D, stride_c=4 * r * D)
    for b in range(nb):
        fa = dt[b].float().t() @ h[b].float()
        assert _rel(g[b, 0].view(D, r), fa[:r].t()) < 1e-5 and _rel(g[b, 2].view(D, r), fa[r:].t()) < 1e-5
        assert float(g[b, 1].abs().max()) == 0 and float(g[b, 3].abs().max()) == 0        # nothing written outside the two outputs


@pytest.mark.parametrize("B,H,W,C,Cout,stride", [(2, 16, 16, 8, 48, 2), (1, 32, 32, 48, 96, 2), (2, 16, 16, 352, 128, 1),
                                                 (1, 24, 24, 72, 32, 1), (2, 8, 8, 1728, 256, 1),
                                                 (2, 256, 256, 8, 48, 2), (2, 128, 128, 48, 96, 2), (2, 64, 64, 96, 192, 2)])   # ConvStream layers
def test_gemm_tn_conv_wgrad(B, H, W, C, Cout, stride):
    ops = _ops()
    x = _rand(B, H, W, C, seed=1).bfloat16()
    OH, OW = (H + 2 - 3) // stride + 1, (W + 2 - 3) // stride + 1
    M = B * OH * OW
    dy = _rand(M, Cout, seed=2).bfloat16()
    dwt = torch.zeros(9 * C, Cout, device="cuda")
    ops.gemm_tn(x, dy, dwt, M=M, I=9 * C, J=Cout, msplit=3, conv=(H, W, C, C, OH, OW, stride))
    w = torch.zeros(Cout, C, 3, 3, device="cuda", requires_grad=True)
    F.conv2d(x.float().permute(0, 3, 1, 2), w, stride=stride, padding=1).backward(
        dy.float().view(B, OH, OW, Cout).permute(0, 3, 1, 2))
    ref = w.grad.permute(2, 3, 1, 0).reshape(9 * C, Cout)
    assert _rel(dwt, ref) < 1e-5


@pytest.mark.parametrize("M,N,K", [(5264, 4608, 1536), (5264, 4608, 448), (5264, 4608, 384), (5000, 4096, 1600), (21056, 1536, 512)])
def test_gemm_store_persistent_blocks(M, N, K):
    """More tiles than blocks (persistent blocks walk several tiles, the next tile's first K tiles are in flight while the
    epilogue of the current one runs); short K loops, ragged last row tile; repeated launches must agree bit for bit."""
    ops = _ops()
    b = _rand(N, K, seed=2).bfloat16()
    bias = _rand(N, seed=3)
    outs = []
    for rep in range(3):
        a = _rand(M, K, seed=10 + rep).bfloat16()
        c = torch.full((M + 8, N), 7.0, device="cuda", dtype=torch.bfloat16)      # guard rows behind the matrix
        ops.gemm(a, b, c[:M], bias=bias)
        ref = a.float() @ b.float().t() + bias
        assert _rel(c[:M].float(), ref) < 4e-3
        assert float((c[:M].float() - ref).abs().max()) < 0.05 * float(ref.abs().max())   # no stale / misplaced 16-byte piece
        assert bool((c[M:] == 7.0).all())
        c2 = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        ops.gemm(a, b, c2, bias=bias)
        assert torch.equal(c[:M], c2)
        outs.append(c2)
    assert not torch.equal(outs[0], outs[1])


@pytest.mark.parametrize("M,D,H,with_aux", [(5264, 1536, 4096, True), (5264, 448, 2048, True), (5264, 1536, 4096, False)])
def test_gemm_swiglu_persistent_blocks(M, D, H, with_aux):
    ops = _ops()
    idx = _pack_swiglu_rows(H).cuda()
    w = _rand(2 * H, D, seed=2, scale=D ** -0.5)
    bias = _rand(2 * H, seed=3, scale=0.1)
    wp, bp = w[idx].bfloat16().contiguous(), bias[idx].contiguous()
    for rep in range(2):
        x = _rand(M, D, seed=20 + rep).bfloat16()
        g = torch.full((M + 4, H), 7.0, device="cuda", dtype=torch.bfloat16)
        u = torch.full((M + 4, 2 * H), 7.0, device="cuda", dtype=torch.bfloat16)
        ops.gemm(x, wp, g[:M], bias=bp, aux=(u[:M] if with_aux else None), epi=ops.EPI_SWIGLU)
        uref = x.float() @ w.bfloat16().float().t() + bias
        a, b = uref[:, :H], uref[:, H:]
        ref = F.silu(a) * b
        assert _rel(g[:M].float(), ref) < 5e-3
        assert float((g[:M].float() - ref).abs().max()) < 0.05 * float(ref.abs().max()) + 0.05
        assert bool((g[M:] == 7.0).all())
        if with_aux:
            assert _rel(u[:M].float()[:, idx.argsort()], uref) < 4e-3 and bool((u[M:] == 7.0).all())
        else:
            assert bool((u == 7.0).all())
