"""GPU: the wave-specialised 256x128 GEMM (csrc/gemm_ws.hip: producer waves own the operand DMA, consumer waves run MFMA + epilogue)
against plain PyTorch fp32 math -- every epilogue it takes over (store, SwiGLU, LayerScale + residual, d(SwiGLU)), ragged M (last tile
row partly empty), several tiles per persistent block, one and two K tiles, the LoRA second K range, C += and the DropPath row
scale.  Replaces nn.Linear of the timm block (/root/reference/src/generators/foundation_models.py:53-57) and QkvWithLoRA
(/root/reference/src/generators/lora.py:29-33) on the benchmark's shapes."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _rel(a, b):
    a, b = a.double(), b.double()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def _rnd(g, *s, dt=torch.bfloat16, scale=1.0):
    return (torch.randn(*s, generator=g, device="cuda") * scale).to(dt)


def _is_ws(M, N, K, epi=0, flags=0, K2=0):
    import ctypes as C
    import miphei_vit_amd._lib as L
    g = L.GemmArgs()
    g.M, g.N, g.K, g.K2, g.epi, g.flags, g.ksplit, g.amode = M, N, K, K2, epi, flags, 1, 0
    return bool(L.lib().mvit_gemm_variant(C.byref(g)) & (1 << 30))       # bit 30: the wave-specialised kernel takes the problem


@pytest.mark.parametrize("M,N,K", [(5264, 1536, 1536), (5264, 4608, 64), (1024, 128, 128), (2303, 384, 4608), (5264, 4608, 1536),
                                   (70000, 128, 192), (5264, 8192, 192)])      # the last one: 1344 tiles = 5.25 rounds -> band mode (1280 tiles + 192 band items)
def test_ws_store(M, N, K):
    import miphei_vit_amd.ops as ops
    assert _is_ws(M, N, K)
    g = torch.Generator(device="cuda").manual_seed(M + N + K)
    a, b, bias = _rnd(g, M, K), _rnd(g, N, K, scale=K ** -0.5), _rnd(g, N, dt=torch.float32)
    ref = a.float() @ b.float().t() + bias
    c = torch.full((M, N), 7.0, device="cuda", dtype=torch.bfloat16)
    ops.gemm(a, b, c, bias=bias)
    assert _rel(c.float(), ref) < 4e-3
    c32 = torch.empty(M, N, device="cuda", dtype=torch.float32)
    ops.gemm(a, b, c32, flags=ops.OUT_F32)
    assert _rel(c32, ref - bias) < 1e-5 * K ** 0.5 + 2e-6
    c0 = _rnd(g, M, N)
    c2 = c0.clone()
    ops.gemm(a, b, c2, flags=ops.ACCUM_BF16)
    assert _rel(c2.float(), c0.float() + ref - bias) < 6e-3
    # strided operands / output (leading dimensions larger than the extents)
    abig, cbig = _rnd(g, M, K + 64), torch.zeros(M, N + 128, device="cuda", dtype=torch.bfloat16)
    ops.gemm(abig, b, cbig, M=M, K=K, lda=K + 64, ldc=N + 128)
    assert _rel(cbig[:, :N].float(), abig[:, :K].float() @ b.float().t()) < 4e-3 and float(cbig[:, N:].abs().max()) == 0.0


@pytest.mark.parametrize("M,N,K,K2", [(5264, 4608, 1536, 16), (2100, 384, 1536, 16), (1300, 256, 192, 64), (5264, 1536, 64, 8)])
def test_ws_second_k_range(M, N, K, K2):
    import miphei_vit_amd.ops as ops
    g = torch.Generator(device="cuda").manual_seed(K2 + M)
    a, b = _rnd(g, M, K), _rnd(g, N, K, scale=K ** -0.5)
    a2, b2 = _rnd(g, M, K2), _rnd(g, N, K2, scale=0.3)
    c = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    ops.gemm(a, b, c, a2=a2, b2=b2, K2=K2)
    assert _rel(c.float(), a.float() @ b.float().t() + a2.float() @ b2.float().t()) < 4e-3


@pytest.mark.parametrize("M,N,K", [(5264, 1536, 1536), (5264, 1536, 4096), (1100, 256, 64), (2600, 3200, 64)])
def test_ws_layerscale_residual(M, N, K):
    import miphei_vit_amd.ops as ops
    g = torch.Generator(device="cuda").manual_seed(3 * M + K)
    a, b = _rnd(g, M, K), _rnd(g, N, K, scale=K ** -0.5)
    bias, gam, res = _rnd(g, N, dt=torch.float32), _rnd(g, N, dt=torch.float32), _rnd(g, M, N, dt=torch.float32)
    rs = (torch.rand(M, generator=g, device="cuda") < 0.8).float() / 0.8
    lin = a.float() @ b.float().t() + bias
    out = torch.empty(M, N, device="cuda", dtype=torch.float32)
    ops.gemm(a, b, out, bias=bias, gamma=gam, aux=res, epi=ops.EPI_RESID, flags=ops.OUT_F32)
    assert _rel(out - res, gam * lin) < 1e-5 * K ** 0.5 + 2e-6
    out2 = res.clone()                                                       # in place: C is also the residual input
    ops.gemm(a, b, out2, bias=bias, gamma=gam, epi=ops.EPI_RESID, flags=ops.OUT_F32)
    assert torch.equal(out2, out)
    ops.gemm(a, b, out, bias=bias, gamma=gam, aux=res, epi=ops.EPI_RESID, flags=ops.OUT_F32, rowscale=rs)
    assert _rel(out - res, rs[:, None] * gam * lin) < 1e-5 * K ** 0.5 + 2e-6


@pytest.mark.parametrize("M,D,H", [(5264, 1536, 4096), (5264, 448, 2048), (1030, 64, 128)])
def test_ws_swiglu_and_its_backward(M, D, H):
    """fc1 with the packed [a32 | b32] column groups -> g = silu(a) * b (+ saved pre-activation), and the dgrad GEMM through fc2 with
    the d(SwiGLU) epilogue (timm SwiGLUPacked)"""
    import miphei_vit_amd.ops as ops
    g = torch.Generator(device="cuda").manual_seed(M + H)
    x, w1, b1 = _rnd(g, M, D), _rnd(g, 2 * H, D, scale=D ** -0.5), _rnd(g, 2 * H, dt=torch.float32, scale=0.1)
    u = torch.empty(M, 2 * H, device="cuda", dtype=torch.bfloat16)
    gate = torch.empty(M, H, device="cuda", dtype=torch.bfloat16)
    ops.gemm(x, w1, gate, bias=b1, aux=u, epi=ops.EPI_SWIGLU)
    pre = (x.float() @ w1.float().t() + b1).view(M, H // 32, 2, 32)
    a_ref, b_ref = pre[:, :, 0].reshape(M, H), pre[:, :, 1].reshape(M, H)
    assert _rel(u.float(), pre.reshape(M, 2 * H)) < 4e-3
    assert _rel(gate.float(), torch.nn.functional.silu(a_ref) * b_ref) < 6e-3
    gate2 = torch.empty_like(gate)
    ops.gemm(x, w1, gate2, bias=b1, epi=ops.EPI_SWIGLU)                       # without the saved pre-activation (inference)
    assert torch.equal(gate2, gate)
    # backward: dG = dY @ W2 (W2t [H, Dout] as the B operand), du = d(silu(a) b) * dG in the packed layout
    Do = 256
    dy, w2t = _rnd(g, M, Do), _rnd(g, H, Do, scale=Do ** -0.5)
    du = torch.empty(M, 2 * H, device="cuda", dtype=torch.bfloat16)
    ops.gemm(dy, w2t, du, aux=u, epi=ops.EPI_DSWIGLU)
    uf = u.float().view(M, H // 32, 2, 32)
    a_, b_ = uf[:, :, 0].reshape(M, H), uf[:, :, 1].reshape(M, H)
    dG = dy.float() @ w2t.float().t()
    sg = torch.sigmoid(a_)
    da, db = dG * b_ * sg * (1 + a_ * (1 - sg)), dG * a_ * sg
    ref = torch.stack([da.view(M, H // 32, 32), db.view(M, H // 32, 32)], dim=2).reshape(M, 2 * H)
    assert _rel(du.float(), ref) < 6e-3


@pytest.mark.parametrize("M,N,K", [(5264, 8192, 128), (2600, 3200, 192), (5121, 8192, 64), (5183, 8192, 64), (5184, 8192, 64), (5375, 8192, 64)])
def test_ws_band_mode_ragged_rows(M, N, K):
    """Shapes whose partly empty last tile row would cost an extra round: the full tile rows run as whole rounds, the ragged band as
    64-row items (one per block).  Band heights 1, 63, 64 and 255 rows around the item boundaries, store and SwiGLU epilogues."""
    import miphei_vit_amd.ops as ops
    g = torch.Generator(device="cuda").manual_seed(M + K)
    a, b, bias = _rnd(g, M, K), _rnd(g, N, K, scale=K ** -0.5), _rnd(g, N, dt=torch.float32)
    ref = a.float() @ b.float().t() + bias
    c = torch.full((M + 3, N), 7.0, device="cuda", dtype=torch.bfloat16)          # guard rows behind the output
    ops.gemm(a, b, c[:M], bias=bias)
    assert _rel(c[:M].float(), ref) < 4e-3 and float((c[M:].float() - 7.0).abs().max()) == 0.0
    if N % 64 == 0:
        H = N // 2
        u = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        gate = torch.full((M + 3, H), 5.0, device="cuda", dtype=torch.bfloat16)
        ops.gemm(a, b, gate[:M], bias=bias, aux=u, epi=ops.EPI_SWIGLU)
        pre = ref.view(M, H // 32, 2, 32)
        a_ref, b_ref = pre[:, :, 0].reshape(M, H), pre[:, :, 1].reshape(M, H)
        assert _rel(u.float(), ref) < 4e-3
        assert _rel(gate[:M].float(), torch.nn.functional.silu(a_ref) * b_ref) < 6e-3 and float((gate[M:].float() - 5.0).abs().max()) == 0.0


@pytest.mark.parametrize("M,N,K", [(5264, 1536, 1536), (5121, 1536, 704), (5375, 1536, 704), (4100, 2048, 768), (5264, 1536, 640), (300 + 1024, 128, 1536)])
def test_ws_single_round_residual_on_the_producer_waves(M, N, K):
    """One-round LayerScale + residual launches (every block owns one tile, >= 11 K tiles: proj and fc2 of the batch-16 step): the
    residual rows are fetched by the producer waves during the K loop, the accumulators parked in LDS and finished there
    (csrc/gemm_ws.hip, flag 0x4000).  Ragged last tile rows (1 and 255 valid rows), guard rows behind the output, leading dimensions
    larger than the extents, no bias / no LayerScale, the in-place form, K just below the threshold (640: the consumer-side epilogue)."""
    import miphei_vit_amd.ops as ops
    assert _is_ws(M, N, K, epi=ops.EPI_RESID, flags=ops.OUT_F32)
    g = torch.Generator(device="cuda").manual_seed(M + 7 * K)
    a, b = _rnd(g, M, K), _rnd(g, N, K, scale=K ** -0.5)
    bias, gam = _rnd(g, N, dt=torch.float32), _rnd(g, N, dt=torch.float32)
    ldr, ldo = N + 64, N + 32
    resb = _rnd(g, M, ldr, dt=torch.float32)
    res = resb[:, :N]
    lin = a.float() @ b.float().t()
    outb = torch.full((M + 5, ldo), 3.0, device="cuda", dtype=torch.float32)
    ops.gemm(a, b, outb, M=M, bias=bias, gamma=gam, aux=resb, ldaux=ldr, ldc=ldo, epi=ops.EPI_RESID, flags=ops.OUT_F32)
    tol = 1e-5 * K ** 0.5 + 2e-6
    assert _rel(outb[:M, :N] - res, gam * (lin + bias)) < tol
    assert float((outb[M:] - 3.0).abs().max()) == 0.0 and float((outb[:, N:] - 3.0).abs().max()) == 0.0   # nothing beyond M rows / N columns
    out2 = torch.empty(M, N, device="cuda", dtype=torch.float32)
    ops.gemm(a, b, out2, aux=res.contiguous(), epi=ops.EPI_RESID, flags=ops.OUT_F32)                       # no bias, no LayerScale
    assert _rel(out2 - res, lin) < tol
    out3 = res.contiguous().clone()                                                                        # in place
    ops.gemm(a, b, out3, bias=bias, gamma=gam, epi=ops.EPI_RESID, flags=ops.OUT_F32)
    assert torch.equal(out3, outb[:M, :N].contiguous())
    for _ in range(20):                                                                                    # run-to-run identical
        out4 = res.contiguous().clone()
        ops.gemm(a, b, out4, bias=bias, gamma=gam, epi=ops.EPI_RESID, flags=ops.OUT_F32)
        assert torch.equal(out4, out3)


@pytest.mark.parametrize("M,H,K", [(5264, 4096, 1536), (5121, 512, 64), (5375, 1024, 128), (40000, 256, 192), (1024, 128, 64),
                                   (5121, 1024, 704), (70000, 256, 640), (5375, 4096, 576)])
def test_ws_dswiglu_operand_on_pseudo_k_tiles(M, H, K):
    """d(SwiGLU) epilogue with the saved pre-activation DMA'd into the operand ring as pseudo K tiles (csrc/gemm_ws.hip, round 5):
    the benchmark's dfc2 shape (2.6 rounds of tiles), ragged last tile rows with 1 and 255 valid rows, many tiles per block with a
    single K tile each (the ring then carries more pseudo tiles than real ones), guard rows behind the output, a padded leading
    dimension of the saved tensor, run-to-run identity.  From 10 K tiles on the operand is prefetched into the producers' registers
    and written into the ring by ds_write (K = 640 / 704: the smallest such loops, one and many tiles per block, ragged rows);
    below that (K = 576 and less) the pseudo tiles are DMA'd."""
    import miphei_vit_amd.ops as ops
    assert _is_ws(M, H, K, epi=ops.EPI_DSWIGLU)
    g = torch.Generator(device="cuda").manual_seed(M + H + K)
    dy, w2t = _rnd(g, M, K), _rnd(g, H, K, scale=K ** -0.5)
    ldu = 2 * H + 64
    ub = _rnd(g, M, ldu)
    u = ub[:, :2 * H]
    dub = torch.full((M + 4, 2 * H), 9.0, device="cuda", dtype=torch.bfloat16)
    ops.gemm(dy, w2t, dub, M=M, aux=ub, ldaux=ldu, epi=ops.EPI_DSWIGLU)
    uf = u.float().reshape(M, H // 32, 2, 32)
    a_, b_ = uf[:, :, 0].reshape(M, H), uf[:, :, 1].reshape(M, H)
    dG = dy.float() @ w2t.float().t()
    sg = torch.sigmoid(a_)
    da, db = dG * b_ * sg * (1 + a_ * (1 - sg)), dG * a_ * sg
    ref = torch.stack([da.view(M, H // 32, 32), db.view(M, H // 32, 32)], dim=2).reshape(M, 2 * H)
    assert _rel(dub[:M].float(), ref) < 6e-3
    assert float((dub[M:].float() - 9.0).abs().max()) == 0.0
    first = dub[:M].clone()
    for _ in range(10):
        dub.fill_(9.0)
        ops.gemm(dy, w2t, dub, M=M, aux=ub, ldaux=ldu, epi=ops.EPI_DSWIGLU)
        assert torch.equal(dub[:M], first)
