"""GPU parity of the direct (LDS-staged) 3x3 convolution against plain PyTorch fp32 math: forward with BatchNorm statistics,
input gradient through the adjoint packing, ragged image sizes, channel slices of wider buffers."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _rel(a, b):
    a, b = a.double(), b.double()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def _rand(*shape, seed=0, scale=1.0):
    g = torch.Generator(device="cuda").manual_seed(seed)
    return torch.randn(*shape, generator=g, device="cuda") * scale


# (B, H, W, Cin, Cin_pad, ldx, Cout, rot): the MIPHEI fus3 layer, a ragged image, small-channel layers, a slice of a wider buffer
@pytest.mark.parametrize("B,H,W,cin,cp,ldx,cout,rot", [(2, 256, 256, 67, 72, 72, 32, 3), (3, 37, 45, 67, 72, 72, 32, 0),
                                                       (2, 64, 96, 3, 8, 8, 32, 0), (2, 48, 64, 32, 32, 32, 32, 0),
                                                       (1, 40, 72, 64, 64, 80, 32, 0), (2, 56, 64, 30, 32, 48, 64, 0)])
def test_direct_conv_forward_and_stats(B, H, W, cin, cp, ldx, cout, rot):
    import miphei_vit_amd.ops as ops
    assert ops.conv3x3_direct_supported(cp, cout)
    w = _rand(cout, cin, 3, 3, seed=1, scale=(9 * cin) ** -0.5)
    x = _rand(B, H, W, cin, seed=2).bfloat16()                     # logical input, parameter channel order
    # packed NHWC buffer: packed channel c holds parameter channel (c + rot) % cin, zero pad to cp, garbage beyond (ldx > cp)
    xb = torch.full((B, H, W, ldx), 3.0, device="cuda", dtype=torch.bfloat16)
    xb[..., :cp] = 0
    perm = (torch.arange(cin, device="cuda") + rot) % cin
    xb[..., :cin] = x[..., perm]
    wp = ops.pack_conv3x3_direct(w, cout, cp, rot=rot)
    y = torch.full((B, H, W, cout + 8), 5.0, device="cuda", dtype=torch.bfloat16)
    nslots = 32
    stats = torch.zeros(nslots * 2 * cout, device="cuda", dtype=torch.float64)
    ops.conv3x3_direct(xb, wp, y, B=B, H=H, W=W, cin_pad=cp, ldx=ldx, cout=cout, ldy=cout + 8, stats=stats, nslots=nslots)
    ref = F.conv2d(x.float().permute(0, 3, 1, 2), w.bfloat16().float(), padding=1).permute(0, 2, 3, 1)
    assert _rel(y[..., :cout].float(), ref) < 4e-3
    assert bool((y[..., cout:] == 5.0).all())                       # only the Cout channels of each pixel are written
    st = stats.view(nslots, 2, cout).sum(0)
    assert _rel(st[0], ref.double().sum((0, 1, 2))) < 1e-4 + 1e-3 and _rel(st[1], (ref.double() ** 2).sum((0, 1, 2))) < 1e-4
    # same result as the implicit-GEMM convolution it replaces (when the channel stride allows that path)
    if ldx == cp:
        wk = torch.empty(cout, 9 * cp, device="cuda", dtype=torch.bfloat16)
        ops.pack_conv3x3_weights(w, wk, None, rot=rot)
        y2 = torch.empty(B * H * W, cout, device="cuda", dtype=torch.bfloat16)
        ops.gemm(xb, wk, y2, M=B * H * W, amode=ops.A_CONV3, conv=(H, W, cp, cp, H, W, 1))
        assert _rel(y[..., :cout].float().reshape(-1, cout), y2.float()) < 3e-3
    y3 = torch.empty_like(y)
    ops.conv3x3_direct(xb, wp, y3, B=B, H=H, W=W, cin_pad=cp, ldx=ldx, cout=cout, ldy=cout + 8)     # no statistics wanted
    assert torch.equal(y3[..., :cout], y[..., :cout])


@pytest.mark.parametrize("B,H,W", [(2, 256, 256), (3, 37, 45)])
def test_direct_conv_input_gradient_of_the_last_fusion_block(B, H, W):
    """dX (the 64 up-sampled channels of the 67) = adjoint convolution on the flipped / transposed weights (mode 1 pack)"""
    import miphei_vit_amd.ops as ops
    cin, cout, rot, nwant = 67, 32, 3, 64
    w = _rand(cout, cin, 3, 3, seed=1, scale=(9 * cin) ** -0.5)
    dy = _rand(B, H, W, cout, seed=3).bfloat16()
    wp = ops.pack_conv3x3_direct(w, nwant, cout, rot=rot, dgrad=True)
    dx = torch.empty(B, H, W, nwant, device="cuda", dtype=torch.bfloat16)
    ops.conv3x3_direct(dy, wp, dx, B=B, H=H, W=W, cin_pad=cout, ldx=cout, cout=nwant, ldy=nwant)
    xx = torch.zeros(B, cin, H, W, device="cuda", requires_grad=True)
    F.conv2d(xx, w.bfloat16().float(), padding=1).backward(dy.float().permute(0, 3, 1, 2))
    ref = xx.grad.permute(0, 2, 3, 1)[..., (torch.arange(nwant, device="cuda") + rot) % cin]     # packed order [up(64) | img(3)]
    assert _rel(dx.float(), ref) < 4e-3


@pytest.mark.parametrize("B,H,W,cin,cp,rot", [(2, 256, 256, 67, 72, 3), (3, 37, 45, 67, 72, 0), (2, 64, 96, 3, 8, 0), (2, 48, 64, 32, 32, 0)])
def test_direct_conv_weight_gradient(B, H, W, cin, cp, rot):
    """dW of the stride-1 3x3 conv from LDS-staged X / dY tiles (contraction over pixels), through the n-major unpack"""
    import miphei_vit_amd.ops as ops
    cout = 32
    x = _rand(B, H, W, cin, seed=2).bfloat16()
    dy = _rand(B, H, W, cout, seed=3).bfloat16()
    xb = torch.zeros(B, H, W, cp, device="cuda", dtype=torch.bfloat16)
    perm = (torch.arange(cin, device="cuda") + rot) % cin
    xb[..., :cin] = x[..., perm]
    dwn = torch.zeros(cout, 9 * cp, device="cuda")
    ops.conv3x3_direct_wgrad(xb, dy, dwn, B=B, H=H, W=W, cin_pad=cp, ldx=cp, cout=cout, ldy=cout)
    dW = torch.empty(cout, cin, 3, 3, device="cuda")
    ops.unpack_conv3x3_wgrad(dwn, dW, cp, rot=rot, n_major=True)
    w = torch.zeros(cout, cin, 3, 3, device="cuda", requires_grad=True)
    F.conv2d(x.float().permute(0, 3, 1, 2), w, padding=1).backward(dy.float().permute(0, 3, 1, 2))
    assert _rel(dW, w.grad) < 2e-3
    # against the TN-GEMM path it replaces
    dWt = torch.zeros(9 * cp, cout, device="cuda")
    ops.gemm_tn(xb, dy, dWt, M=B * H * W, I=9 * cp, J=cout, ldb=cout, ldci=cout, msplit=8, conv=(H, W, cp, cp, H, W, 1))
    dW2 = torch.empty_like(dW)
    ops.unpack_conv3x3_wgrad(dWt, dW2, cp, rot=rot)
    assert _rel(dW, dW2) < 2e-3


def test_direct_conv_rejects_unsupported_shapes():
    import miphei_vit_amd.ops as ops
    assert not ops.conv3x3_direct_supported(176, 64) and not ops.conv3x3_direct_supported(72, 48)
    x = torch.zeros(1, 8, 8, 176, device="cuda", dtype=torch.bfloat16)
    with pytest.raises(RuntimeError):
        ops.conv3x3_direct(x, x, x, B=1, H=8, W=8, cin_pad=176, ldx=176, cout=64, ldy=176)


# (B, H, W, Cin, ldx, Cout): the three wide MIPHEI fusion layers at batch 2 (1728 -> 256 @ 32^2, 352 -> 128 @ 64^2, 176 -> 64 @
# 128^2), ragged image sizes, a channel count that ends inside a 32-channel chunk and inside a 64-channel slice, a wider buffer
@pytest.mark.parametrize("B,H,W,cin,ldx,cout", [(2, 32, 32, 1728, 1728, 256), (2, 64, 64, 352, 352, 128), (2, 128, 128, 176, 176, 64),
                                                (3, 37, 45, 40, 48, 72), (1, 9, 70, 24, 24, 8), (2, 16, 32, 96, 96, 200)])
def test_chunked_conv_forward_stats_and_input_gradient(B, H, W, cin, ldx, cout):
    """csrc/conv_chunked.hip against F.conv2d on the same bf16-rounded operands: forward with BatchNorm statistics (and without),
    against the implicit-GEMM convolution it replaces, and the input gradient through the adjoint packing against autograd."""
    import miphei_vit_amd.ops as ops
    w = _rand(cout, cin, 3, 3, seed=1, scale=(9 * cin) ** -0.5)
    x = _rand(B, H, W, cin, seed=2).bfloat16()
    xb = torch.full((B, H, W, ldx), 3.0, device="cuda", dtype=torch.bfloat16)      # garbage beyond the used channels (ldx > cin)
    xb[..., :cin] = x
    wp = ops.pack_conv3x3_chunked(w)
    ldy = cout + 8
    y = torch.full((B, H, W, ldy), 5.0, device="cuda", dtype=torch.bfloat16)
    nslots = 256
    stats = torch.zeros(nslots * 2 * cout, device="cuda", dtype=torch.float64)
    ops.conv3x3_chunked(xb, wp, y, B=B, H=H, W=W, cin=cin, ldx=ldx, cout=cout, ldy=ldy, stats=stats, nslots=nslots)
    ref = F.conv2d(x.float().permute(0, 3, 1, 2), w.bfloat16().float(), padding=1).permute(0, 2, 3, 1)
    assert _rel(y[..., :cout].float(), ref) < 4e-3
    assert float((y[..., :cout].float() - ref).abs().max()) < 0.05 * float(ref.abs().max()) + 0.02      # no misplaced tile / chunk
    assert bool((y[..., cout:] == 5.0).all())
    st = stats.view(nslots, 2, cout).sum(0)
    assert _rel(st[0], ref.double().sum((0, 1, 2))) < 1e-4 + 2e-3 and _rel(st[1], (ref.double() ** 2).sum((0, 1, 2))) < 1e-4
    y3 = torch.empty_like(y)
    ops.conv3x3_chunked(xb, wp, y3, B=B, H=H, W=W, cin=cin, ldx=ldx, cout=cout, ldy=ldy)
    assert torch.equal(y3[..., :cout], y[..., :cout])
    if ldx == cin:      # the implicit-GEMM convolution of the same layer
        wk = torch.empty(cout, 9 * cin, device="cuda", dtype=torch.bfloat16)
        ops.pack_conv3x3_weights(w, wk, None, rot=0)
        y2 = torch.empty(B * H * W, cout, device="cuda", dtype=torch.bfloat16)
        ops.gemm(xb, wk, y2, M=B * H * W, amode=ops.A_CONV3, conv=(H, W, cin, cin, H, W, 1))
        assert _rel(y[..., :cout].float().reshape(-1, cout), y2.float()) < 3e-3
    # input gradient: dX[b, y, x, ci] from dY through the flipped / transposed weights
    dy = _rand(B, H, W, cout, seed=3).bfloat16()
    wpb = ops.pack_conv3x3_chunked(w, dgrad=True)
    dx = torch.full((B, H, W, cin + 8), 7.0, device="cuda", dtype=torch.bfloat16)
    ops.conv3x3_chunked(dy, wpb, dx, B=B, H=H, W=W, cin=cout, ldx=cout, cout=cin, ldy=cin + 8)
    xx = torch.zeros(B, cin, H, W, device="cuda", requires_grad=True)
    F.conv2d(xx, w.bfloat16().float(), padding=1).backward(dy.float().permute(0, 3, 1, 2))
    assert _rel(dx[..., :cin].float(), xx.grad.permute(0, 2, 3, 1)) < 4e-3
    assert bool((dx[..., cin:] == 7.0).all())


@pytest.mark.parametrize("B,H,W,cin,ldx,cout", [(2, 32, 32, 1728, 1728, 256), (2, 64, 64, 352, 352, 128), (2, 128, 128, 176, 176, 64),
                                                (3, 37, 45, 40, 48, 72), (1, 9, 70, 24, 24, 8), (16, 32, 32, 96, 96, 64)])
def test_chunked_conv_weight_gradient(B, H, W, cin, ldx, cout):
    """dW on the chunked LDS-staged tiles (contraction over pixels, transposing LDS reads) against autograd of F.conv2d on the
    same bf16-rounded operands and against the TN-GEMM path it replaces; accumulation into a non-zero buffer; one tile group per
    (slice, chunk) pair (first case: 4 x 54 pairs) and several groups meeting in atomics (the others)."""
    import miphei_vit_amd.ops as ops
    x = _rand(B, H, W, cin, seed=2).bfloat16()
    xb = torch.full((B, H, W, ldx), 3.0, device="cuda", dtype=torch.bfloat16)
    xb[..., :cin] = x
    dy = _rand(B, H, W, cout, seed=3).bfloat16()
    dwn = torch.full((cout, 9 * cin), 0.5, device="cuda")
    ops.conv3x3_chunked_wgrad(xb, dy, dwn, B=B, H=H, W=W, cin=cin, cin_pad=cin, ldx=ldx, cout=cout, ldy=cout)
    wz = torch.zeros(cout, cin, 3, 3, device="cuda", requires_grad=True)
    F.conv2d(x.float().permute(0, 3, 1, 2), wz, padding=1).backward(dy.float().permute(0, 3, 1, 2))
    ref = wz.grad.permute(0, 2, 3, 1).reshape(cout, 9 * cin)           # [n][(ky, kx, c)]
    assert _rel(dwn - 0.5, ref) < 2e-3
    if ldx == cin:
        dwt = torch.zeros(9 * cin, cout, device="cuda")
        ops.gemm_tn(xb, dy.view(-1, cout), dwt, M=B * H * W, I=9 * cin, J=cout, ldb=cout, ldci=cout, msplit=4, conv=(H, W, cin, cin, H, W, 1))
        assert _rel(dwn - 0.5, dwt.t()) < 2e-3
