"""GPU: decoder data movement - the tap-table resampler and the specialised bilinear x2 kernel against F.interpolate
(Fusion_Block.forward, /root/reference/src/generators/mipheivit.py:89; Encoder.forward regrid, mipheivit.py:147-151,161-162)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _rel(a, b):
    a, b = a.double(), b.double()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def _nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous()


@pytest.mark.parametrize("B,h,w,C,bn,extra", [(2, 8, 8, 64, True, True), (1, 5, 7, 24, False, False), (3, 16, 16, 8, True, False),
                                             (1, 1, 1, 16, True, True), (2, 32, 32, 128, False, True)])
def test_upsample2x_bilinear(B, h, w, C, bn, extra):
    import miphei_vit_amd.ops as ops
    from miphei_vit_amd.resample import taps
    g = torch.Generator(device="cpu").manual_seed(h * 100 + C)
    x = torch.randn(B, C, h, w, generator=g).cuda()
    src = _nhwc(x).bfloat16()
    ld_src, ld_dst = C + 8, C + 24                      # channel slices of wider buffers on both sides
    srcw = torch.full((B, h, w, ld_src), 7.0, device="cuda", dtype=torch.bfloat16)
    srcw[..., :C] = src
    scale = (torch.rand(C, generator=g) + 0.5).cuda() if bn else None
    shift = (torch.randn(C, generator=g) * 0.3).cuda() if bn else None
    e8 = torch.randn(B, 2 * h, 2 * w, 8, generator=g).cuda().bfloat16() if extra else None
    dst = torch.full((B, 2 * h, 2 * w, ld_dst), 3.0, device="cuda", dtype=torch.bfloat16)
    ops.upsample2x_bilinear(srcw, dst.view(-1)[8:], B=B, h=h, w=w, C=C, ld_src=ld_src, ld_dst=ld_dst, src_bstride=h * w * ld_src,
                            dst_bstride=4 * h * w * ld_dst, scale=scale, shift=shift, extra8=e8)
    xs = src.float().permute(0, 3, 1, 2)
    if bn:
        xs = F.relu(xs * scale.view(1, C, 1, 1) + shift.view(1, C, 1, 1))
    ref = _nhwc(F.interpolate(xs, scale_factor=2, mode="bilinear", align_corners=False))
    assert _rel(dst[..., 8:8 + C].float(), ref) < 4e-3                       # bf16 output rounding
    assert float((dst[..., :8].float() - 3.0).abs().max()) == 0              # channels in front of the slice untouched
    if extra:
        assert torch.equal(dst[..., 8 + C:16 + C], e8) and float((dst[..., 16 + C:].float() - 3.0).abs().max()) == 0
    else:
        assert float((dst[..., 8 + C:].float() - 3.0).abs().max()) == 0
    # the tap-table kernel computes the same map
    t2 = taps("bilinear", h, 2 * h, "cuda"), taps("bilinear", w, 2 * w, "cuda")
    if h == w:
        d2 = torch.zeros(B, 2 * h, 2 * w, C, device="cuda", dtype=torch.bfloat16)
        ops.resample2d(srcw, d2, t2[0], t2[1], B=B, h=h, w=w, H=2 * h, W=2 * w, C=C, ld_src=ld_src, ld_dst=C,
                       src_bstride=h * w * ld_src, dst_bstride=4 * h * w * C, scale=scale, shift=shift)
        assert _rel(d2.float(), dst[..., 8:8 + C].float()) < 4e-3


@pytest.mark.parametrize("B,h,w,C", [(2, 8, 8, 64), (1, 5, 7, 24), (1, 1, 1, 16), (1, 1, 4, 8), (2, 16, 16, 128)])
def test_upsample2x_bilinear_bwd(B, h, w, C):
    """adjoint of the x2 bilinear map against autograd of F.interpolate, d_out a channel slice of a wider buffer"""
    import miphei_vit_amd.ops as ops
    g = torch.Generator(device="cpu").manual_seed(h * 10 + C)
    ld = C + 16
    dout = torch.randn(B, 2 * h, 2 * w, ld, generator=g).cuda().bfloat16()
    din = torch.full((B, h, w, C + 8), 5.0, device="cuda", dtype=torch.bfloat16)
    ops.upsample2x_bilinear_bwd(dout.view(-1)[8:], din, B=B, h=h, w=w, C=C, ld_dout=ld, ld_din=C + 8, dout_bstride=4 * h * w * ld,
                                din_bstride=h * w * (C + 8))
    x = torch.zeros(B, C, h, w, device="cuda", requires_grad=True)
    F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=False).backward(dout[..., 8:8 + C].float().permute(0, 3, 1, 2))
    assert _rel(din[..., :C].float(), _nhwc(x.grad)) < 4e-3
    assert float((din[..., C:].float() - 5.0).abs().max()) == 0


def test_upsample2x_rejects_bad_arguments():
    import miphei_vit_amd.ops as ops
    src = torch.zeros(1, 4, 4, 16, device="cuda", dtype=torch.bfloat16)
    dst = torch.zeros(1, 8, 8, 16, device="cuda", dtype=torch.bfloat16)
    with pytest.raises(RuntimeError):      # C not a multiple of 8
        ops.upsample2x_bilinear(src, dst, B=1, h=4, w=4, C=12, ld_src=16, ld_dst=16, src_bstride=256, dst_bstride=1024)
    with pytest.raises(RuntimeError):      # no room for the extra 8 channels
        ops.upsample2x_bilinear(src, dst, B=1, h=4, w=4, C=16, ld_src=16, ld_dst=16, src_bstride=256, dst_bstride=1024, extra8=dst)
    with pytest.raises(RuntimeError):      # scale without shift
        ops.upsample2x_bilinear(src, dst, B=1, h=4, w=4, C=16, ld_src=16, ld_dst=16, src_bstride=256, dst_bstride=1024,
                                scale=torch.ones(16, device="cuda"))
