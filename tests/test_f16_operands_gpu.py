"""GPU: the fp16-operand library (libmiphei_hip_f16.so: the same sources built with -DMVIT_F16, csrc/common.hpp) kernel by kernel
against plain PyTorch fp32 math -- the arithmetic type of the reference's evaluation convention `generator.eval().cuda().half()`
(/root/reference/evaluation/eval_orion.py:191, 214-215).  fp16 carries 11 mantissa bits against bf16's 8: the tolerances here are
4x tighter than the bf16 tests' for the same kernels, which a library that silently ran bf16 arithmetic on fp16 bit patterns (or the
other way round) would miss by orders of magnitude."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _rel(a, b):
    a, b = a.double(), b.double()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def _rnd(g, *s, dt=torch.float16, scale=1.0):
    return (torch.randn(*s, generator=g, device="cuda") * scale).to(dt)


@pytest.mark.parametrize("M,N,K", [(5264, 1536, 1536), (1024, 128, 128), (300, 96, 200), (5264, 8192, 192)])
def test_dense_gemm_store_accumulate_and_f32_output(M, N, K):
    import miphei_vit_amd.ops as ops
    from miphei_vit_amd import _lib
    g = torch.Generator(device="cuda").manual_seed(M + N + K)
    a, b, bias = _rnd(g, M, K), _rnd(g, N, K, scale=K ** -0.5), _rnd(g, N, dt=torch.float32)
    ref = a.float() @ b.float().t() + bias
    with _lib.operands("f16"):
        c = torch.full((M, N), 7.0, device="cuda", dtype=torch.float16)
        ops.gemm(a, b, c, bias=bias)
        c32 = torch.empty(M, N, device="cuda", dtype=torch.float32)
        ops.gemm(a, b, c32, flags=ops.OUT_F32)
        with pytest.raises(TypeError):                     # a bf16 tensor inside the fp16 mode is a caller error, not reinterpreted bits
            ops.gemm(a.bfloat16(), b, c)
    assert _rel(c.float(), ref) < 1e-3                     # (bf16 operands: 4e-3, tests/test_gemm_ws_gpu.py)
    assert _rel(c32, ref - bias) < 1e-5 * K ** 0.5 + 2e-6
    with pytest.raises(TypeError):                         # and outside it the bf16 library refuses fp16 tensors
        ops.gemm(a, b, c)


@pytest.mark.parametrize("M,D,H", [(5264, 1536, 4096), (1030, 64, 128)])
def test_swiglu_epilogue_and_layerscale_residual(M, D, H):
    """fc1 + SwiGLU (packed a | b rows in groups of 32) and fc2 + LayerScale + residual of the timm block in fp16 operands."""
    import miphei_vit_amd.ops as ops
    from miphei_vit_amd import _lib
    g = torch.Generator(device="cuda").manual_seed(M + H)
    x, w1, b1 = _rnd(g, M, D), _rnd(g, 2 * H, D, scale=D ** -0.5), _rnd(g, 2 * H, dt=torch.float32, scale=0.1)
    w2, gamma, resid = _rnd(g, D, H, scale=H ** -0.5), _rnd(g, D, dt=torch.float32), _rnd(g, M, D, dt=torch.float32)
    a_, b_ = (x.float() @ w1.float().t() + b1).view(M, H // 32, 2, 32).unbind(2)        # groups of 32: a | b
    href = (torch.nn.functional.silu(a_) * b_).reshape(M, H)
    with _lib.operands("f16"):
        h = torch.empty(M, H, device="cuda", dtype=torch.float16)
        ops.gemm(x, w1, h, bias=b1, epi=ops.EPI_SWIGLU)
        out = resid.clone()
        ops.gemm(h, w2, out, gamma=gamma, epi=ops.EPI_RESID, flags=ops.OUT_F32)
    assert _rel(h.float(), href) < 1.5e-3
    assert _rel(out, resid + gamma * (h.float() @ w2.float().t())) < 5e-4


def test_layernorm_operand_pack():
    import miphei_vit_amd.ops as ops
    from miphei_vit_amd import _lib
    M, D = 5264, 1536
    g = torch.Generator(device="cuda").manual_seed(3)
    x, w, b = _rnd(g, M, D, dt=torch.float32, scale=2.0), _rnd(g, D, dt=torch.float32), _rnd(g, D, dt=torch.float32)
    ref = torch.nn.functional.layer_norm(x, (D,), w, b, 1e-6)
    with _lib.operands("f16"):
        y = torch.empty(M, D, device="cuda", dtype=torch.float16)
        ops.layernorm_fwd(x, w, b, y, 1e-6)
    assert _rel(y.float(), ref) < 4e-4                     # one fp16 rounding of the output (bf16: 3e-3)


@pytest.mark.parametrize("B,N,H,Dh", [(2, 329, 3, 64), (1, 1301, 2, 64), (2, 86, 3, 32)])
def test_attention_forward_and_backward(B, N, H, Dh):
    """Forward, its rounding residual and both backward forms (one pass per pair at N = 329, two kernels otherwise) on fp16 operands."""
    import miphei_vit_amd.ops as ops
    from miphei_vit_amd import _lib
    g = torch.Generator(device="cuda").manual_seed(B * 1000 + N)
    qkv = (torch.randn(B, N, 3, H, Dh, generator=g, device="cuda") * 1.5).half()
    scale = Dh ** -0.5
    x = qkv.float().requires_grad_(True)
    q, k, v = x.permute(2, 0, 3, 1, 4).unbind(0)
    s = (q @ k.transpose(-1, -2)) * scale
    ref = (s.softmax(-1) @ v).transpose(1, 2).reshape(B, N, H * Dh)
    dO = torch.randn(B, N, H * Dh, generator=g, device="cuda").half()
    ref.backward(dO.float())
    with _lib.operands("f16"):
        out, res = (torch.empty(B, N, H * Dh, device="cuda", dtype=torch.float16) for _ in range(2))
        lse = torch.empty(B, H, N, device="cuda")
        ops.attention_fwd(qkv, out, lse, B, N, H, Dh, scale, out_res=res)
        dqkv, dsum = torch.zeros_like(qkv), torch.empty(B, H, N, device="cuda")
        ops.attention_bwd(qkv, out, dO, lse, dsum, dqkv, B, N, H, Dh, scale, out_res=res)
    assert _rel(out.float(), ref) < 1.5e-3                 # (bf16 operands: 6e-3)
    assert _rel(lse, torch.logsumexp(s, -1)) < 1e-4
    for i, name in enumerate("qkv"):
        assert _rel(dqkv[:, :, i].float(), x.grad[:, :, i]) < 2.5e-3, name      # (bf16 operands: 1e-2)
