"""GPU parity of the fused attention kernels against plain PyTorch fp32 math (incl. autograd)."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu


def _rel(a, b):
    a, b = a.double(), b.double()
    return float(((a - b) ** 2).sum().sqrt() / b.pow(2).sum().sqrt().clamp_min(1e-30))


@pytest.mark.parametrize("B,N,H,Dh", [(2, 329, 3, 64), (1, 69, 4, 16), (2, 86, 3, 32), (1, 1301, 2, 64), (3, 128, 2, 64)])
def test_attention_fwd_bwd(B, N, H, Dh):
    import miphei_vit_amd.ops as ops
    g = torch.Generator(device="cuda").manual_seed(B * 1000 + N)
    qkv = (torch.randn(B, N, 3, H, Dh, generator=g, device="cuda") * 1.5).bfloat16()
    # spike one key against one query so the running max jumps mid-sequence (online-softmax rescale path)
    qkv[0, N // 3, 0, 0] *= 4
    qkv[0, N - 2, 1, 0] = qkv[0, N // 3, 0, 0]
    scale = Dh ** -0.5
    out = torch.empty(B, N, H * Dh, device="cuda", dtype=torch.bfloat16)
    lse = torch.empty(B, H, N, device="cuda")
    ops.attention_fwd(qkv, out, lse, B, N, H, Dh, scale)
    x = qkv.float().requires_grad_(True)
    q, k, v = x.permute(2, 0, 3, 1, 4).unbind(0)
    s = (q @ k.transpose(-1, -2)) * scale
    ref = (s.softmax(-1) @ v).transpose(1, 2).reshape(B, N, H * Dh)
    assert _rel(out.float(), ref) < 6e-3
    assert _rel(lse, torch.logsumexp(s, -1)) < 1e-4
    dO = torch.randn(B, N, H * Dh, generator=g, device="cuda").bfloat16()
    ref.backward(dO.float())
    dqkv = torch.zeros_like(qkv)
    dsum = torch.empty(B, H, N, device="cuda")
    ops.attention_bwd(qkv, out, dO, lse, dsum, dqkv, B, N, H, Dh, scale)
    for i, name in enumerate("qkv"):
        assert _rel(dqkv[:, :, i].float(), x.grad[:, :, i]) < 1.5e-2, name
