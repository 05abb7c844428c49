"""GPU parity of the fused attention kernels against plain PyTorch fp32 math (incl. autograd)."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu


def _rel(a, b):
    a, b = a.double(), b.double()
    return float(((a - b) ** 2).sum().sqrt() / b.pow(2).sum().sqrt().clamp_min(1e-30))


@pytest.mark.parametrize("B,N,H,Dh", [(2, 329, 3, 64), (1, 69, 4, 16), (2, 86, 3, 32), (1, 1301, 2, 64), (3, 128, 2, 64),
                                       (2, 329, 24, 64),       # H-Optimus-0's own head count (24-head stride pattern of the packed qkv)
                                       # Dh = 64, N <= 336: the one-pass backward (one workgroup per (batch, head) pair, round 6) at its
                                       # edges -- the largest N it takes and the first it leaves to the two-kernel form, one key wave, a
                                       # single query step, key waves / query blocks that end exactly on and one past a boundary
                                       (1, 336, 2, 64), (1, 337, 2, 64), (2, 48, 2, 64), (2, 49, 3, 64), (1, 7, 2, 64), (2, 32, 1, 64),
                                       (1, 33, 2, 64), (2, 257, 12, 64), (1, 96, 2, 64), (1, 97, 2, 64), (1, 320, 2, 64), (1, 305, 2, 64)])
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
    res = torch.empty_like(out)
    ops.attention_fwd(qkv, out, lse, B, N, H, Dh, scale, out_res=res)
    x = qkv.float().requires_grad_(True)
    q, k, v = x.permute(2, 0, 3, 1, 4).unbind(0)
    s = (q @ k.transpose(-1, -2)) * scale
    ref = (s.softmax(-1) @ v).transpose(1, 2).reshape(B, N, H * Dh)
    assert _rel(out.float(), ref) < 6e-3
    e_o = _rel(out.float(), ref)
    assert _rel(out.float() + res.float(), ref) < max(2.5e-3, 0.8 * e_o)   # out + its rounding residual = the f32 accumulators (P is bf16 in PV)
    assert _rel(lse, torch.logsumexp(s, -1)) < 1e-4
    dO = torch.randn(B, N, H * Dh, generator=g, device="cuda").bfloat16()
    ref.backward(dO.float())
    dqkv = torch.zeros_like(qkv)
    dsum = torch.empty(B, H, N, device="cuda")
    ops.attention_bwd(qkv, out, dO, lse, dsum, dqkv, B, N, H, Dh, scale, out_res=res)
    for i, name in enumerate("qkv"):
        assert _rel(dqkv[:, :, i].float(), x.grad[:, :, i]) < 1e-2, name
    ops.attention_bwd(qkv, out, dO, lse, dsum, dqkv, B, N, H, Dh, scale)       # without the residual: D from the bf16 out
    for i, name in enumerate("qkv"):
        assert _rel(dqkv[:, :, i].float(), x.grad[:, :, i]) < 1.5e-2, name


def test_attention_backward_near_uniform_scores():
    """Near-uniform attention (small logits; the regime of the deep blocks under the synthetic initialisation): dQ is a small
    residual of large terms and D = sum_d dO * O must be consistent with dP.  With the rounding residual of O the kernel stays at
    the bf16 noise floor; from the bf16 O alone dQ is tens of per cent off (the unfused bf16-autocast softmax backward is not)."""
    import miphei_vit_amd.ops as ops
    B, N, H, Dh = 2, 329, 4, 64
    g = torch.Generator(device="cuda").manual_seed(5)
    qkv = torch.randn(B, N, 3, H, Dh, generator=g, device="cuda")
    qkv[:, :, 0] *= 0.05                                               # tiny queries: scores ~ 0, softmax ~ uniform
    qkv[:, :, 1] += 1.5                                                # keys with a common component (mean key != 0)
    qkv[:, :, 2] += 1.0
    qkv = qkv.bfloat16()
    scale = Dh ** -0.5
    out, res = (torch.empty(B, N, H * Dh, device="cuda", dtype=torch.bfloat16) for _ in range(2))
    lse = torch.empty(B, H, N, device="cuda")
    ops.attention_fwd(qkv, out, lse, B, N, H, Dh, scale, out_res=res)
    x = qkv.double().requires_grad_(True)
    q, k, v = x.permute(2, 0, 3, 1, 4).unbind(0)
    ref = (((q @ k.transpose(-1, -2)) * scale).softmax(-1) @ v).transpose(1, 2).reshape(B, N, H * Dh)
    dO = torch.randn(B, N, H * Dh, generator=g, device="cuda").bfloat16()
    ref.backward(dO.double())
    dqkv, dsum = torch.zeros_like(qkv), torch.empty(B, H, N, device="cuda")
    ops.attention_bwd(qkv, out, dO, lse, dsum, dqkv, B, N, H, Dh, scale, out_res=res)
    e_q = _rel(dqkv[:, :, 0].float(), x.grad[:, :, 0])
    assert e_q < 2e-2 and _rel(dqkv[:, :, 1].float(), x.grad[:, :, 1]) < 1e-2, e_q
    ops.attention_bwd(qkv, out, dO, lse, dsum, dqkv, B, N, H, Dh, scale)
    assert _rel(dqkv[:, :, 0].float(), x.grad[:, :, 0]) > 3 * e_q      # what the residual buys


def test_attention_is_run_to_run_identical_at_the_benchmark_shape():
    """B = 16, N = 329, H = 24, Dh = 64 (BASELINE configs[1]): the same launch repeated must give the same bits.  Round 4 found the forward
    kernel returning, in ~2.5 % of launches, one 32-query slab with wrong output columns 32..63 (lse intact; 10-20 % off) -- invisible to
    tolerance tests over the whole tensor, visible as run-to-run differences (tools/debug/attn_race.py): the last fragment reads of a step
    were still in flight at the next step's s_barrier (attention.hip, the wait in front of it).  400 forward and 200 backward launches: the
    old forward kernel fails this with probability 1 - 4e-5 (the backward kernels' rate needed the step-level soak of
    tests/test_deterministic_gpu.py)."""
    import miphei_vit_amd.ops as ops
    B, N, H, Dh = 16, 329, 24, 64
    g = torch.Generator(device="cuda").manual_seed(1)
    qkv = torch.randn(B, N, 3, H, Dh, generator=g, device="cuda").bfloat16()
    scale = Dh ** -0.5
    out0, res0 = (torch.empty(B, N, H * Dh, device="cuda", dtype=torch.bfloat16) for _ in range(2))
    lse0 = torch.empty(B, H, N, device="cuda")
    ops.attention_fwd(qkv, out0, lse0, B, N, H, Dh, scale, out_res=res0)
    bad = 0
    for _ in range(400):
        out, res, lse = torch.empty_like(out0), torch.empty_like(res0), torch.empty_like(lse0)
        ops.attention_fwd(qkv, out, lse, B, N, H, Dh, scale, out_res=res)
        bad += int(not (torch.equal(out, out0) and torch.equal(res, res0) and torch.equal(lse, lse0)))
    assert bad == 0, f"{bad} of 400 forward launches differ from the first"
    dO = torch.randn(B, N, H * Dh, generator=g, device="cuda").bfloat16()
    dq0, ds0 = torch.zeros_like(qkv), torch.empty(B, H, N, device="cuda")
    ops.attention_bwd(qkv, out0, dO, lse0, ds0, dq0, B, N, H, Dh, scale, out_res=res0)
    for _ in range(200):
        dq, ds = torch.zeros_like(qkv), torch.empty_like(ds0)
        ops.attention_bwd(qkv, out0, dO, lse0, ds, dq, B, N, H, Dh, scale, out_res=res0)
        bad += int(not (torch.equal(dq, dq0) and torch.equal(ds, ds0)))
    assert bad == 0, f"{bad} of 200 backward launches differ from the first"
