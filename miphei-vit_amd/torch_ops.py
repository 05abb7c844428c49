"""``torch.library`` registration of the per-kernel entry points ("registered through torch.library custom
ops with autograd so Lightning / AMP / DDP see ordinary ops").

The fused engine (``engine.py``) sequences the C-ABI itself and does not go through these; they exist so that an ``nn.Module``
graph written the reference's way -- timm ``Block`` / ``Attention`` / ``Mlp`` modules (``/root/reference/src/generators/
foundation_models.py:53-57``), ``WeightedMSELoss`` (``/root/reference/src/loss.py:47-57``) -- can call the same HIP kernels op by op:

    torch.ops.mvit.linear(x, weight, bias)             nn.Linear / F.linear        -> mvit_gemm_bf16 (+ TN GEMM for dW)
    torch.ops.mvit.layer_norm(x, weight, bias, eps)    nn.LayerNorm(eps=1e-6)      -> mvit_layernorm_fwd / _bwd
    torch.ops.mvit.attention(qkv, num_heads)           F.scaled_dot_product_attention on the packed qkv of timm Attention
                                                                                   -> mvit_attention_fwd / _bwd
    torch.ops.mvit.weighted_mse(pred, target, w, lam)  WeightedMSELoss.forward     -> mvit_wmse_fwd_bwd

Arithmetic is the reference's bf16-mixed mode: bf16 operands, f32 accumulation, f32 LayerNorm statistics.  Every op has a fake
(meta) implementation, so FakeTensor tracing / torch.compile shape propagation work, and an autograd formula built on the
backward kernels.  CUDA (ROCm) tensors only: there is no CPU implementation -- a CPU call raises from the dispatcher.
"""
from __future__ import annotations

from typing import Optional

import torch
from torch import Tensor

from . import ops
from .ops import OUT_F32

_BF = torch.bfloat16


def _as_bf16_2d(x: Tensor) -> Tensor:
    x2 = x.reshape(-1, x.shape[-1])
    if x2.dtype != _BF:
        x2 = x2.to(_BF)
    return x2.contiguous()


# ------------------------------------------------------------------ linear
@torch.library.custom_op("mvit::linear", mutates_args=(), device_types="cuda")
def linear(x: Tensor, weight: Tensor, bias: Optional[Tensor] = None) -> Tensor:
    """y = x @ weight^T + bias on the bf16 MFMA GEMM.  Output dtype: bf16 for bf16 / fp16 input (autocast), else f32."""
    K, N = x.shape[-1], weight.shape[0]
    if weight.dim() != 2 or weight.shape[1] != K or K % 8:
        raise ValueError(f"mvit::linear: weight {tuple(weight.shape)} does not match input [..., {K}] (K must be a multiple of 8)")
    a, b = _as_bf16_2d(x), (weight if weight.dtype == _BF else weight.to(_BF)).contiguous()
    low = x.dtype in (_BF, torch.float16)
    out = torch.empty(a.shape[0], N, device=x.device, dtype=_BF if low else torch.float32)
    ops.gemm(a, b, out, bias=None if bias is None else bias.float().contiguous(), flags=0 if low else OUT_F32)
    out = out.view(*x.shape[:-1], N)
    return out.to(x.dtype) if low and x.dtype != _BF else out


@linear.register_fake
def _(x, weight, bias=None):
    low = x.dtype in (_BF, torch.float16)
    return x.new_empty(*x.shape[:-1], weight.shape[0], dtype=x.dtype if low else torch.float32)


@torch.library.custom_op("mvit::linear_backward", mutates_args=(), device_types="cuda")
def linear_backward(grad: Tensor, x: Tensor, weight: Tensor, need_dw: bool, need_db: bool) -> tuple[Tensor, Tensor, Tensor]:
    K, N = x.shape[-1], weight.shape[0]
    g, a = _as_bf16_2d(grad), _as_bf16_2d(x)
    M = g.shape[0]
    wt = torch.empty(K, N, device=x.device, dtype=_BF)            # B operand of dx = g @ W is W^T, K-contiguous
    w16 = (weight if weight.dtype == _BF else weight.to(_BF)).contiguous()
    ops.transpose_bf16(w16, wt, N, K, K, N)
    low = x.dtype in (_BF, torch.float16)
    dx = torch.empty(M, K, device=x.device, dtype=_BF if low else torch.float32)
    ops.gemm(g, wt, dx, flags=0 if low else OUT_F32)
    dw = torch.zeros(N, K, device=x.device, dtype=torch.float32) if need_dw else torch.empty(0, device=x.device)
    if need_dw:   # dW[n, k] = sum_m g[m, n] x[m, k]: both operands m-major as produced (TN MFMA GEMM, split over m)
        ops.gemm_tn(g, a, dw, M=M, I=N, J=K, lda=N, ldb=K, ldci=K, ldcj=1, msplit=max(1, min(32, M // 256)))
    db = g.float().sum(0) if need_db else torch.empty(0, device=x.device)
    return dx.view(x.shape).to(x.dtype), dw.to(weight.dtype), db


@linear_backward.register_fake
def _(grad, x, weight, need_dw, need_db):
    return (torch.empty_like(x), torch.empty_like(weight) if need_dw else x.new_empty(0),
            x.new_empty(weight.shape[0], dtype=torch.float32) if need_db else x.new_empty(0))


def _linear_setup(ctx, inputs, output):
    x, weight, bias = inputs
    ctx.save_for_backward(x, weight)
    ctx.has_bias = bias is not None


def _linear_bwd(ctx, grad):
    x, weight = ctx.saved_tensors
    need_dw, need_db = ctx.needs_input_grad[1], ctx.has_bias and ctx.needs_input_grad[2]
    dx, dw, db = torch.ops.mvit.linear_backward(grad.contiguous(), x, weight, need_dw, need_db)
    return dx, (dw if need_dw else None), (db if need_db else None)


linear.register_autograd(_linear_bwd, setup_context=_linear_setup)


# ------------------------------------------------------------------ layer norm
@torch.library.custom_op("mvit::layer_norm", mutates_args=(), device_types="cuda")
def layer_norm(x: Tensor, weight: Tensor, bias: Tensor, eps: float = 1e-6) -> Tensor:
    """bf16(LayerNorm(x)) over the last dimension, statistics in f32 (x is the f32 residual stream or a bf16 activation)."""
    D = x.shape[-1]
    if D % 4 or D > 2048:
        raise ValueError("mvit::layer_norm: last dimension must be a multiple of 4 and <= 2048")
    x2 = x.reshape(-1, D).float().contiguous()
    out = torch.empty(x2.shape, device=x.device, dtype=_BF)
    ops.layernorm_fwd(x2, weight.float().contiguous(), bias.float().contiguous(), out, float(eps))
    return out.view(x.shape)


@layer_norm.register_fake
def _(x, weight, bias, eps=1e-6):
    return x.new_empty(x.shape, dtype=_BF)


@torch.library.custom_op("mvit::layer_norm_backward", mutates_args=(), device_types="cuda")
def layer_norm_backward(grad: Tensor, x: Tensor, weight: Tensor, eps: float) -> Tensor:
    D = x.shape[-1]
    x2 = x.reshape(-1, D).float().contiguous()
    dx = torch.empty_like(x2)
    ops.layernorm_bwd(_as_bf16_2d(grad), x2, weight.float().contiguous(), dx, None, None, float(eps), accumulate=False)
    return dx.view(x.shape).to(x.dtype)


@layer_norm_backward.register_fake
def _(grad, x, weight, eps):
    return torch.empty_like(x)


def _ln_setup(ctx, inputs, output):
    x, weight, bias, eps = inputs
    ctx.save_for_backward(x, weight)
    ctx.eps = eps


def _ln_bwd(ctx, grad):
    x, weight = ctx.saved_tensors
    dx = torch.ops.mvit.layer_norm_backward(grad.contiguous(), x, weight, ctx.eps) if ctx.needs_input_grad[0] else None
    dw = db = None
    if ctx.needs_input_grad[1] or ctx.needs_input_grad[2]:
        # affine gradients: tiny reductions, only needed when the norm is trained (the MIPHEI-ViT recipe freezes it)
        xf, g = x.float(), grad.float()
        xhat = (xf - xf.mean(-1, keepdim=True)) * torch.rsqrt(xf.var(-1, unbiased=False, keepdim=True) + ctx.eps)
        red = tuple(range(x.dim() - 1))
        dw = (g * xhat).sum(red).to(weight.dtype) if ctx.needs_input_grad[1] else None
        db = g.sum(red).to(weight.dtype) if ctx.needs_input_grad[2] else None
    return dx, dw, db, None


layer_norm.register_autograd(_ln_bwd, setup_context=_ln_setup)


# ------------------------------------------------------------------ attention
@torch.library.custom_op("mvit::attention_forward", mutates_args=(), device_types="cuda")
def attention_forward(qkv: Tensor, num_heads: int) -> tuple[Tensor, Tensor, Tensor]:
    B, N, C3 = qkv.shape
    D = C3 // 3
    Dh = D // num_heads
    if C3 % 3 or D % num_heads or Dh % 8 or Dh > 64:
        raise ValueError("mvit::attention: qkv must be [B, N, 3*H*Dh] with Dh a multiple of 8, <= 64")
    q = (qkv if qkv.dtype == _BF else qkv.to(_BF)).contiguous()
    out = torch.empty(B, N, D, device=qkv.device, dtype=_BF)
    lse = torch.empty(B, num_heads, N, device=qkv.device, dtype=torch.float32)
    res = torch.empty_like(out)       # bf16 rounding residual of out: the backward pass's D term (see include/miphei_hip.h)
    ops.attention_fwd(q, out, lse, B, N, num_heads, Dh, Dh ** -0.5, out_res=res)
    return out, lse, res


@attention_forward.register_fake
def _(qkv, num_heads):
    B, N, C3 = qkv.shape
    return (qkv.new_empty(B, N, C3 // 3, dtype=_BF), qkv.new_empty(B, num_heads, N, dtype=torch.float32),
            qkv.new_empty(B, N, C3 // 3, dtype=_BF))


@torch.library.custom_op("mvit::attention_backward", mutates_args=(), device_types="cuda")
def attention_backward(grad: Tensor, qkv: Tensor, out: Tensor, lse: Tensor, out_res: Tensor, num_heads: int) -> Tensor:
    B, N, C3 = qkv.shape
    Dh = C3 // 3 // num_heads
    q = (qkv if qkv.dtype == _BF else qkv.to(_BF)).contiguous()
    dqkv = torch.empty_like(q)
    dsum = torch.empty(B, num_heads, N, device=qkv.device, dtype=torch.float32)
    ops.attention_bwd(q, out, (grad if grad.dtype == _BF else grad.to(_BF)).contiguous(), lse, dsum, dqkv, B, N, num_heads, Dh, Dh ** -0.5,
                      out_res=out_res)
    return dqkv.to(qkv.dtype)


@attention_backward.register_fake
def _(grad, qkv, out, lse, out_res, num_heads):
    return torch.empty_like(qkv)


def attention(qkv: Tensor, num_heads: int) -> Tensor:
    """softmax(q k^T / sqrt(Dh)) v for the packed projection of timm ``Attention`` ([B, N, 3, H, Dh] flattened): [B, N, H*Dh]."""
    return torch.ops.mvit.attention_forward(qkv, num_heads)[0]


def _attn_setup(ctx, inputs, output):
    qkv, num_heads = inputs
    out, lse, res = output
    ctx.save_for_backward(qkv, out, lse, res)
    ctx.num_heads = num_heads
    ctx.mark_non_differentiable(lse, res)


def _attn_bwd(ctx, grad_out, _grad_lse, _grad_res):
    qkv, out, lse, res = ctx.saved_tensors
    return torch.ops.mvit.attention_backward(grad_out.contiguous(), qkv, out, lse, res, ctx.num_heads), None


attention_forward.register_autograd(_attn_bwd, setup_context=_attn_setup)


# ------------------------------------------------------------------ weighted MSE
@torch.library.custom_op("mvit::weighted_mse_forward", mutates_args=(), device_types="cuda")
def weighted_mse_forward(pred: Tensor, target: Tensor, marker_weights: Tensor, lambda_factor: float) -> tuple[Tensor, Tensor]:
    if pred.dim() != 4 or pred.shape != target.shape:
        raise ValueError("mvit::weighted_mse: pred and target must be equal-shaped [B, C, H, W]")
    p, t = pred.float().contiguous(), target.float().contiguous()
    B, C, H, W = p.shape
    acc = torch.zeros(1, device=pred.device, dtype=torch.float64)
    d = torch.empty_like(p)
    ops.wmse_fwd_bwd(p, t, marker_weights.float().contiguous(), acc, d, float(lambda_factor))
    return (acc * (float(lambda_factor) / (C * B * H * W))).float().reshape(()), d


@weighted_mse_forward.register_fake
def _(pred, target, marker_weights, lambda_factor):
    return pred.new_empty((), dtype=torch.float32), pred.new_empty(pred.shape, dtype=torch.float32)


def weighted_mse(pred: Tensor, target: Tensor, marker_weights: Tensor, lambda_factor: float) -> Tensor:
    return torch.ops.mvit.weighted_mse_forward(pred, target, marker_weights, lambda_factor)[0]


def _wmse_setup(ctx, inputs, output):
    ctx.save_for_backward(output[1])
    ctx.mark_non_differentiable(output[1])
    ctx.in_dtype = inputs[0].dtype


def _wmse_bwd(ctx, grad_loss, _grad_d):
    (d,) = ctx.saved_tensors
    return (grad_loss * d).to(ctx.in_dtype), None, None, None


weighted_mse_forward.register_autograd(_wmse_bwd, setup_context=_wmse_setup)

__all__ = ["linear", "layer_norm", "attention", "weighted_mse"]
