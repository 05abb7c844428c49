"""ctypes binding of libmiphei_hip.so (C-ABI declared in include/miphei_hip.h).

The product path has no fallback: if the library is missing or a symbol is absent this raises.
"""
from __future__ import annotations

import contextlib
import ctypes as C
import os
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libmiphei_hip.so")
# measurement builds (make -C miphei-vit_amd/csrc dbg, or BUILD=... LIB=variants/... EXTRA=-D...) live apart from the product library and
# are only ever loaded by tools/ (or by pytest --variant-lib PATH): nothing in the package or the test suite reads the environment
VARIANTS_DIR = os.path.join(_HERE, "csrc", "variants")
DBG_LIB_PATH = os.path.join(VARIANTS_DIR, "libmiphei_hip_dbg.so")

# enum mvit_epilogue
EPI_STORE, EPI_GELU, EPI_SWIGLU, EPI_RESID, EPI_PATCH, EPI_STATS, EPI_DSWIGLU, EPI_DGELU = range(8)
OUT_F32, ATOMIC, ACCUM_BF16, RELU = 1, 2, 4, 8
A_DENSE, A_CONV3, A_CONV3_T, A_PATCH = 0, 1, 2, 3

vp, ci, cf, cd, ll = C.c_void_p, C.c_int, C.c_float, C.c_double, C.c_longlong


class GemmArgs(C.Structure):
    _fields_ = [
        ("A", vp), ("B", vp), ("C", vp), ("A2", vp), ("B2", vp), ("bias", vp), ("gamma", vp),
        ("aux", vp), ("pos", vp), ("stats", vp), ("rowscale", vp),
        ("M", ci), ("N", ci), ("K", ci), ("K2", ci),
        ("lda", ci), ("ldb", ci), ("ldc", ci), ("lda2", ci), ("ldb2", ci), ("ldaux", ci),
        ("epi", ci), ("flags", ci), ("ksplit", ci), ("amode", ci),
        ("conv_H", ci), ("conv_W", ci), ("conv_C", ci), ("conv_ld", ci), ("conv_OH", ci), ("conv_OW", ci),
        ("conv_stride", ci),
        ("patch_P", ci), ("patch_ntok", ci), ("patch_prefix", ci),
        ("nslots", ci),
    ]


class GemmTnArgs(C.Structure):
    _fields_ = [("A", vp), ("B", vp), ("C", vp), ("ldci", ll), ("ldcj", ll),
                ("M", ci), ("I", ci), ("J", ci), ("lda", ci), ("ldb", ci), ("amode", ci), ("msplit", ci),
                ("conv_H", ci), ("conv_W", ci), ("conv_C", ci), ("conv_ld", ci), ("conv_OH", ci), ("conv_OW", ci),
                ("conv_stride", ci), ("C2", vp), ("isplit", ci), ("j1", ci), ("jlo2", ci), ("batch", ci),
                ("strideA", ll), ("strideB", ll), ("strideC", ll), ("split_stride", ll)]


class ConvPackDesc(C.Structure):
    _fields_ = [("W", vp), ("wk", vp), ("wd", vp), ("Cout", ci), ("Cin", ci), ("Cp", ci), ("rot", ci)]


class CcPackDesc(C.Structure):
    _fields_ = [("W", vp), ("out", vp), ("Cout", ci), ("Cin", ci), ("mode", ci), ("pad_", ci)]


class ConvUnpackDesc(C.Structure):
    _fields_ = [("dWt", vp), ("dW", vp), ("Cout", ci), ("Cin", ci), ("Cp", ci), ("rot", ci), ("accumulate", ci), ("n_major", ci)]


# name -> argtypes (restype is always int); mirrors include/miphei_hip.h
SIGNATURES = {
    "mvit_gemm_bf16": [C.POINTER(GemmArgs), vp],
    "mvit_gemm_variant": [C.POINTER(GemmArgs)],
    "mvit_gemm_tn_bf16": [C.POINTER(GemmTnArgs), vp],
    "mvit_layernorm_fwd": [vp, vp, vp, vp, ci, ci, cf, vp],
    "mvit_layernorm_lora_fwd": [vp, vp, vp, vp, vp, vp, ci, ci, cf, ci, vp],
    "mvit_conv3x3_direct_supported": [ci, ci],
    "mvit_conv3x3_direct": [vp, vp, vp, vp, ci, ci, ci, ci, ci, ci, ci, ci, vp],
    "mvit_conv3x3_chunked_pack_elems": [ci, ci],
    "mvit_conv3x3_chunked_pack": [vp, vp, ci, ci, ci, vp],
    "mvit_conv3x3_chunked_pack_multi": [C.POINTER(CcPackDesc), ci, vp],
    "mvit_conv3x3_chunked": [vp, vp, vp, vp, ci, ci, ci, ci, ci, ci, ci, ci, vp],
    "mvit_conv3x3_chunked_wgrad": [vp, vp, vp, ci, ci, ci, ci, ci, ci, ci, ci, vp],
    "mvit_pack_conv3x3_direct": [vp, vp, ci, ci, ci, ci, ci, ci, ci, vp],
    "mvit_lora_pack": [vp, vp, vp, vp, vp, ci, ci, ci, cf, vp],
    "mvit_unpack_conv3x3_wgrad": [vp, vp, ci, ci, ci, ci, ci, ci, vp],
    "mvit_conv3x3_direct_wgrad": [vp, vp, vp, ci, ci, ci, ci, ci, ci, ci, vp],
    "mvit_layernorm_bwd": [vp, vp, vp, vp, vp, vp, ci, ci, cf, ci, vp, vp],
    "mvit_skinny_xw": [vp, ci, vp, ci, vp, ci, ci, ci, ci, vp],
    "mvit_skinny_xw2": [vp, vp, vp, vp, vp, vp, ci, ci, ci, ci, ci, ci, vp],
    "mvit_prefix_tokens": [vp, vp, vp, ci, ci, ci, ci, vp],
    "mvit_cast_f32_bf16": [vp, vp, C.c_longlong, vp],
    "mvit_scale_cols_cast": [vp, vp, vp, ci, ci, vp, vp],
    "mvit_attention_fwd": [vp, vp, vp, vp, ci, ci, ci, ci, cf, vp],
    "mvit_attention_bwd": [vp, vp, vp, vp, vp, vp, vp, ci, ci, ci, ci, cf, vp],
    "mvit_resample2d": [vp, vp, vp, vp, vp, vp, vp, vp, ci, ci, ci, ci, ci, ci, ci, ci, ll, ll, ci, vp],
    "mvit_pack_conv3x3_weights_multi": [C.POINTER(ConvPackDesc), ci, vp],
    "mvit_unpack_conv3x3_wgrad_multi": [C.POINTER(ConvUnpackDesc), ci, vp],
    "mvit_upsample2x_bilinear": [vp, vp, vp, vp, vp, ci, ci, ci, ci, ci, ci, ll, ll, vp],
    "mvit_upsample2x_bilinear_bwd": [vp, vp, ci, ci, ci, ci, ci, ci, ll, ll, vp],
    "mvit_image_to_nhwc": [vp, vp, ci, ci, ci, ci, ci, vp],
    "mvit_bn_finalize": [vp, vp, vp, vp, vp, vp, vp, vp, vp, ci, ci, cd, cf, cf, ci, vp],
    "mvit_bn_relu_apply": [vp, vp, vp, vp, ll, ci, ci, ci, cf, C.c_ulonglong, vp],
    "mvit_bn_relu_bwd_reduce": [vp, ci, vp, vp, vp, vp, vp, vp, ll, ci, ci, cf, C.c_ulonglong, vp],
    "mvit_bn_relu_bwd_apply": [vp, ci, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, ll, ci, ci, cd, cf, C.c_ulonglong, vp],
    "mvit_pack_conv3x3_weights": [vp, vp, vp, ci, ci, ci, ci, vp],
    "mvit_pixel_shuffle2x": [vp, vp, ci, ci, ci, ci, ll, ci, vp],
    "mvit_transpose_bf16": [vp, vp, ci, ci, ci, ll, vp],
    "mvit_heads_moments": [vp, vp, ll, ci, vp],
    "mvit_heads_bn_from_moments": [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, ci, ci, cd, cf, cf, ci, vp],
    "mvit_heads_gate_fwd": [vp, vp, vp, vp, vp, vp, vp, vp, ll, ci, vp],
    "mvit_heads_conv_fwd": [vp, vp, vp, vp, vp, ci, ci, ci, ci, vp],
    "mvit_heads_conv_bwd_scratch_bytes": [ll],
    "mvit_heads_conv_bwd": [vp, vp, vp, vp, vp, vp, ll, vp, vp, vp, vp, ci, ci, ci, ci, vp],
    "mvit_heads_gate_bwd_scratch_bytes": [],
    "mvit_heads_gate_bwd": [vp] * 14 + [ll] + [vp] * 6 + [ll, ci, cd, vp],
    "mvit_pix_metrics_scratch_bytes": [ci],
    "mvit_pix_metrics_update": [vp, vp, vp, vp, ll, ci, ci, ci, ci, cf, cf, vp],
    "mvit_wmse_fwd_bwd": [vp, vp, vp, vp, vp, ci, ci, ll, cf, vp],
    "mvit_sqnorm": [vp, vp, ll, vp],
    "mvit_sqnorm_ordered": [vp, vp, vp, ll, vp],
    "mvit_u8_nhwc_to_f32_nchw": [vp, vp, vp, vp, ci, ci, ll, vp],
    "mvit_f32_to_u8_export": [vp, vp, ll, vp],
    "mvit_augment_tiles": [vp, vp, vp, vp, vp, ci, ci, ci, ci, ci, ci, C.c_ulonglong, C.c_ulonglong, cf, cf, cf, cf, vp, vp, vp],
    "mvit_augment_draw": [ci, ci, ci, ci, C.c_ulonglong, C.c_ulonglong, cf, cf, cf, cf, vp],
    "mvit_cell_means": [vp, vp, vp, ci, ci, ci, ci, ci, cf, ci, ci, vp, vp, vp, vp, vp, vp, vp, vp, vp],
    "mvit_adam_clip_step": [vp, vp, vp, vp, vp, ll, cf, cf, cf, cf, cf, cf, cf, vp, vp],
    "mvit_occupy_cus": [ci, ci, vp],
}

# Two product libraries from the same sources (csrc/Makefile): bf16 operands (training, BASELINE.json's precision) and IEEE fp16 operands
# (`generator.eval().cuda().half()`, the reference's evaluation convention, /root/reference/evaluation/eval_orion.py:191, 214-215).
# Which one a call reaches is a per-thread mode set by the engine for the duration of a forward pass (`operands("f16")`); everything
# outside such a block is the bf16 library.
LIB_PATH_F16 = os.path.join(_HERE, "libmiphei_hip_f16.so")
_libs = {}
_mode = threading.local()


def operand_mode() -> str:
    return getattr(_mode, "v", "bf16")


def operand_torch_dtype():
    import torch
    return torch.float16 if operand_mode() == "f16" else torch.bfloat16


@contextlib.contextmanager
def operands(mode: str):
    """`with operands("f16"):` -- every ops.* call of this thread goes to the fp16-operand library (16-bit tensors are torch.float16)."""
    if mode not in ("bf16", "f16"):
        raise ValueError(f"operand mode {mode!r}: 'bf16' or 'f16'")
    prev = operand_mode()
    _mode.v = mode
    try:
        yield
    finally:
        _mode.v = prev


def lib():
    """The HIP library of the current operand mode, loaded once; fails loudly when it is missing (no CPU/PyTorch fallback)."""
    mode = operand_mode()
    handle = _libs.get(mode)
    if handle is None:
        path = LIB_PATH_F16 if mode == "f16" else LIB_PATH
        if not os.path.exists(path):
            raise RuntimeError(
                f"{path} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950). The MI355X path has no fallback.")
        # torch must load ITS HIP runtime first: libmiphei_hip.so then binds to that same libamdhip64 instance (same
        # soname).  Loaded the other way round the process ends up with two runtimes and every launch on a torch stream
        # fails with hipErrorNoDevice.
        import torch  # noqa: F401
        handle = C.CDLL(path)
        for name, argtypes in SIGNATURES.items():
            fn = getattr(handle, name)  # AttributeError if the symbol is not exported
            fn.argtypes = argtypes
            fn.restype = C.c_longlong if name.endswith(("_bytes", "_elems")) else C.c_int
        _libs[mode] = handle
    return handle


def check(rc: int, what: str):
    if rc != 0:
        raise RuntimeError(f"{what} failed with code {rc}")
