"""Error behaviour of the C-ABI (include/miphei_hip.h: "return 0 / hipError_t / -1 on bad arguments", nothing is launched):
the ctypes layer turns a non-zero code into RuntimeError, which is what a caller of the reference surface sees."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _ops():
    import miphei_vit_amd.ops as ops
    return ops


def test_gemm_rejects_unaligned_k_and_bad_epilogue_operands():
    ops = _ops()
    a = torch.zeros(64, 20, device="cuda", dtype=torch.bfloat16)      # K = 20: not a multiple of 8
    b = torch.zeros(32, 20, device="cuda", dtype=torch.bfloat16)
    c = torch.zeros(64, 32, device="cuda", dtype=torch.bfloat16)
    with pytest.raises(RuntimeError, match="mvit_gemm_bf16"):
        ops.gemm(a, b, c)
    a = torch.zeros(64, 64, device="cuda", dtype=torch.bfloat16)
    b = torch.zeros(128, 64, device="cuda", dtype=torch.bfloat16)
    c = torch.zeros(64, 128, device="cuda", dtype=torch.bfloat16)
    with pytest.raises(RuntimeError):                                  # d(SwiGLU) epilogue without the saved pre-activation
        ops.gemm(a, b, c, epi=ops.EPI_DSWIGLU)
    with pytest.raises(RuntimeError):                                  # split-K without the atomic flag
        ops.gemm(a, b, c, ksplit=2)
    with pytest.raises(TypeError):                                     # fp32 operand: caught before the call
        ops.gemm(a.float(), b, c)


def test_heads_reject_too_many_heads_and_short_scratch():
    ops = _ops()
    M = 64
    x = torch.zeros(M, 32, device="cuda", dtype=torch.bfloat16)
    G = torch.zeros(M, 16, device="cuda", dtype=torch.bfloat16)
    z = lambda *s: torch.zeros(*s, device="cuda")
    with pytest.raises(RuntimeError, match="mvit_heads_gate_fwd"):
        ops.heads_gate_fwd(x, z(17 * 16, 32), z(17 * 16), z(17 * 16), z(17 * 16), z(17 * 16), z(17), G, M, 17)
    short = torch.zeros(8, device="cuda")
    with pytest.raises(RuntimeError, match="mvit_heads_conv_bwd"):
        ops.heads_conv_bwd(z(1, 2, 8, 8), z(1, 2, 8, 8), x, G, z(2, 9, 32), short, z(M, 16), z(M, 32), z(18, 32), z(64, 32), 1, 8, 8, 2)
    with pytest.raises(RuntimeError, match="mvit_heads_gate_bwd"):
        nch = 32
        ops.heads_gate_bwd(x, G, z(M, 16), z(M, 32), z(nch, 32), z(nch), z(nch), z(nch), z(nch), z(nch), z(nch), z(nch),
                           torch.zeros(1056, device="cuda", dtype=torch.float64), short, z(nch, 32), z(nch), z(nch), z(nch), z(2),
                           torch.zeros(M, 32, device="cuda", dtype=torch.bfloat16), M, 2)


def test_generator_refuses_cpu_tensors_and_wrong_sizes():
    from miphei_vit_amd.generators import get_vitmatte
    model = get_vitmatte("tiny", 128, 3, use_lora=True, pretrained=False)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        model(torch.zeros(1, 3, 128, 128))                              # model still on the CPU
    model = model.cuda().eval()
    with pytest.raises((ValueError, RuntimeError, AssertionError)):
        model(torch.zeros(1, 3, 96, 96, device="cuda"))                 # not the configured input size
    with pytest.raises(ValueError):
        model.set_input_size((200, 200))                                # reference guard: power of two, >= 128
