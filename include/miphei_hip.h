/*
 * libmiphei_hip.so -- C-ABI of the MI355X (gfx950) kernels behind the MIPHEI-ViT generator hot path.
 *
 * The reference (Sanofi-Public/MIPHEI-ViT) is pure PyTorch: it has no FFI for this path.  Each entry
 * point below replaces the ATen/cuDNN/cuBLAS work issued by one reference call site (cited per
 * function as /root/reference/<file>:<line>); the Python host in miphei-vit_amd/ binds them with
 * ctypes (INTEGRATION.md shows the stub a reference maintainer would add).
 *
 * Conventions
 *  - plain pointers are DEVICE pointers borrowed from the caller (torch tensors); nothing is
 *    allocated, freed or synchronised inside, and every launch goes to the hipStream_t passed
 *    as `stream` (a void* holding torch.cuda.current_stream().cuda_stream) -> hipGraph-capturable.
 *  - bf16 tensors are raw uint16 bit patterns; "f32" is IEEE float; row-major unless stated.
 *  - return 0 on success, a hipError_t value or MVIT_EINVAL (-1) for bad arguments.
 *  - thread-compatible: no global mutable state.
 */
#ifndef MIPHEI_HIP_H
#define MIPHEI_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MVIT_API __attribute__((visibility("default")))

typedef void* mvit_stream_t;

/* ---------------------------------------------------------------- GEMM / implicit-GEMM conv */
enum mvit_epilogue {
  MVIT_EPI_STORE = 0,    /* C = acc (+bias)                                                  */
  MVIT_EPI_GELU = 1,     /* u = acc+bias; aux<-u (optional); C = gelu_erf(u)                 */
  MVIT_EPI_SWIGLU = 2,   /* packed fc1: C[m, g] = silu(a)*b, aux<-[a|b] (optional)           */
  MVIT_EPI_RESID = 3,    /* C(f32, in place) += gamma[n]*(acc+bias[n])   (LayerScale + res)  */
  MVIT_EPI_PATCH = 4,    /* patch-embed: row remap past prefix tokens, +bias +pos_embed      */
  MVIT_EPI_STATS = 5,    /* C = bf16(acc) and per-column sum / sum-of-squares (BatchNorm)    */
  MVIT_EPI_DSWIGLU = 6,  /* C[m, packed a|b] = d(silu(a)*b) * acc, aux = saved [a|b]         */
  MVIT_EPI_DGELU = 7     /* C = acc * gelu'(aux)                                             */
};
enum mvit_gemm_flags { MVIT_OUT_F32 = 1, MVIT_ATOMIC = 2 };
enum mvit_amode { MVIT_A_DENSE = 0, MVIT_A_CONV3 = 1, MVIT_A_CONV3_T = 2 };

/*
 * C[M,N] = A[M,K] * B[N,K]^T (+ A2[M,K2] * B2[N,K2]^T), bf16 operands, f32 accumulate on MFMA.
 * K, K2, lda, ldb, lda2, ldb2 and conv_C / conv_ld must be multiples of 8 (16-byte operand loads).
 * amode CONV3: A is an NHWC bf16 activation [B, conv_H, conv_W, conv_ld] and row m=(b,oy,ox) of the
 *   virtual im2col matrix gathers the 3x3 window (pad 1, stride conv_stride), k=(ky,kx,c), K=9*conv_C.
 * amode CONV3_T: the adjoint gather (dgrad): rows m index the conv *input* grid [B,conv_OH,conv_OW],
 *   A is dY [B, conv_H, conv_W, conv_ld].
 * Replaces: every nn.Linear of the timm ViT built at src/generators/foundation_models.py:53-57,
 *   LoRA (src/generators/lora.py:16-18,29-33), nn.Conv2d in Basic_Conv3x3 (src/generators/mipheivit.py:32)
 *   and Fusion_Block (mipheivit.py:86), and their autograd backward (src/models.py:135).
 */
typedef struct mvit_gemm_args {
  const void* A; const void* B; void* C;
  const void* A2; const void* B2;
  const float* bias; const float* gamma;
  void* aux; const float* pos; double* stats;
  int M, N, K, K2;
  int lda, ldb, ldc, lda2, ldb2, ldaux;
  int epi, flags, ksplit, amode;
  int conv_H, conv_W, conv_C, conv_ld, conv_OH, conv_OW, conv_stride;
  int patch_P, patch_ntok, patch_prefix;
  int nslots;
} mvit_gemm_args;

MVIT_API int mvit_gemm_bf16(const mvit_gemm_args* args, mvit_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif
