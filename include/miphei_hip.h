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
 *  - "bf16" tensors are raw uint16 bit patterns: bf16 in libmiphei_hip.so; the same sources built with -DMVIT_F16 (libmiphei_hip_f16.so,
 *    the reference's evaluation convention generator.eval().cuda().half()) read and write IEEE fp16 in the same places; "f32" is IEEE float; row-major unless stated.
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
  MVIT_EPI_RESID = 3,    /* C(f32) = (aux f32 | C) + gamma[n]*(acc+bias[n]) (LayerScale+res) */
  MVIT_EPI_PATCH = 4,    /* patch-embed: row remap past prefix tokens, +bias +pos_embed      */
  MVIT_EPI_STATS = 5,    /* C = bf16(acc) and per-column sum / sum-of-squares (BatchNorm)    */
  MVIT_EPI_DSWIGLU = 6,  /* C[m, packed a|b] = d(silu(a)*b) * acc, aux = saved [a|b]         */
  MVIT_EPI_DGELU = 7     /* C = acc * gelu'(aux)                                             */
};
enum mvit_gemm_flags {
  MVIT_OUT_F32 = 1,
  MVIT_ATOMIC = 2,      /* f32 atomicAdd (split-K) */
  MVIT_ACCUM_BF16 = 4,  /* C(bf16) += */
  MVIT_RELU = 8         /* C = max(acc + bias, 0): convolution with an eval-mode BatchNorm folded into its weights (scale) and bias
                         * (shift) and the ReLU behind it (Basic_Conv3x3, src/generators/mipheivit.py:20-41, in model.eval());
                         * EPI_STORE with a CONV3 operand and bf16 output only */
};
enum mvit_amode { MVIT_A_DENSE = 0, MVIT_A_CONV3 = 1, MVIT_A_CONV3_T = 2, MVIT_A_PATCH = 3 };

/*
 * C[M,N] = A[M,K] * B[N,K]^T (+ A2[M,K2] * B2[N,K2]^T), bf16 operands, f32 accumulate on MFMA.
 * K, K2, lda, ldb, lda2, ldb2 and conv_C / conv_ld must be multiples of 8 (16-byte operand loads).
 * amode CONV3: A is an NHWC bf16 activation [B, conv_H, conv_W, conv_ld] and row m=(b,oy,ox) of the
 *   virtual im2col matrix gathers the 3x3 window (pad 1, stride conv_stride), k=(ky,kx,c), K=9*conv_C.
 * amode CONV3_T: the adjoint gather (dgrad): rows m index the conv *input* grid [B,conv_OH,conv_OW],
 *   A is dY [B, conv_H, conv_W, conv_ld].
 * amode PATCH: the patch-embedding convolution (kernel = stride = conv_stride, timm PatchEmbed.proj built at
 *   src/generators/foundation_models.py:53-57) as an in-kernel window gather: A is the bf16 NHWC image [B, conv_H, conv_W, 8]
 *   (3 colour channels + 5 zeros: one 16-byte piece per pixel, the decoder's image operand), row m = (b, py, px) of the
 *   conv_OH x conv_OW patch grid, k = (dy * conv_stride + dx) * 8 + c, K = conv_stride^2 * 8; pixels beyond the last whole
 *   patch are ignored; epilogue EPI_PATCH only.  No im2col buffer exists.
 * Replaces: every nn.Linear of the timm ViT built at src/generators/foundation_models.py:53-57,
 *   LoRA (src/generators/lora.py:16-18,29-33), nn.Conv2d in Basic_Conv3x3 (src/generators/mipheivit.py:32)
 *   and Fusion_Block (mipheivit.py:86), and their autograd backward (src/models.py:135).
 */
typedef struct mvit_gemm_args {
  const void* A; const void* B; void* C;
  const void* A2; const void* B2;
  const float* bias; const float* gamma;
  void* aux; const float* pos; double* stats;
  const float* rowscale;   /* EPI_RESID: optional per-row factor of the branch, C = aux + rowscale[row]*gamma*(acc+bias): timm DropPath
                              (stochastic depth, per sample) around the LayerScale'd branch; NULL = 1 */
  int M, N, K, K2;
  int lda, ldb, ldc, lda2, ldb2, ldaux;
  int epi, flags, ksplit, amode;
  int conv_H, conv_W, conv_C, conv_ld, conv_OH, conv_OW, conv_stride;
  int patch_P, patch_ntok, patch_prefix;
  int nslots;
} mvit_gemm_args;

MVIT_API int mvit_gemm_bf16(const mvit_gemm_args* args, mvit_stream_t stream);
/* which tile variant mvit_gemm_bf16 runs for this problem: (BM << 20) | (BN << 8) | (WAVES_M << 4) | WAVES_N, with bit 30 set when
 * the wave-specialised kernel (csrc/gemm_ws.hip) takes it
 * (measurement only: bench.py attributes its HIP-event timings to one kernel instantiation with it) */
MVIT_API int mvit_gemm_variant(const mvit_gemm_args* args);

/*
 * C(f32)[i*ldci + j*ldcj] += sum_m A[m,i] * B[m,j]: both operands m-major bf16 (the contraction runs over ROWS), MFMA
 * fragments gathered with ds_read_b64_tr_b16.  amode DENSE: A = [M, I] (lda).  amode CONV3: A = virtual im2col of an
 * NHWC activation [B, conv_H, conv_W, conv_ld] (3x3, pad 1, stride conv_stride), row m = output pixel (b,oy,ox),
 * I = 9*conv_C.  msplit blocks share the M range (f32 atomics; C must be pre-zeroed or hold the running sum).
 * Weight gradients of nn.Conv2d (src/generators/mipheivit.py:32,86), LoRA dA/dB (src/generators/lora.py:11-12) and the
 * head convolutions (src/generators/unet.py:431), i.e. the parameter side of loss.backward() (src/models.py:135).
 */
typedef struct mvit_gemm_tn_args {
  const void* A; const void* B; float* C;
  long long ldci, ldcj;
  int M, I, J, lda, ldb, amode, msplit;
  int conv_H, conv_W, conv_C, conv_ld, conv_OH, conv_OW, conv_stride;
  /* optional second output from the same pass (NULL = off): rows i >= isplit go to C2[(i-isplit), ...].  With jlo2 > 0
   * the columns are split too: rows < isplit keep columns j < j1 (in C), rows >= isplit keep columns j >= jlo2
   * (C2, column j - jlo2), everything else is dropped - the two LoRA adapters' dB from one pass over [dq | dk | dv]. */
  float* C2;
  int isplit, j1, jlo2;
  /* batch > 1 (dense A only): `batch` independent products from one launch; product b reads A + b*strideA, B + b*strideB
   * and accumulates into C + b*strideC (and C2 + b*strideC), strides in elements - the LoRA weight gradients of a group of
   * ViT blocks, whose per-block launches (4 steps of m each) are latency-bound. */
  int batch;
  long long strideA, strideB, strideC;
  /* split_stride > 0: slice z of the m range accumulates into C + z * split_stride (and C2 + z * split_stride) instead of
   * sharing C through f32 atomics: one contributing block per element, the caller adds the msplit copies in a fixed order
   * (the deterministic mode of the host, MIPHEI_DETERMINISTIC=1); 0 = shared output */
  long long split_stride;
} mvit_gemm_tn_args;
MVIT_API int mvit_gemm_tn_bf16(const mvit_gemm_tn_args* args, mvit_stream_t stream);

/* ---------------------------------------------------------------- row-wise encoder kernels */
/* out(bf16)[M,D] = LayerNorm(x f32 [M,D]) * w + b, biased variance, eps inside the sqrt.
 * Replaces timm Block.norm1/norm2 and VisionTransformer.norm (nn.LayerNorm eps 1e-6) reached through
 * src/generators/foundation_models.py:53-57 and Encoder.forward src/generators/mipheivit.py:154. */
MVIT_API int mvit_layernorm_fwd(const float* x, const float* w, const float* b, void* out_bf16, int M, int D, float eps,
                                mvit_stream_t stream);
/* LayerNorm forward fused with the LoRA down-projection of the block's qkv adapter: out = bf16(LN(x)) as above and
 * t[M,R2] = out @ AcatT^T (bf16, f32 accumulate), AcatT [R2, D] = rows of [A_q | A_v]^T, R2 = 2*rank <= 16.
 * Replaces `self.lora_q.A` / `self.lora_v.A` products of QkvWithLoRA.forward (src/generators/lora.py:16-18,29-33) on the LN1
 * output of timm Block.forward. */
MVIT_API int mvit_layernorm_lora_fwd(const float* x, const float* w, const float* b, void* out, const void* AcatT, void* t,
                                     int M, int D, float eps, int R2, mvit_stream_t stream);
/* dx (+)= dLN/dx(dh); statistics recomputed from x.  If gamma_next/dy are given also writes
 * dy(bf16) = rowscale_next[row] * gamma_next * dx_total (the LayerScale-scaled gradient of the preceding residual branch;
 * rowscale_next = that branch's per-sample DropPath factors, NULL = 1). */
MVIT_API int mvit_layernorm_bwd(const void* dh_bf16, const float* x, const float* w, float* dx, const float* gamma_next,
                                void* dy_bf16, int M, int D, float eps, int accumulate, const float* rowscale_next,
                                mvit_stream_t stream);
/* out(bf16)[M,R] = X(bf16)[M,K] @ W(bf16)[R,K]^T, R <= 16 (one wave per 16 rows on the 16x16x32 MFMA).
 * LoRALayer.forward x @ A (src/generators/lora.py:16-18) and its adjoint dq @ B^T. */
MVIT_API int mvit_skinny_xw(const void* X, int ldx, const void* W, int ldw, void* out, int ldo, int M, int K, int R,
                            mvit_stream_t stream);
/* two such products of one shape in one launch (the q and v adapters' dt = dq @ B_q^T, dv @ B_v^T) */
MVIT_API int mvit_skinny_xw2(const void* X0, const void* W0, void* out0, const void* X1, const void* W1, void* out1, int ldx,
                             int ldw, int ldo, int M, int K, int R, mvit_stream_t stream);
/* x[b,0]=cls, x[b,1..R]=reg  (timm _pos_embed with no_embed_class=True). */
MVIT_API int mvit_prefix_tokens(float* x, const float* cls, const float* reg, int B, int ntok, int D, int R,
                                mvit_stream_t stream);
MVIT_API int mvit_cast_f32_bf16(const float* src, void* dst_bf16, long long n, mvit_stream_t stream);
/* out(bf16)[M,D] = x(f32)[M,D] * gamma[D] * rowscale[M] (rowscale may be NULL) */
MVIT_API int mvit_scale_cols_cast(const float* x, const float* gamma, void* out_bf16, int M, int D, const float* rowscale,
                                  mvit_stream_t stream);

/* ---------------------------------------------------------------- fused multi-head attention */
/* out(bf16)[B,N,H*Dh] = softmax(q k^T * scale) v per head, reading the packed projection qkv(bf16)[B,N,3,H,Dh];
 * lse(f32)[B,H,N] (optional) = log-sum-exp of the scaled scores, kept for the backward pass.  Dh <= 64, Dh % 8 == 0.
 * out_res(bf16)[B,N,H*Dh] (optional, training) = the bf16 rounding residual of out: the backward pass forms
 * D = sum_d dO * (out + out_res), consistent with its dP to ~16 bits (with the bf16 out alone dQ loses 20-30 % where attention is
 * near-uniform: the D error does not cancel in dS = P (dP - D) as it does in the unfused softmax backward).
 * Replaces F.scaled_dot_product_attention inside timm Attention (model built at
 * src/generators/foundation_models.py:53-57; q,v carry LoRA deltas from src/generators/lora.py:29-33). */
MVIT_API int mvit_attention_fwd(const void* qkv, void* out, void* out_res, float* lse, int B, int N, int H, int Dh, float scale,
                                mvit_stream_t stream);
/* dqkv(bf16)[B,N,3,H,Dh] from d_out(bf16)[B,N,H*Dh]: the backward of the call above (loss.backward() through timm's Attention;
 * the q and v gradients feed the LoRA adapters, src/generators/lora.py:29-33).  dsum(f32)[B,H,N] is caller-provided and receives
 * D = sum_d dO * (out + out_res); out_res may be NULL (D from the bf16 out alone).  Dh = 64 and N <= 336: ONE launch, one workgroup per
 * (batch, head) pair, dQ reduced inside the workgroup in a fixed order (round 6); otherwise two launches (dQ, then dK / dV).  Either
 * way bit-identical from run to run: no atomics. */
MVIT_API int mvit_attention_bwd(const void* qkv, const void* out, const void* out_res, const void* d_out, const float* lse,
                                float* dsum, void* dqkv, int B, int N, int H, int Dh, float scale, mvit_stream_t stream);

/* ---------------------------------------------------------------- decoder data movement (NHWC bf16) */
/* dst[b,oy,ox,c] = sum_{ty,tx} ty_w[oy,ty]*tx_w[ox,tx] * f(src[b, ty_idx[oy,ty], tx_idx[ox,tx], c]),
 * f = identity or relu(v*scale[c]+shift[c]) (BatchNorm+ReLU of the producer fused into the gather).
 * With host-built tap tables this is F.interpolate(bilinear x2, align_corners=False) of Fusion_Block
 * (src/generators/mipheivit.py:89), the bicubic regrid of Encoder.forward (mipheivit.py:147-151,161-162)
 * and the adjoint (backward) of either.  C, ld_src, ld_dst multiples of 8; *_bstride in elements. */
MVIT_API int mvit_resample2d(const void* src, void* dst, const int* ty_idx, const float* ty_w, const int* tx_idx,
                             const float* tx_w, const float* scale, const float* shift, int B, int h, int w, int H, int W,
                             int C, int ld_src, int ld_dst, long long src_bstride, long long dst_bstride, int T,
                             mvit_stream_t stream);
/* The x2 bilinear case of the above (F.interpolate(scale_factor=2, mode="bilinear", align_corners=False), Fusion_Block.forward,
 * src/generators/mipheivit.py:89) without tap tables: dst[b, 0:2h, 0:2w, 0:C] from src[b, 0:h, 0:w, 0:C], the producer's
 * BatchNorm+ReLU (scale/shift, both or neither) applied to every source value.  extra8 (optional): NHWC bf16 [B, 2h, 2w, 8],
 * copied into channels [C, C+8) of dst - the image slice D0 of the last concat buffer (Detail_Capture.forward,
 * mipheivit.py:208-211).  C, ld_src, ld_dst multiples of 8; *_bstride in elements. */
MVIT_API int mvit_upsample2x_bilinear(const void* src, void* dst, const float* scale, const float* shift, const void* extra8,
                                      int B, int h, int w, int C, int ld_src, int ld_dst, long long src_bstride,
                                      long long dst_bstride, mvit_stream_t stream);
/* Backward of mvit_upsample2x_bilinear w.r.t. its (post-activation) input: d_in[b, 0:h, 0:w, 0:C] (bf16) from
 * d_out[b, 0:2h, 0:2w, 0:C] (bf16, a channel slice of the concat gradient) - the adjoint of the interpolation matrix. */
MVIT_API int mvit_upsample2x_bilinear_bwd(const void* d_out, void* d_in, int B, int h, int w, int C, int ld_dout, int ld_din,
                                          long long dout_bstride, long long din_bstride, mvit_stream_t stream);
/* NCHW f32 image -> channels [0,C) of an NHWC bf16 buffer with pixel stride ld_dst; nzero trailing channels cleared
 * (D0 skip of Detail_Capture.forward, mipheivit.py:208-211, and the ConvStream input). */
MVIT_API int mvit_image_to_nhwc(const float* img, void* dst, int B, int S, int C, int ld_dst, int nzero,
                                mvit_stream_t stream);
/* nn.BatchNorm2d (eps 1e-5, momentum 0.1; mipheivit.py:33) from the conv epilogue's [nslots][2][C] f64 sums:
 * scale = gamma*rstd, shift = beta - mean*scale; training!=0 uses batch statistics and updates the running ones. */
MVIT_API int mvit_bn_finalize(const double* stats, const float* gamma, const float* beta, float* running_mean,
                              float* running_var, float* scale, float* shift, float* mean_out, float* rstd_out, int C,
                              int nslots, double count, float eps, float momentum, int training, mvit_stream_t stream);
/* out = dropout(relu(x*scale + shift)).  drop_p in [0,1): nn.Dropout(p) of Conv2DBlock / Deconv2DBlock in train mode
 * (src/generators/unet.py:441-519; 0 = off, the MIPHEI-ViT decoder has none): element e of the [M, C] activation is kept iff the
 * 16-bit field (e & 3) of splitmix64(drop_seed + (e >> 2) * 0x9E3779B97F4A7C15) is >= round(drop_p * 65536), kept values are
 * scaled by 1 / (1 - round(drop_p*65536)/65536).  The backward entry points recompute the same mask from (drop_seed, e). */
MVIT_API int mvit_bn_relu_apply(const void* x, const float* scale, const float* shift, void* out, long long M, int C,
                                int ld_x, int ld_out, float drop_p, unsigned long long drop_seed, mvit_stream_t stream);
/* backward of dropout(relu(BN(x))): reduce (sum g, sum g*xhat into [nslots][2][C] f64) then apply (dx, dgamma+=, dbeta+=). */
MVIT_API int mvit_bn_relu_bwd_reduce(const void* dy, int ld_dy, const void* x, const float* scale, const float* shift,
                                     const float* mean, const float* rstd, double* stats, long long M, int C, int nslots,
                                     float drop_p, unsigned long long drop_seed, mvit_stream_t stream);
MVIT_API int mvit_bn_relu_bwd_apply(const void* dy, int ld_dy, const void* x, const float* scale, const float* shift,
                                    const float* mean, const float* rstd, const float* gamma, const double* stats,
                                    float* dgamma, float* dbeta, void* dx, long long M, int C, int nslots, double count,
                                    float drop_p, unsigned long long drop_seed, mvit_stream_t stream);
/* dst[c][r] = src[r][c] (bf16). */
/* nn.Conv2d weight [Cout,Cin,3,3] f32 -> the packed bf16 operands of the implicit-GEMM convolution (mipheivit.py:20-31):
 * wk [Cout, 9*Cp] (forward), wd [Cp, 9*Cout] (dgrad, may be NULL); packed channel c = source channel (c+rot) mod Cin */
MVIT_API int mvit_pack_conv3x3_weights(const float* W, void* wk, void* wd, int Cout, int Cin, int Cp, int rot,
                                       mvit_stream_t stream);
/* The same for up to 8 weights in one launch (the decoder's seven convolutions are re-packed every training step). */
typedef struct mvit_conv_pack_desc {
  const float* W; void* wk; void* wd;   /* wd may be NULL */
  int Cout, Cin, Cp, rot;
} mvit_conv_pack_desc;
MVIT_API int mvit_pack_conv3x3_weights_multi(const mvit_conv_pack_desc* descs, int n, mvit_stream_t stream);
/* ConvTranspose2d(k=2, s=2) output placement (src/generators/unet.py:304-372,490-498): the GEMM result packed[M, 4*C]
 * (column (dy*2+dx)*C + c) <-> NHWC image [B, 2H, 2W] with row stride ld_img (a channel slice of a concat buffer).
 * inverse = 0: packed -> image; 1: image gradient -> packed. */
MVIT_API int mvit_pixel_shuffle2x(void* packed, void* img, int B, int H, int W, int C, long long ld_img, int inverse,
                                  mvit_stream_t stream);
MVIT_API int mvit_transpose_bf16(const void* src, void* dst, int R, int Cc, int ld_src, long long ld_dst,
                                 mvit_stream_t stream);

/* ---------------------------------------------------------------- fused output heads (<=16 SegmentationHeads) */
/* x = fusion output [M,32] bf16.  Stacked parameters: W1[NH,16,32], b1/gamma/beta/running_*[NH*16], W2[NH,16],
 * b2[NH], W3[NH,9,32] ((ky,kx) major), b3[NH].   src/generators/unet.py:407-438, mipheivit.py:198-218. */
MVIT_API int mvit_heads_moments(const void* x, double* mom /*[nslots][32+32*32]*/, long long M, int nslots,
                                mvit_stream_t stream);
MVIT_API int mvit_heads_bn_from_moments(const double* mom, const float* W1, const float* b1, const float* gamma,
                                        const float* beta, float* running_mean, float* running_var, float* scale,
                                        float* shift, float* mean_out, float* rstd_out, double* mom_sum, int NH, int nslots,
                                        double count, float eps, float momentum, int training, mvit_stream_t stream);
MVIT_API int mvit_heads_gate_fwd(const void* x, const float* W1, const float* b1, const float* scale, const float* shift,
                                 const float* W2, const float* b2, void* G /*bf16 [M,16]*/, long long M, int NH,
                                 mvit_stream_t stream);
MVIT_API int mvit_heads_conv_fwd(const void* x, const void* G, const float* W3, const float* b3, float* out /*NCHW f32*/,
                                 int B, int H, int W, int NH, mvit_stream_t stream);
/* scratch: bf16 dz map [M,16] + per-block partial sums of dW3 */
MVIT_API long long mvit_heads_conv_bwd_scratch_bytes(long long M);
MVIT_API int mvit_heads_conv_bwd(const float* dY, const float* Y, const void* x, const void* G, const float* W3,
                                 void* scratch, long long scratch_bytes, float* dG /*[M,16]*/, float* dXc /*[M,32]*/,
                                 float* dW3 /*[NH*9][32], overwritten*/,
                                 float* db3 /*[64 slots][32] partial sums, zeroed, summed by the caller*/, int B,
                                 int H, int W, int NH, mvit_stream_t stream);
/* scratch for mvit_heads_gate_bwd: per-block partial sums of the reduction pass + the apply-pass coefficients */
MVIT_API long long mvit_heads_gate_bwd_scratch_bytes(void);
MVIT_API int mvit_heads_gate_bwd(const void* x, const void* G, const float* dG, const float* dXc, const float* W1,
                                 const float* b1, const float* scale, const float* shift, const float* mean,
                                 const float* rstd, const float* gamma, const float* W2, const double* mom_sum,
                                 void* scratch, long long scratch_bytes, float* dW1, float* dgamma, float* dbeta,
                                 float* dW2, float* db2, void* dF /*bf16 [M,32]*/, long long M, int NH, double count,
                                 mvit_stream_t stream);

/* ---------------------------------------------------------------- per-step image metrics */
/* torchmetrics 1.6.2 PeakSignalNoiseRatio / StructuralSimilarityIndexMeasure state updates with data_range=(lo, hi), as
 * called at src/models.py:140-143 (train) and in evaluation_step: pred / target f32 NCHW;
 * state[0] += sum (clamp(p)-clamp(t))^2, state[1] += numel, state[2] += sum_b SSIM_b (11x11 Gaussian window, sigma 1.5,
 * windows fully inside the image), state[3] += B.  scratch: slotted partial sums (mvit_pix_metrics_scratch_bytes(B) bytes),
 * zero before the first call, left zero by every call. */
MVIT_API long long mvit_pix_metrics_scratch_bytes(int B);
MVIT_API int mvit_pix_metrics_update(const float* pred, const float* target, double* state /*[4]*/, double* scratch,
                                     long long scratch_bytes, int B, int C, int H, int W, float lo, float hi,
                                     mvit_stream_t stream);

/* ---------------------------------------------------------------- direct 3x3 convolution with LDS-staged input tiles */
/* Y[b,y,x,0..Cout) = sum_{ky,kx,c} X[b, y+ky-1, x+kx-1, c] * Wp[(ky,kx)][n][c]   (stride 1, zero padding 1, NHWC bf16, f32 accumulate)
 * for the few-channel full-resolution layers: the conv of the last Fusion_Block (src/generators/mipheivit.py:76-93, 67 -> 32
 * channels at 256 x 256) and, on weights packed with mode 1, its input gradient.  A persistent block stages (8+2) x (32+2) pixel
 * halo tiles in LDS by DMA (double-buffered) and reads all nine taps from there.  stats (nullable): [nslots][2][Cout] f64 sums of
 * the f32 results and of their squares (BatchNorm statistics, as MVIT_EPI_STATS).
 * Cin_pad = channels read per pixel (multiple of 8; pixel stride ldx >= Cin_pad), supported (Cin_pad, Cout): see
 * mvit_conv3x3_direct_supported. */
MVIT_API int mvit_conv3x3_direct_supported(int Cin_pad, int Cout);
MVIT_API int mvit_conv3x3_direct(const void* X, const void* Wp, void* Y, double* stats, int nslots, int B, int H, int W, int Cin_pad,
                                 int ldx, int Cout, int ldy, mvit_stream_t stream);
/* Weight gradient of the same convolution, dWn[n][(ky,kx,c)] (f32, [Cout][9*Cin_pad], accumulated into with atomics; unpack with
 * mvit_unpack_conv3x3_wgrad(n_major = 1)) = sum over pixels of X[pixel+(ky-1,kx-1)][c] * dY[pixel][n]: the parameter side of
 * loss.backward() for the last Fusion_Block conv (src/models.py:135).  X halo tiles and dY tiles are staged in LDS once; the
 * contraction over pixels uses transposing LDS reads.  (Cin_pad, Cout) in {(72,32), (32,32), (8,32)}. */
MVIT_API int mvit_conv3x3_direct_wgrad(const void* X, const void* dY, float* dWn, int B, int H, int W, int Cin_pad, int ldx,
                                       int Cout, int ldy, mvit_stream_t stream);
/* nn.Conv2d weight [Cout,Cin,3,3] f32 -> Wp [9][n_out][ceil16(k_pad) + 8] bf16 (zero padded).
 * mode 0 (forward, k_in = Cin, n_out >= Cout):  Wp[tap][n][c]   = W[n][(c+rot) % Cin][ky][kx]
 * mode 1 (input gradient, k_in = Cout, n_out <= Cin input channels wanted):  Wp[tap][ci][co] = W[co][(ci+rot) % Cin][2-ky][2-kx] */
MVIT_API int mvit_pack_conv3x3_direct(const float* W, void* out, int Cout, int Cin, int n_out, int k_in, int k_pad, int rot,
                                      int mode, mvit_stream_t stream);

/* Direct 3x3 convolution (stride 1, pad 1, NHWC bf16) for the WIDE decoder layers -- Fusion_Block convs 1728 -> 256 @ 32^2,
 * 352 -> 128 @ 64^2, 176 -> 64 @ 128^2 (src/generators/mipheivit.py:76-93) and their input gradients -- with LDS-staged input
 * tiles: a work item is (8 x 32-pixel tile, 64-channel output slice), a step one 32-channel chunk of the input; the chunk's halo
 * tile and its nine [64 x 32] weight taps are DMA'd into a two-slot LDS ring and all nine taps read the tile from there.
 *   mvit_conv3x3_chunked_pack: nn.Conv2d weight [Cout,Cin,3,3] f32 -> the kernel's operand; mode 0 forward (n = output channel,
 *     k = input channel), mode 1 input gradient (n = input channel of the forward conv, k = its output channel, taps flipped);
 *     mvit_conv3x3_chunked_pack_elems(N, K) = bf16 elements of the packed operand.
 *   mvit_conv3x3_chunked: Y[b,y,x,:Cout] = conv(X[b,y,x,:Cin]); stats (nullable): [nslots][2][Cout] f64 sum / sum of squares of
 *     the f32 results (BatchNorm2d batch statistics), one slot per block when nslots >= the CU count.  Cin, Cout % 8 == 0. */
MVIT_API long long mvit_conv3x3_chunked_pack_elems(int N, int K);
MVIT_API int mvit_conv3x3_chunked_pack(const float* W, void* out_bf16, int Cout, int Cin, int mode, mvit_stream_t stream);
/* several such packs in one launch (the per-step operand packs of the Fusion_Block convolutions, src/generators/mipheivit.py:76-93: their
 * weights change with every optimiser step): descs is a HOST array of n <= MVIT_CC_PACK_MAX descriptors, out buffers sized by
 * mvit_conv3x3_chunked_pack_elems */
#define MVIT_CC_PACK_MAX 8
typedef struct mvit_cc_pack_desc {
  const float* W; void* out;
  int Cout, Cin, mode, pad_;
} mvit_cc_pack_desc;
MVIT_API int mvit_conv3x3_chunked_pack_multi(const mvit_cc_pack_desc* descs, int n, mvit_stream_t stream);
MVIT_API int mvit_conv3x3_chunked(const void* X, const void* Wp, void* Y, double* stats, int nslots, int B, int H, int W, int Cin,
                                  int ldx, int Cout, int ldy, mvit_stream_t stream);
/* Weight gradient of the same layers on the same staging: dWn(f32)[Cout][9 * Cin_pad] += sum over pixels of
 * dY[pixel][n] * X[pixel + (ky-1, kx-1)][c] (output-channel major, k = tap * Cin_pad + c: the layout mvit_unpack_conv3x3_wgrad
 * takes with n_major = 1).  A block owns one (64-channel output slice, 32-channel input chunk) pair and walks a group of tiles with
 * the nine [32 x 32] tap blocks of each wave in registers; one tile group per pair when there are >= CU-count pairs (single writer
 * per element), otherwise groups meet in f32 atomics (the host's deterministic mode uses mvit_gemm_tn_bf16 instead). */
MVIT_API int mvit_conv3x3_chunked_wgrad(const void* X, const void* dY, float* dWn, int B, int H, int W, int Cin, int Cin_pad, int ldx,
                                        int Cout, int ldy, mvit_stream_t stream);

/* ---------------------------------------------------------------- per-step operand packs of the trainable tensors */
/* LoRA adapters (src/generators/lora.py:8-33) from the flat f32 parameter region `lora` = L x [Aq [D,r] | Bq [r,D] | Av | Bv]
 * to the bf16 operands of the kernels, R2 = 2r: AcatT [L,R2,D], Acat [L,D,R2] (may be NULL), B2 [L,3D,R2] (alpha*B on the q
 * rows / v rows, zero elsewhere: the K-extension of the qkv GEMM), Bqv [L,2,r,D] = alpha*Bq, alpha*Bv (may be NULL). */
MVIT_API int mvit_lora_pack(const float* lora, void* AcatT, void* Acat, void* B2, void* Bqv, int L, int D, int r, float alpha,
                            mvit_stream_t stream);
/* Weight gradient of a 3x3 convolution from the TN-GEMM layout dWt [(ky,kx,c_pad), Cout] f32 (n_major != 0: from the direct
 * kernel's [Cout, (ky,kx,c_pad)]) to nn.Conv2d's [Cout,Cin,3,3] (inverse of mvit_pack_conv3x3_weights incl. its channel
 * rotation); accumulate != 0 adds to dW. */
MVIT_API int mvit_unpack_conv3x3_wgrad(const float* dWt, float* dW, int Cout, int Cin, int Cp, int rot, int accumulate,
                                       int n_major, mvit_stream_t stream);
/* The same for up to 8 gradients in one launch. */
typedef struct mvit_conv_unpack_desc {
  const float* dWt; float* dW;
  int Cout, Cin, Cp, rot, accumulate, n_major;
} mvit_conv_unpack_desc;
MVIT_API int mvit_unpack_conv3x3_wgrad_multi(const mvit_conv_unpack_desc* descs, int n, mvit_stream_t stream);

/* ---------------------------------------------------------------- loss / optimiser */
/* WeightedMSELoss (src/loss.py:47-57): loss_acc += sum_c w_c sum (p-t)^2 (caller multiplies by lambda/(C*B*HW));
 * dY = 2*lambda/(C*B*HW) * w_c * (p-t).  pred/target/dY are NCHW f32. */
MVIT_API int mvit_wmse_fwd_bwd(const float* pred, const float* target, const float* w, double* loss_acc, float* dY, int B,
                               int C, long long HW, float lambda_factor, mvit_stream_t stream);
MVIT_API int mvit_sqnorm(const float* x, double* out, long long n, mvit_stream_t stream);
/* the same sum with the per-block partials (scratch: 256 doubles) added in block order by one thread instead of f64 atomics:
 * run-to-run identical (the host's MIPHEI_DETERMINISTIC mode; torch.nn.utils.clip_grad_norm_ of src/models.py:136) */
MVIT_API int mvit_sqnorm_ordered(const float* x, double* out, double* scratch256, long long n, mvit_stream_t stream);
/* clip_grad_norm_(max_norm) + torch.optim.Adam step on flat f32 buffers (src/models.py:136-138,359-371);
 * sqnorm = device scalar holding sum g^2 (no host sync), bias_c{1,2} = 1-beta^t.
 * NaN guard (src/models.py:102-105, `isnan(fake)` -> save weights -> raise): when *sqnorm is NaN/Inf, or *nonfinite_flag is
 * already set, the launch changes nothing and sets *nonfinite_flag = 1 (device int, sticky, may be NULL); the host reads the
 * flag asynchronously and finds the last finite weights in p. */
MVIT_API int mvit_adam_clip_step(float* p, const float* g, float* m, float* v, const double* sqnorm, long long n, float lr,
                                 float beta1, float beta2, float eps, float bias_c1, float bias_c2, float max_norm,
                                 int* nonfinite_flag, mvit_stream_t stream);

/* ---------------------------------------------------------------- on-device input / output stage */
/* dst(f32 NCHW)[b,c,p] = src(u8 NHWC)[b,p,c] * scale[c] + shift[c].  With scale = 1/std, shift = -mean/std this is
 * NormalizationLayer("he") of src/dataset.py:545-575 (H-Optimus-0 mean/std, dataset.py:599-601) after the HWC->CHW of
 * ToTensor; with scale = 1.8/255, shift = -0.9 the target transform (dataset.py:573).  HW % 4 == 0. */
MVIT_API int mvit_u8_nhwc_to_f32_nchw(const void* src_u8, float* dst, const float* scale, const float* shift, int B, int C,
                                      long long HW, mvit_stream_t stream);
/* Training-time spatial augmentation + normalisation of a batch of uint8 tiles, jointly for the H&E image [B,Hs,Ws,3] and the mIF
 * target [B,Hs,Ws,C] (either may be NULL): RandomCrop(H, W) -> HorizontalFlip(p_hflip) -> VerticalFlip(p_vflip) ->
 * CoarseDropout(p_drop, one hole, height / width uniform integers in [0, hole_frac * size], fill 0), the Compose of
 * src/dataset.py:458-468 with additional_targets={'image_target': 'image'} (same draw for both).  Outputs: out_img f32 NCHW
 * (px - mean) / std (NormalizationLayer "he", dataset.py:570), out_tgt f32 NCHW px / 255 * 1.8 - 0.9 (dataset.py:573) with the
 * reference's f32 operation order (bit-identical to the numpy expressions), out_nhwc8 (optional) bf16 [B,H,W,8] = the engine's
 * decoder image buffer (channels 3..7 zero).  Draws are counter-based: splitmix64(seed + golden * (16 * (sample0 + b) + k + 1)),
 * k = 0..8 -> crop y, crop x, hflip, vflip, dropout, hole h, hole w, hole y, hole x; mvit_augment_draw recomputes them on the
 * host.  W % 4 == 0. */
MVIT_API int mvit_augment_tiles(const void* img_u8, const void* tgt_u8, float* out_img, float* out_tgt, void* out_nhwc8, int B,
                                int C, int Hs, int Ws, int H, int W, unsigned long long seed, unsigned long long sample0,
                                float p_hflip, float p_vflip, float p_drop, float hole_frac, const float* mean3,
                                const float* std3, mvit_stream_t stream);
MVIT_API int mvit_augment_draw(int Hs, int Ws, int H, int W, unsigned long long seed, unsigned long long sample, float p_hflip,
                               float p_vflip, float p_drop, float hole_frac, int* out9);
/* dst(u8)[i] = trunc(clamp((src[i]+0.9)/1.8, 0, 1) * 255): SavePredictionsCallback, src/callbacks.py:345-346.  n % 4 == 0. */
MVIT_API int mvit_f32_to_u8_export(const float* src, void* dst_u8, long long n, mvit_stream_t stream);

/* ---------------------------------------------------------------- validation-time cell extractor */
/* Per-nucleus statistics of a batch as a segmented reduction over the label map (ids are slide-global: nothing is sized by the
 * label value).  Replaces MeanCellExtrator.forward / extract_mean (src/utils.py:23-121: F.interpolate 'area' of pred / target and
 * 'nearest-exact' of the label map when scale_factor < 1, then per image torch.unique + scatter_add_ + divide) and the reduction of
 * CellMetrics.update (src/metrics.py:38-74; want_sums != 0 returns sums instead of means).
 *   pred / target: f32 [B,C,H,W] (target may be NULL), nuclei: int32 or int64 [B,H,W], C <= 32, 0 < scale_factor <= 1.
 *   scratch: rec_count [B] int, rec_key [B,rmax] int, rec_val [B,rmax,2C+1] f32 (rmax a power of two <= 16384: partial records
 *   per image, one per (tile, nucleus) pair).
 *   outputs, per image b: n_unique[b]; rows [b, 0 .. n_unique[b]) of out_ids (ascending), out_count, out_pred [.,C], out_target.
 * A host that finds rec_count[b] > rmax must treat the result as invalid (more nucleus fragments than the scratch holds) and retry
 * with a larger rmax or on row chunks of the images; int64 labels above INT32_MAX are not representable (the host checks). */
MVIT_API int mvit_cell_means(const float* pred, const float* target, const void* nuclei, int label_is_int64, int B, int C, int H,
                             int W, float scale_factor, int rmax, int want_sums, int* rec_count, int* rec_key, float* rec_val,
                             int* n_unique, int* out_ids, float* out_count, float* out_pred, float* out_target,
                             mvit_stream_t stream);

/* ---------------------------------------------------------------- multi-GPU pre-flight helper (measurement, csrc/standin.hip) */
/* Holds `blocks` workgroups (256 threads, 64 VGPRs per lane, no LDS, no memory traffic) on the chip for `usec` microseconds:
 * a stand-in for a collective's ring kernels on a one-GPU box.  Replaces nothing in the reference (single-device training,
 * /root/reference/src/train.py:205-207); used by bench.py --comm-standin to price the CU residency RCCL's kernels take away
 * from the persistent GEMM launches of the data-parallel step. */
MVIT_API int mvit_occupy_cus(int blocks, int usec, mvit_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif
