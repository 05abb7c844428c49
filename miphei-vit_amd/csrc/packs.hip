// Per-step re-packs of the trainable tensors between their parameter layout (flat f32 buffer, reference state-dict shapes) and
// the operand layouts of the MFMA kernels.  Tiny and HBM-trivial: what matters is that each is ONE launch instead of a dozen
// strided torch copies (a dependent kernel boundary costs ~1.5 us on this chip).
//   LoRA adapters   src/generators/lora.py:8-33 (A [D,r], B [r,D] per adapter, alpha folded into the B-side operands)
//   conv weights    nn.Conv2d [Cout,Cin,3,3] of Basic_Conv3x3 / Fusion_Block, src/generators/mipheivit.py:20-41,76-93
#include "common.hpp"
#include "../../include/miphei_hip.h"

namespace {

// flat LoRA region of one block: Aq [D,r] | Bq [r,D] | Av [D,r] | Bv [r,D]  (f32)
// outputs (bf16), R2 = 2r:
//   AcatT  [L, R2, D]   rows j<r: Aq[:,j], j>=r: Av[:,j-r]            (B operand of t = LN1(x) @ [Aq|Av])
//   Acat   [L, D, R2]   [Aq | Av]                                     (B2 operand of the dqkv GEMM: dh += dt @ Acat^T)
//   B2     [L, 3D, R2]  rows n<D: alpha*Bq[j,n] for j<r; rows n>=2D: alpha*Bv[j-r,n-2D] for j>=r; else 0  (qkv K-extension)
//   Bqv    [L, 2, r, D] alpha*Bq, alpha*Bv                            (B operands of dt = dq @ Bq^T, dv @ Bv^T)
__global__ __launch_bounds__(256) void lora_pack_kernel(const float* __restrict__ lora, bf16_t* __restrict__ AcatT,
                                                        bf16_t* __restrict__ Acat, bf16_t* __restrict__ B2, bf16_t* __restrict__ Bqv,
                                                        int D, int r, float alpha) {
  const int l = blockIdx.y, R2 = 2 * r;
  const size_t per = (size_t)4 * r * D;
  const float* Aq = lora + l * per;
  const float* Bq = Aq + (size_t)r * D;
  const float* Av = Bq + (size_t)r * D;
  const float* Bv = Av + (size_t)r * D;
  const int nA = R2 * D, nB2 = 3 * D * R2;
  const int total = nA + nA + nB2 + nA;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
    if (i < nA) {                       // AcatT[j, d]
      const int j = i / D, d = i - j * D;
      AcatT[(size_t)l * nA + i] = f2bf(j < r ? Aq[(size_t)d * r + j] : Av[(size_t)d * r + j - r]);
    } else if (i < 2 * nA) {            // Acat[d, j]
      const int k = i - nA, d = k / R2, j = k - d * R2;
      if (Acat) Acat[(size_t)l * nA + k] = f2bf(j < r ? Aq[(size_t)d * r + j] : Av[(size_t)d * r + j - r]);
    } else if (i < 2 * nA + nB2) {      // B2[n, j]
      const int k = i - 2 * nA, n = k / R2, j = k - n * R2;
      float v = 0.f;
      if (n < D && j < r) v = alpha * Bq[(size_t)j * D + n];
      else if (n >= 2 * D && j >= r) v = alpha * Bv[(size_t)(j - r) * D + n - 2 * D];
      B2[(size_t)l * nB2 + k] = f2bf(v);
    } else {                            // Bqv[a, j, d]
      const int k = i - 2 * nA - nB2;
      if (Bqv) Bqv[(size_t)l * nA + k] = f2bf(alpha * (k < r * D ? Bq[k] : Bv[k - r * D]));
    }
  }
}

// dWt [(ky,kx,c_pad), Cout] f32 (output of the TN weight-gradient GEMM; n_major: [Cout, (ky,kx,c_pad)], the direct kernel's)
// -> parameter layout dW [Cout, Cin, 3, 3];
// packed channel c is parameter channel (c + rot) mod Cin (see pack_conv_w_kernel); the Cp - Cin pad rows are dropped.
struct UnpackBatch { mvit_conv_unpack_desc d[8]; };   // (up to 8 gradients per launch, blockIdx.y = gradient)
// A block moves a [32 packed channels] x [32 output channels] patch for all 9 taps through LDS: the TN layout is read along co
// (128-byte runs), the parameter layout written along (ci, tap) (up to 1152-byte runs per output channel) - the element-wise
// gather it replaces read 4 bytes per 1-7 KB stride (52 us for the seven decoder weights, 19 MB each way).
__global__ __launch_bounds__(256) void unpack_conv_wgrad_kernel(const UnpackBatch ub) {
  const mvit_conv_unpack_desc& q = ub.d[blockIdx.y];
  const float* __restrict__ dWt = q.dWt;
  float* __restrict__ dW = q.dW;
  const int Cout = q.Cout, Cin = q.Cin, Cp = q.Cp, rot = q.rot % q.Cin;
  __shared__ float tile[9][33][33];   // (odd strides in both dimensions: the (tap, channel)-major read-out is conflict-free)
  const int ncb = (Cp + 31) / 32, nob = (Cout + 31) / 32;
  for (int blk = blockIdx.x; blk < ncb * nob; blk += gridDim.x) {
    const int c0 = (blk % ncb) * 32, o0 = (blk / ncb) * 32;
    __syncthreads();
    for (int e = threadIdx.x; e < 9 * 32 * 32; e += 256) {
      const int t = e / 1024, r = (e >> 5) & 31, f = e & 31;     // f: fast index of the source layout
      float v = 0.f;
      if (q.n_major) {        // [co][t][c]: c fastest
        const int co = o0 + r, c = c0 + f;
        if (co < Cout && c < Cp) v = dWt[(size_t)co * 9 * Cp + (size_t)t * Cp + c];
        tile[t][f][r] = v;    // tile[t][c][co]
      } else {                // [t][c][co]: co fastest
        const int c = c0 + r, co = o0 + f;
        if (c < Cp && co < Cout) v = dWt[((size_t)t * Cp + c) * Cout + co];
        tile[t][r][f] = v;
      }
    }
    __syncthreads();
    // destination: dW[(co * Cin + ci) * 9 + t], ci = (c + rot) mod Cin for packed channel c < Cin; (c, t) fastest
    for (int e = threadIdx.x; e < 32 * 32 * 9; e += 256) {
      const int co = o0 + e / 288, rem = e % 288, c = c0 + rem / 9, t = rem % 9;
      if (co < Cout && c < Cin) {
        const int ci = (c + rot) % Cin;
        float* p = dW + ((size_t)co * Cin + ci) * 9 + t;
        const float v = tile[t][c - c0][co - o0];
        *p = q.accumulate ? *p + v : v;
      }
    }
  }
}

}  // namespace

extern "C" {

MVIT_API int mvit_lora_pack(const float* lora, void* AcatT, void* Acat, void* B2, void* Bqv, int L, int D, int r, float alpha,
                            mvit_stream_t stream) {
  MVIT_CLEAR_ERROR();
  if (!lora || !AcatT || !B2 || L <= 0 || D <= 0 || r <= 0 || 2 * r > 16) return MVIT_EINVAL;
  const int total = 3 * 2 * r * D + 3 * D * 2 * r;
  hipLaunchKernelGGL(lora_pack_kernel, dim3((total + 255) / 256 > 64 ? 64 : (total + 255) / 256, L), dim3(256), 0,
                     (hipStream_t)stream, lora, (bf16_t*)AcatT, (bf16_t*)Acat, (bf16_t*)B2, (bf16_t*)Bqv, D, r, alpha);
  return MVIT_LAUNCH_CHECK();
}

MVIT_API int mvit_unpack_conv3x3_wgrad(const float* dWt, float* dW, int Cout, int Cin, int Cp, int rot, int accumulate,
                                       int n_major, mvit_stream_t stream) {
  MVIT_CLEAR_ERROR();
  mvit_conv_unpack_desc d{dWt, dW, Cout, Cin, Cp, rot, accumulate, n_major};
  return mvit_unpack_conv3x3_wgrad_multi(&d, 1, stream);
}

MVIT_API int mvit_unpack_conv3x3_wgrad_multi(const mvit_conv_unpack_desc* descs, int n, mvit_stream_t stream) {
  MVIT_CLEAR_ERROR();
  if (!descs || n <= 0 || n > 8) return MVIT_EINVAL;
  UnpackBatch ub{};
  int most = 0;
  for (int i = 0; i < n; ++i) {
    const mvit_conv_unpack_desc& q = descs[i];
    if (!q.dWt || !q.dW || q.Cout <= 0 || q.Cin <= 0 || q.Cp < q.Cin || q.rot < 0) return MVIT_EINVAL;
    ub.d[i] = q;
    most = most > q.Cout * q.Cin * 9 ? most : q.Cout * q.Cin * 9;
  }
  hipLaunchKernelGGL(unpack_conv_wgrad_kernel, dim3((most + 9215) / 9216 > 1024 ? 1024 : (most + 9215) / 9216, n), dim3(256), 0,
                     (hipStream_t)stream, ub);
  return MVIT_LAUNCH_CHECK();
}

}  // extern "C"
