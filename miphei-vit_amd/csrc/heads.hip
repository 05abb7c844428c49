// Fused per-marker output heads (16 x SegmentationHead on the shared 32-channel feature map).
//
// Reference: AttentionBlock / SegmentationHead, src/generators/unet.py:407-438, instantiated once per
// marker at src/generators/mipheivit.py:198-205 and run at :213-218 (9 op launches per head, the
// 32-channel map re-read 16 times).  Here the map is read once per stage:
//   moments   : sum x and sum x x^T over all pixels (MFMA on transposed tiles) -> the train-mode BatchNorm statistics
//               of every head's 1x1 conv output follow analytically (mean = w.mu + b, var = w^T Cov w)
//   gate      : g[p,h] = sigmoid(w2 . relu(BN(W1 x + b1)) + b2) for all heads: the 256 stacked gate channels of a
//               32-pixel tile are one MFMA block chain per wave (bf16 operands, fp32 accumulate)
//   conv      : y[p,h] = tanh(b3 + sum_{3x3} W3[h] . (x*g_h)), one thread per pixel, NCHW f32 output
// and the matching backward passes.  HBM traffic is the 32-channel map plus the 16-channel gate / output.
#include "common.hpp"
#include "../../include/miphei_hip.h"

namespace {

constexpr int XC = 32;   // feature channels
constexpr int HC = 16;   // hidden channels of the gate (F_int = 32 // 2)
constexpr int MAXH = 16; // heads
constexpr int NMOM = XC + XC * XC;
constexpr int DB3_SLOTS = 64; // db3 partial sums: [DB3_SLOTS][32] floats (one 128-byte line per slot)

__device__ __forceinline__ void load_x32(const bf16_t* __restrict__ p, float (&x)[XC]) {
#pragma unroll
  for (int v = 0; v < 4; ++v) {
    const uint4 t = ((const uint4*)p)[v];
    const uint32_t u[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      x[v * 8 + 2 * j] = lo16f(u[j]);
      x[v * 8 + 2 * j + 1] = hi16f(u[j]);
    }
  }
}
__device__ __forceinline__ void load_g16(const bf16_t* __restrict__ p, float (&g)[MAXH]) {
#pragma unroll
  for (int v = 0; v < 2; ++v) {
    const uint4 t = ((const uint4*)p)[v];
    const uint32_t u[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      g[v * 8 + 2 * j] = lo16f(u[j]);
      g[v * 8 + 2 * j + 1] = hi16f(u[j]);
    }
  }
}

// ------------------------------------------------------------------ MFMA building blocks of the gate kernels
// v_mfma_f32_32x32x16_bf16 operand layouts (lane l: l31 = l & 31, half = l >> 5):
//   A[32 x 16]: row l31, K = 8*half + e (e = 0..7);  B[16 x 32]: col l31, K = 8*half + e;
//   D[32 x 32]: col l31, row acc_row(j, half) for accumulator register j = 0..15.
// A D block is chained into the next MFMA as a K operand without leaving registers: registers j = 8*ks + e of a lane
// become its eight K slots of k-step ks, i.e. K index 8*half + e stands for row acc_row(8*ks + e, half); the other
// operand is built with the same permutation.
typedef short v4s __attribute__((ext_vector_type(4)));
union Frag {
  bf16x8 v;
  uint32_t u[4];
  uint4 q;
  v4s h[2];
};
constexpr int NBLK = MAXH * HC / 32;  // 32-channel blocks of the stacked gate channels (2 heads per block)

__device__ __forceinline__ int acc_row(int j, int half) { return (j & 3) + 8 * (j >> 2) + 4 * half; }
__device__ __forceinline__ bf16x8 frag8(const float* f) {
  Frag r;
#pragma unroll
  for (int j = 0; j < 4; ++j) r.u[j] = pack2bf(f[2 * j], f[2 * j + 1]);
  return r.v;
}
__device__ __forceinline__ bf16x8 frag_lo(uint32_t w0) {  // K slots 0,1 from one packed word, the rest zero
  Frag r;
  r.u[0] = w0, r.u[1] = 0u, r.u[2] = 0u, r.u[3] = 0u;
  return r.v;
}
// fp32 value as a (hi, lo) bf16 pair packed into one word: hi + lo carries ~16 mantissa bits through a bf16 MFMA
__device__ __forceinline__ uint32_t split_bf(float v) {
  const uint32_t hi = pack2bf(v, 0.f) & 0xffffu;
  const float rem = v - lo16f(hi);
  return hi | (pack2bf(rem, 0.f) << 16);
}
__device__ __forceinline__ f32x16 mfma(bf16x8 a, bf16x8 b, f32x16 c) { return mvit_mfma32(a, b, c, 0, 0, 0); }
__device__ __forceinline__ f32x16 zero16() {
  f32x16 z;
#pragma unroll
  for (int j = 0; j < 16; ++j) z[j] = 0.f;
  return z;
}
// transposed 32x32 bf16 tile (64-byte rows) as a K operand: slot e of k-step ks <-> tile row acc_row(8*ks+e, half), col l31
__device__ __forceinline__ bf16x8 tile_T_frag(const char* tile, int ks, int lane) {
  const int i = lane & 15, half = lane >> 5;
  const char* p = tile + (ks * 16 + 4 * half + (i >> 2)) * 64 + (16 * ((lane >> 4) & 1) + 4 * (i & 3)) * 2;
  Frag r;
  r.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s*)p);
  r.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s*)(p + 8 * 64));
  return r.v;
}
// the two 16-byte pieces of pixel row p this lane feeds as a K operand (channel = 16*ks + 8*half + e)
__device__ __forceinline__ void load_x_frags(const bf16_t* __restrict__ x, long long p, bool live, int half, Frag& x0, Frag& x1) {
  if (live) {
    const uint4* r = (const uint4*)(x + (size_t)p * XC);
    x0.q = r[half];
    x1.q = r[2 + half];
  } else {
    x0.q = make_uint4(0, 0, 0, 0);
    x1.q = make_uint4(0, 0, 0, 0);
  }
}
// (hi, lo) bf16 pair of operands for eight fp32 values: hi + lo keeps ~16 mantissa bits through the bf16 MFMA, which
// pins the relu mask sign(a) to the fp32 result
__device__ __forceinline__ void frag8_split(const float* f, bf16x8& hi, bf16x8& lo) {
  float h[8], l[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    h[e] = lo16f(pack2bf(f[e], 0.f));
    l[e] = f[e] - h[e];
  }
  hi = frag8(h);
  lo = frag8(l);
}
// stacked 1x1-conv weights with the BatchNorm scale folded in, as the operand indexed by (channel = lane, K = x channel)
// (hi part in registers, lo part in a lane-indexed LDS table written by wave 0)
__device__ __forceinline__ void load_w1s(const float* __restrict__ W1, const float* __restrict__ b1, const float* __restrict__ scale,
                                         const float* __restrict__ shift, int nch, int lane, bool write_lo,
                                         bf16x8 (&w)[NBLK][2], bf16x8 (*wlo)[64], float (&bias)[NBLK]) {
  const int l31 = lane & 31, half = lane >> 5;
  // Every load below is UNCONDITIONAL on a clamped channel and masked by a multiplication: with `ok ? W1[..] : 0` hipcc branches
  // around each load and waits for it (152 dependent L2 round trips per block in the gate kernel: most of its 95 us).
#pragma unroll
  for (int blk = 0; blk < NBLK; ++blk) {
    const int ch = blk * 32 + l31;
    const bool ok = ch < nch;
    const int chc = ok ? ch : nch - 1;
    const float okf = ok ? 1.f : 0.f;
    const float sc = scale[chc] * okf;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      float f[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) f[e] = W1[(size_t)chc * XC + ks * 16 + half * 8 + e] * sc;
      bf16x8 lo;
      frag8_split(f, w[blk][ks], lo);
      if (write_lo) wlo[2 * blk + ks][lane] = lo;
    }
    bias[blk] = (b1[chc] * sc + shift[chc]) * okf;
  }
}

// ------------------------------------------------------------------ moments: sum x (32), sum x x^T (32x32) on MFMA
__global__ __launch_bounds__(256) void moments_kernel(const bf16_t* __restrict__ x, double* __restrict__ mom, long long M,
                                                      int nslots) {
  __shared__ __attribute__((aligned(16))) char tiles[4][2048];
  __shared__ float red[NMOM];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l31 = lane & 31, half = lane >> 5;
  for (int e = threadIdx.x; e < NMOM; e += 256) red[e] = 0.f;
  __syncthreads();
  char* tile = tiles[wave];
  Frag ones;
  ones.u[0] = ones.u[1] = ones.u[2] = ones.u[3] = MVIT_ONE2;
  f32x16 acc = zero16(), s = zero16();
  const long long ntile = (M + 31) / 32;
  for (long long t = (long long)blockIdx.x * 4 + wave; t < ntile; t += (long long)gridDim.x * 4) {
    const long long p = t * 32 + l31;
    Frag x0, x1;
    load_x_frags(x, p, p < M, half, x0, x1);
    *(uint4*)(tile + l31 * 64 + half * 16) = x0.q;
    *(uint4*)(tile + l31 * 64 + 32 + half * 16) = x1.q;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const bf16x8 f0 = tile_T_frag(tile, 0, lane), f1 = tile_T_frag(tile, 1, lane);
    acc = mfma(f0, f0, acc);
    acc = mfma(f1, f1, acc);
    s = mfma(f0, ones.v, s);
    s = mfma(f1, ones.v, s);
    __builtin_amdgcn_wave_barrier();
  }
  // the four waves add their partial moments in wave order (no LDS atomics: the block's sum is independent of scheduling)
  for (int w = 0; w < 4; ++w) {
    if (wave == w) {
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        red[XC + acc_row(j, half) * XC + l31] += acc[j];
        if (l31 == 0) red[acc_row(j, half)] += s[j];
      }
    }
    __syncthreads();
  }
  double* o = mom + (size_t)(blockIdx.x % nslots) * NMOM;
  for (int e = threadIdx.x; e < NMOM; e += 256) atomicAdd(o + e, (double)red[e]);
}

// ------------------------------------------------------------------ BN statistics of every gate channel from the moments
__global__ __launch_bounds__(1024) void bn_from_moments_kernel(const double* __restrict__ mom, const float* __restrict__ W1,
                                                               const float* __restrict__ b1, const float* __restrict__ gamma,
                                                               const float* __restrict__ beta, float* __restrict__ rmean,
                                                               float* __restrict__ rvar, float* __restrict__ scale,
                                                               float* __restrict__ shift, float* __restrict__ mean_o,
                                                               float* __restrict__ rstd_o, double* __restrict__ mom_sum,
                                                               int NCH, int nslots, double count, float eps, float momentum,
                                                               int training) {
  __shared__ double ms[NMOM];
  __shared__ double cov[XC * XC];
  if (training) {
    for (int e = threadIdx.x; e < NMOM; e += 1024) {
      double s0 = 0., s1 = 0., s2 = 0., s3 = 0.;
      int k = 0;
      for (; k + 4 <= nslots; k += 4) {
        s0 += mom[(size_t)k * NMOM + e];
        s1 += mom[(size_t)(k + 1) * NMOM + e];
        s2 += mom[(size_t)(k + 2) * NMOM + e];
        s3 += mom[(size_t)(k + 3) * NMOM + e];
      }
      for (; k < nslots; ++k) s0 += mom[(size_t)k * NMOM + e];
      const double s = (s0 + s1) + (s2 + s3);
      ms[e] = s;
      if (mom_sum) mom_sum[e] = s;
    }
    __syncthreads();
    {
      const int j = threadIdx.x >> 5, k = threadIdx.x & 31;
      cov[threadIdx.x] = ms[XC + j * XC + k] / count - (ms[j] / count) * (ms[k] / count);
    }
    __syncthreads();
  }
  static_assert(MAXH * HC * 4 <= 1024, "four lanes per gate channel in one block");
  // four lanes per channel: lane q takes rows q, q + 4, ... of the quadratic form w^T cov w (all 1024 threads work)
  const int ch4 = threadIdx.x >> 2, q4 = threadIdx.x & 3;
  double mean = 0., var = 0.;
  if (training) {
    const bool ok = ch4 < NCH;
    const float* w = W1 + (size_t)(ok ? ch4 : 0) * XC;
    double wd[XC];
#pragma unroll
    for (int k = 0; k < XC; ++k) wd[k] = (double)w[k];
    double wm = 0., v = 0.;
#pragma unroll 1
    for (int j = q4; j < XC; j += 4) {
      wm += wd[j] * ms[j];
      double row = 0.;
#pragma unroll
      for (int k = 0; k < XC; ++k) row += wd[k] * cov[j * XC + k];
      v += wd[j] * row;
    }
    wm += __shfl_xor(wm, 1, 64), v += __shfl_xor(v, 1, 64);
    wm += __shfl_xor(wm, 2, 64), v += __shfl_xor(v, 2, 64);
    mean = wm / count + (ok ? b1[ch4] : 0.f);
    var = v < 0. ? 0. : v;
  }
  // (channel ch4 is finished by its lane 0; the statements below keep the one-thread-per-channel form)
  if (ch4 >= NCH || q4 != 0) return;
  const int chn = ch4;
  if (training) {
    const double unb = count > 1. ? var * count / (count - 1.) : var;
    rmean[chn] = (float)((1. - momentum) * rmean[chn] + momentum * mean);
    rvar[chn] = (float)((1. - momentum) * rvar[chn] + momentum * unb);
  } else {
    mean = rmean[chn];
    var = rvar[chn];
  }
  const double rstd = 1. / sqrt(var + (double)eps);
  const double sc = gamma[chn] * rstd;
  scale[chn] = (float)sc;
  shift[chn] = (float)(beta[chn] - mean * sc);
  if (mean_o) mean_o[chn] = (float)mean;
  if (rstd_o) rstd_o[chn] = (float)rstd;
}

// ------------------------------------------------------------------ gate forward
// per 32-pixel tile and wave: a[ch,px] = W1s x + bias (3 MFMAs per 32 channels, the bias rides on a ones operand),
// relu in registers, psi[h,px] = sum_ch w2[ch] relu(a) as a second MFMA on the chained block, sigmoid, 16-byte store.
__global__ __launch_bounds__(256, 2) void gate_fwd_kernel(const bf16_t* __restrict__ x, const float* __restrict__ W1,
                                                          const float* __restrict__ b1, const float* __restrict__ scale,
                                                          const float* __restrict__ shift, const float* __restrict__ W2,
                                                          const float* __restrict__ b2, bf16_t* __restrict__ G, long long M,
                                                          int NH) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l31 = lane & 31, half = lane >> 5;
  const int nch = NH * HC, nblk = (nch + 31) / 32;
  __shared__ bf16x8 w2s[NBLK * 2][64];  // psi operand, lane-indexed (read back as conflict-free 16-byte rows)
  __shared__ bf16x8 wAl[NBLK * 2][64];  // lo parts of W1s
  bf16x8 wA[NBLK][2];
  float biasf[NBLK];
  uint32_t wbias[NBLK];
  load_w1s(W1, b1, scale, shift, nch, lane, wave == 0, wA, wAl, biasf);
#pragma unroll
  for (int blk = 0; blk < NBLK; ++blk) wbias[blk] = half == 0 ? split_bf(biasf[blk]) : 0u;
  for (int f = wave; f < NBLK * 2; f += 4) {  // row l31 of the psi operand selects head f = 2*blk + ks
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int c = (f >> 1) * 32 + acc_row(8 * (f & 1) + e, half);
      v[e] = W2[c < nch ? c : nch - 1] * ((l31 == f && c < nch) ? 1.f : 0.f);   // (unconditional load, see load_w1s)
    }
    w2s[f][lane] = frag8(v);
  }
  __syncthreads();
  const bf16x8 ones = frag_lo(half == 0 ? MVIT_ONE2 : 0u);
  float b2r[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) b2r[j] = b2[min(acc_row(j, half), NH - 1)] * (acc_row(j, half) < NH ? 1.f : 0.f);

  const long long ntile = (M + 31) / 32;
  // (Measured on this loop and dropped: the four chains of a group of channel blocks issued round-robin with two psi accumulators --
  // 256 registers, 78 vs 72 us; the next tile's pixel rows requested one iteration ahead -- 73.8 vs 71.9 us.  A wave's tile is a
  // dependent chain of 56 MFMAs and 320 VALU operations, and at two waves per SIMD the kernel runs at that chain's latency.)
  for (long long t = (long long)blockIdx.x * 4 + wave; t < ntile; t += (long long)gridDim.x * 4) {
    const long long p = t * 32 + l31;
    const bool live = p < M;
    Frag x0, x1;
    load_x_frags(x, p, live, half, x0, x1);
    f32x16 psi = zero16();
#pragma unroll
    for (int blk = 0; blk < NBLK; ++blk) {
      if (blk < nblk) {
        f32x16 a = mfma(wA[blk][0], x0.v, zero16());
        a = mfma(wA[blk][1], x1.v, a);
        a = mfma(wAl[2 * blk][lane], x0.v, a);
        a = mfma(wAl[2 * blk + 1][lane], x1.v, a);
        a = mfma(frag_lo(wbias[blk]), ones, a);
        float r[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) r[j] = __int_as_float(max(__float_as_int(a[j]), 0));   // ReLU as ONE v_max_i32 (fmaxf is two v_max_f32: the kernel is VALU-bound)
        // (non-finite inputs: -0.0 and every negative float compare below integer 0 -> 0, as fmaxf; a NaN keeps its payload if its sign
        //  bit is clear and becomes 0 if it is set -- NaN handling in the gate is UNSPECIFIED, the step's NaN guard is the gradient norm
        //  in optim.hip, which any NaN upstream of the gate reaches through the encoder's shared feature map)
        psi = mfma(w2s[2 * blk][lane], frag8(r), psi);
        psi = mfma(w2s[2 * blk + 1][lane], frag8(r + 8), psi);
      }
    }
    // psi rows (heads) of this lane: acc_row(j, half), j < 8  ->  heads 4*half+{0..3} and 8+4*half+{0..3}
    float g[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) g[j] = acc_row(j, half) < NH ? sigmoidf_(psi[j] + b2r[j]) : 0.f;
    if (live) {
      bf16_t* o = G + (size_t)p * MAXH + 4 * half;
      *(uint2*)o = make_uint2(pack2bf(g[0], g[1]), pack2bf(g[2], g[3]));
      *(uint2*)(o + 8) = make_uint2(pack2bf(g[4], g[5]), pack2bf(g[6], g[7]));
    }
  }
}

// ------------------------------------------------------------------ gated 3x3 conv + tanh, on MFMA
// y[p,h] = tanh(b3[h] + sum_d g[q_d,h] * T[q_d,(d,h)]),  T[q,(d,h)] = W3[h,d,:] . x[q,:],  q_d = p + offset(d).
// A block walks 4 x 32 pixel tiles: T for the tile plus its one-pixel halo (204 pixels = 7 MFMA pixel tiles, five
// 32-row (d,h) blocks each, W3 as bf16 hi+lo pairs) is formed once per pixel instead of once per (pixel, head, tap);
// the gated products are parked in LDS in fp32 and every output pixel sums its nine shifted entries.  The x / gate
// rows of the next tile are loaded before the current one is computed.
constexpr int CF_TH = 4, CF_TW = 32;
constexpr int CF_RW = CF_TW + 2, CF_RH = CF_TH + 2, CF_NR = CF_RH * CF_RW, CF_NT = (CF_NR + 31) / 32;
constexpr int CF_PS = 592;  // bytes per halo pixel: 9 taps x 16 heads fp32 (576) + pad -> conflict-free 16-byte stores
constexpr int CF_LDS = CF_NT * 32 * CF_PS;
static_assert(CF_NT <= 8, "one MFMA pixel tile per wave");

struct CfIn {
  Frag x0, x1;
  uint4 g;
};
// tanh through one exp and one reciprocal (|err| ~ 1e-7 relative on (-1, 1); saturates cleanly for large |y|)
__device__ __forceinline__ float fast_tanh(float y) {
  const float e = __expf(2.f * fminf(fmaxf(y, -15.f), 15.f));
  return 1.f - 2.f * __builtin_amdgcn_rcpf(e + 1.f);   // (v_rcp_f32, 1 ulp; an IEEE division is ten VALU instructions)
}

__global__ __launch_bounds__(512) void conv_fwd_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ G,
                                                       const float* __restrict__ W3, const float* __restrict__ b3,
                                                       float* __restrict__ out, int B, int H, int W, int NH) {
  __shared__ bf16x8 W3r[10][64], W3l[10][64];  // rows (d,h) of block blk (row order of conv_bwd_kernel), K = x channel
  extern __shared__ __attribute__((aligned(16))) char Ps[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l31 = lane & 31, half = lane >> 5;
  for (int f = wave; f < 10; f += 8) {
    const int blk = f >> 1, ks = f & 1;
    const int j = (l31 & 3) + 4 * (l31 >> 3), hr = (l31 >> 2) & 1;
    const int d = 2 * blk + (j >> 3), h = 8 * hr + (j & 7);
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = (d < 9 && h < NH) ? W3[((size_t)h * 9 + d) * XC + ks * 16 + half * 8 + e] : 0.f;
    frag8_split(v, W3r[f][lane], W3l[f][lane]);
  }
  const int tiles_x = (W + CF_TW - 1) / CF_TW, tiles_y = (H + CF_TH - 1) / CF_TH;
  const int nblocks = B * tiles_y * tiles_x;
  const int pix = threadIdx.x & 127, hq = threadIdx.x >> 7;  // phase 2: four threads per output pixel, four heads each
  const int oy = pix / CF_TW, ox = pix - oy * CF_TW;
  float bias[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) bias[e] = (hq * 4 + e < NH) ? b3[hq * 4 + e] : 0.f;

  auto load_tile = [&](int bid, CfIn& in) __attribute__((always_inline)) {  // wave t owns MFMA pixel tile t of the halo region
    const int tx = bid % tiles_x, ty = (bid / tiles_x) % tiles_y, b = bid / (tiles_x * tiles_y);
    const int r = wave * 32 + l31;
    const int ry = r / CF_RW, rx = r - ry * CF_RW;
    const int iy = ty * CF_TH - 1 + ry, ix = tx * CF_TW - 1 + rx;
    const bool ok = bid < nblocks && wave < CF_NT && r < CF_NR && iy >= 0 && iy < H && ix >= 0 && ix < W;
    const long long p = ok ? ((long long)b * H + iy) * W + ix : 0;
    load_x_frags(x, p, ok, half, in.x0, in.x1);
    in.g = ok ? *(const uint4*)(G + (size_t)p * MAXH + half * 8) : make_uint4(0, 0, 0, 0);
  };
  // one block per CU (the fp32 product tile fills LDS), so memory latency is covered by depth, not by occupancy:
  // the rows of the next three tiles are in flight while the current one is computed
  CfIn cur, n1, n2, n3;
  load_tile(blockIdx.x, cur);
  load_tile(blockIdx.x + gridDim.x, n1);
  load_tile(blockIdx.x + 2 * gridDim.x, n2);
  __syncthreads();  // W3r / W3l
  for (int bid = blockIdx.x; bid < nblocks; bid += gridDim.x) {
    load_tile(bid + 3 * gridDim.x, n3);
    // ---- phase 1: gated products of the tile + halo
    if (wave < CF_NT) {
      const uint32_t gu[4] = {cur.g.x, cur.g.y, cur.g.z, cur.g.w};
      float g[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) g[e] = ((e & 1) ? hi16f(gu[e >> 1]) : lo16f(gu[e >> 1]));
      char* prow = Ps + (size_t)(wave * 32 + l31) * CF_PS + half * 32;
      int tl = lane;  // opaque per iteration: the operand-table reads stay inside the loop
      asm volatile("" : "+v"(tl));
#pragma unroll
      for (int blk = 0; blk < 5; ++blk) {
        f32x16 T = mfma(W3r[2 * blk][tl], cur.x0.v, zero16());
        T = mfma(W3r[2 * blk + 1][tl], cur.x1.v, T);
        T = mfma(W3l[2 * blk][tl], cur.x0.v, T);
        T = mfma(W3l[2 * blk + 1][tl], cur.x1.v, T);
#pragma unroll
        for (int dd = 0; dd < 2; ++dd) {
          const int d = 2 * blk + dd;
          if (d < 9) {
            *(float4*)(prow + d * 64) = make_float4(g[0] * T[8 * dd], g[1] * T[8 * dd + 1], g[2] * T[8 * dd + 2], g[3] * T[8 * dd + 3]);
            *(float4*)(prow + d * 64 + 16) =
                make_float4(g[4] * T[8 * dd + 4], g[5] * T[8 * dd + 5], g[6] * T[8 * dd + 6], g[7] * T[8 * dd + 7]);
          }
        }
      }
    }
    __syncthreads();
    // ---- phase 2: sum of the nine shifted entries, tanh, NCHW store
    {
      const int tx = bid % tiles_x, ty = (bid / tiles_x) % tiles_y, b = bid / (tiles_x * tiles_y);
      const int py = ty * CF_TH + oy, px = tx * CF_TW + ox;
      if (py < H && px < W) {
        float y0 = bias[0], y1 = bias[1], y2 = bias[2], y3 = bias[3];
#pragma unroll
        for (int d = 0; d < 9; ++d) {
          // (a 4-float VECTOR type: through HIP's float4 struct hipcc scalarised this read into ds_read2_b32 + 2 ds_read_b32, whose
          //  4-byte lanes at a 592-byte stride are 4-way bank conflicts -- 45 % of the kernel's LDS cycles in round 3's counters)
          typedef float f32x4_lds __attribute__((ext_vector_type(4)));
          const f32x4_lds a = *(const f32x4_lds*)(Ps + (size_t)((oy + d / 3) * CF_RW + ox + d % 3) * CF_PS + d * 64 + hq * 16);
          y0 += a.x, y1 += a.y, y2 += a.z, y3 += a.w;
        }
        const size_t plane = (size_t)H * W;
        float* o = out + ((size_t)b * NH + hq * 4) * plane + (size_t)py * W + px;
        if (hq * 4 < NH) o[0] = fast_tanh(y0);
        if (hq * 4 + 1 < NH) o[plane] = fast_tanh(y1);
        if (hq * 4 + 2 < NH) o[2 * plane] = fast_tanh(y2);
        if (hq * 4 + 3 < NH) o[3 * plane] = fast_tanh(y3);
      }
    }
    __syncthreads();  // the products are consumed: the next tile may overwrite them
    cur = n1, n1 = n2, n2 = n3;
  }
}

// ------------------------------------------------------------------ backward of the gated conv, on MFMA
// dz[p,h] = dY*(1-y^2) (pre-pass, bf16 NHWC).  Per pixel q and 3x3 offset d (p = q - d + 1):
//   E[q,(d,h)]  = g[q,h] * dz[p,h]
//   dXc[q,c]    = sum_(d,h) E[q,(d,h)] W3[h,d,c]                 (K = 144 chain of nine k-steps, one per offset)
//   T[q,(d,h)]  = sum_c x[q,c] W3[h,d,c];  dG[q,h] = sum_d dz[p,h] T[q,(d,h)]
//   dW3[h,d,c]  = sum_q E[q,(d,h)] x[q,c]                        (E and x tiles transposed through LDS)
// db3[h] = sum_p dz[p,h] comes out of the pre-pass.
constexpr int CB_BLOCKS = 512;              // persistent grid of the main pass = rows of the dW3 partial buffer
constexpr int CB_COLS = 160;                // 9*16 (d,h) columns padded to five 32-column blocks
constexpr int CB_PART = CB_COLS * XC;       // floats per partial row
constexpr int ET_STRIDE = 336;              // bytes per pixel row of the E tile (160 bf16 + pad: conflict-free 16-byte stores)

constexpr int DZ_BLOCKS = 2048;   // grid cap of conv_bwd_dz_kernel = rows of its per-block db3 partials
// FULL: NH == MAXH, no per-head bounds test around the loads (with it hipcc waits for each head's pair of loads before the next)
template <bool FULL>
__global__ __launch_bounds__(256) void conv_bwd_dz_kernel(const float* __restrict__ dY, const float* __restrict__ Y,
                                                          bf16_t* __restrict__ dzb, float* __restrict__ dbpart, int B,
                                                          long long HW, int NH) {
  __shared__ float dbw[4][MAXH];
  const long long M = (long long)B * HW;
  float db_acc[MAXH];
#pragma unroll
  for (int h = 0; h < MAXH; ++h) db_acc[h] = 0.f;
  for (long long p = (long long)blockIdx.x * 256 + threadIdx.x; p < M; p += (long long)gridDim.x * 256) {
    const long long b = p / HW, pix = p - b * HW;
    float dz[MAXH];
#pragma unroll
    for (int h = 0; h < MAXH; ++h) {
      dz[h] = 0.f;
      if (FULL || h < NH) {
        const size_t i = ((size_t)b * NH + h) * HW + pix;
        const float yy = Y[i];
        dz[h] = dY[i] * (1.f - yy * yy);
        db_acc[h] += dz[h];
      }
    }
    uint4 o[2];
    uint32_t* ou = (uint32_t*)o;
#pragma unroll
    for (int j = 0; j < 8; ++j) ou[j] = pack2bf(dz[2 * j], dz[2 * j + 1]);
    ((uint4*)(dzb + (size_t)p * MAXH))[0] = o[0];
    ((uint4*)(dzb + (size_t)p * MAXH))[1] = o[1];
  }
  // bias gradient: one partial row per block, the four waves added in wave order (no atomics: run-to-run identical); the rows
  // are summed in block order by conv_bwd_dw3_kernel
#pragma unroll
  for (int h = 0; h < MAXH; ++h) {
    const float s = wave_sum(db_acc[h]);
    if ((threadIdx.x & 63) == 0) dbw[threadIdx.x >> 6][h] = s;
  }
  __syncthreads();
  if (threadIdx.x < MAXH) dbpart[(size_t)blockIdx.x * MAXH + threadIdx.x] = ((dbw[0][threadIdx.x] + dbw[1][threadIdx.x]) + dbw[2][threadIdx.x]) + dbw[3][threadIdx.x];
}

// transposed K operand from a tile with `stride`-byte rows: slot e of k-step ks <-> tile row acc_row(8*ks+e, half), col l31
__device__ __forceinline__ bf16x8 tile_T_frag_s(const char* tile, int stride, int ks, int lane) {
  const int i = lane & 15, half = lane >> 5;
  const char* p = tile + (ks * 16 + 4 * half + (i >> 2)) * stride + (16 * ((lane >> 4) & 1) + 4 * (i & 3)) * 2;
  Frag r;
  r.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s*)p);
  r.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s*)(p + 8 * stride));
  return r.v;
}

__global__ __launch_bounds__(256, 1) void conv_bwd_kernel(const bf16_t* __restrict__ dzb, const bf16_t* __restrict__ x,
                                                          const bf16_t* __restrict__ G, const float* __restrict__ W3,
                                                          float* __restrict__ dG, float* __restrict__ dXc,
                                                          float* __restrict__ part, int B, int H, int W, int NH) {
  // lane-indexed operand tables: W3c[d] (rows = x channel, K = head) for dXc, W3r[blk][ks] (rows = (d,h), K = x channel) for T
  __shared__ bf16x8 W3c[9][64];
  __shared__ bf16x8 W3r[10][64];
  __shared__ bf16x8 W3rl[10][64];  // lo parts: T feeds the gate gradient, whose per-head sums cancel heavily
  __shared__ __attribute__((aligned(16))) char tiles[4][32 * ET_STRIDE + 2048];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l31 = lane & 31, half = lane >> 5;
  for (int f = wave; f < 19; f += 4) {
    float v[8];
    if (f < 9) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int h = half * 8 + e;
        v[e] = h < NH ? W3[((size_t)h * 9 + f) * XC + l31] : 0.f;
      }
      W3c[f][lane] = frag8(v);
    } else {
      // row rho = l31 of block blk holds (d, h) = (2*blk + (j >> 3), 8*hr + (j & 7)) with rho = acc_row(j, hr): the lane that
      // receives accumulator register j then owns heads 8*half .. 8*half+7, the ones its dz / gate loads cover
      const int blk = (f - 9) >> 1, ks = (f - 9) & 1;
      const int j = (l31 & 3) + 4 * (l31 >> 3), hr = (l31 >> 2) & 1;
      const int d = 2 * blk + (j >> 3), h = 8 * hr + (j & 7);
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = (d < 9 && h < NH) ? W3[((size_t)h * 9 + d) * XC + ks * 16 + half * 8 + e] : 0.f;
      frag8_split(v, W3r[f - 9][lane], W3rl[f - 9][lane]);
    }
  }
  __syncthreads();
  char* et = tiles[wave];
  char* xt = et + 32 * ET_STRIDE;
  // columns 144..159 of the E tile are never written: zero them once so the padded dW3 columns stay finite
  *(uint4*)(et + l31 * ET_STRIDE + 288 + half * 16) = make_uint4(0, 0, 0, 0);
  f32x16 dw[5];
#pragma unroll
  for (int c = 0; c < 5; ++c) dw[c] = zero16();

  const int M = B * H * W, ntile = (M + 31) / 32;  // the launcher bounds M to 31 bits: 32-bit pixel arithmetic
  for (int t = blockIdx.x * 4 + wave; t < ntile; t += gridDim.x * 4) {
    const int p = t * 32 + l31;
    const bool live = p < M;
    const int pp = live ? p : 0;
    const int qx = pp % W;
    const int qy = (pp / W) % H;
    Frag x0, x1;
    load_x_frags(x, p, live, half, x0, x1);
    float g[8];
    {
      uint4 gq = make_uint4(0, 0, 0, 0);
      if (live) gq = *(const uint4*)(G + (size_t)p * MAXH + half * 8);
      const uint32_t gu[4] = {gq.x, gq.y, gq.z, gq.w};
#pragma unroll
      for (int e = 0; e < 8; ++e) g[e] = ((e & 1) ? hi16f(gu[e >> 1]) : lo16f(gu[e >> 1]));
    }
    uint4 dzq[9];
#pragma unroll
    for (int d = 0; d < 9; ++d) {
      const int py = qy - d / 3 + 1, px = qx - d % 3 + 1;
      dzq[d] = make_uint4(0, 0, 0, 0);
      if (live && py >= 0 && py < H && px >= 0 && px < W)
        dzq[d] = *(const uint4*)(dzb + (size_t)(p + (1 - d / 3) * W + (1 - d % 3)) * MAXH + half * 8);
    }
    __builtin_amdgcn_wave_barrier();
    *(uint4*)(xt + l31 * 64 + half * 16) = x0.q;
    *(uint4*)(xt + l31 * 64 + 32 + half * 16) = x1.q;
    int tl = lane;  // opaque per iteration: keeps the operand-table reads inside the loop instead of 76 hoisted registers
    asm volatile("" : "+v"(tl));
    f32x16 dxc = zero16();
    float dg[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) dg[e] = 0.f;
#pragma unroll
    for (int blk = 0; blk < 5; ++blk) {
      f32x16 T = mfma(W3r[2 * blk][tl], x0.v, zero16());  // T^T[(d,h), q] for offsets d = 2*blk, 2*blk+1
      T = mfma(W3r[2 * blk + 1][tl], x1.v, T);
      T = mfma(W3rl[2 * blk][tl], x0.v, T);
      T = mfma(W3rl[2 * blk + 1][tl], x1.v, T);
#pragma unroll
      for (int dd = 0; dd < 2; ++dd) {
        const int d = 2 * blk + dd;
        if (d < 9) {
          const uint32_t du[4] = {dzq[d].x, dzq[d].y, dzq[d].z, dzq[d].w};
          float ev[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float dz = ((e & 1) ? hi16f(du[e >> 1]) : lo16f(du[e >> 1]));
            dg[e] += dz * T[8 * dd + e];
            ev[e] = g[e] * dz;
          }
          Frag E;
          E.v = frag8(ev);
          *(uint4*)(et + l31 * ET_STRIDE + d * 32 + half * 16) = E.q;
          dxc = mfma(W3c[d][tl], E.v, dxc);  // dXc^T[c, q]
        }
      }
    }
    if (live) {
      *(float4*)(dG + (size_t)p * MAXH + half * 8) = make_float4(dg[0], dg[1], dg[2], dg[3]);
      *(float4*)(dG + (size_t)p * MAXH + half * 8 + 4) = make_float4(dg[4], dg[5], dg[6], dg[7]);
#pragma unroll
      for (int q = 0; q < 4; ++q)
        *(float4*)(dXc + (size_t)p * XC + 8 * q + 4 * half) = make_float4(dxc[4 * q], dxc[4 * q + 1], dxc[4 * q + 2], dxc[4 * q + 3]);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const bf16x8 xT0 = tile_T_frag(xt, 0, lane), xT1 = tile_T_frag(xt, 1, lane);
#pragma unroll
    for (int c = 0; c < 5; ++c) {  // dW3^T[ch, (d,h)]: rows = x channel, cols = 32 (d,h) columns of block c
      dw[c] = mfma(xT0, tile_T_frag_s(et + c * 64, ET_STRIDE, 0, lane), dw[c]);
      dw[c] = mfma(xT1, tile_T_frag_s(et + c * 64, ET_STRIDE, 1, lane), dw[c]);
    }
  }
  // ---- block reduction, one partial row [160 (d,h) columns][32 channels] per block
  float* zb = (float*)&tiles[0][0];  // [5][16][64]
  __syncthreads();
  for (int w = 0; w < 4; ++w) {
    if (wave == w) {
#pragma unroll
      for (int c = 0; c < 5; ++c)
#pragma unroll
        for (int j = 0; j < 16; ++j) {
          float* q = zb + (c * 16 + j) * 64 + lane;
          *q = (w == 0 ? 0.f : *q) + dw[c][j];
        }
    }
    __syncthreads();
  }
  float* o = part + (size_t)blockIdx.x * CB_PART;
  for (int e = threadIdx.x; e < 5 * 16 * 64; e += 256) {
    const int ln = e & 63, j = (e >> 6) & 15, c = e >> 10;
    o[(c * 32 + (ln & 31)) * XC + acc_row(j, ln >> 5)] = zb[e];
  }
}

// dW3[(h*9+d)*32 + c] = sum over the partial rows of column d*16+h
__global__ __launch_bounds__(256) void conv_bwd_dw3_kernel(const float* __restrict__ part, float* __restrict__ dW3, int NH, int nrows,
                                                           const float* __restrict__ dbpart, int nbrows, float* __restrict__ db3) {
  const int o = blockIdx.x * 256 + threadIdx.x;
  if (blockIdx.x == gridDim.x - 1) {               // db3[h] (row 0 of the caller's [DB3_SLOTS][32] table): block partials in a fixed order
    // 16 row segments x 16 heads: every thread sums its segment front to back (independent loads, eight in flight), then the
    // segment sums are added in segment order -- the same association every run
    __shared__ double seg[16][MAXH];
    const int h = threadIdx.x & 15, sg = threadIdx.x >> 4;
    const int per = (nbrows + 15) / 16, b0 = sg * per, b1 = min(nbrows, b0 + per);
    double t = 0.;
    int b = b0;
    for (; b + 8 <= b1; b += 8) {
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = dbpart[(size_t)(b + e) * MAXH + h];
#pragma unroll
      for (int e = 0; e < 8; ++e) t += v[e];
    }
    for (; b < b1; ++b) t += dbpart[(size_t)b * MAXH + h];
    seg[sg][h] = t;
    __syncthreads();
    if (threadIdx.x < NH) {
      double a = 0.;
#pragma unroll
      for (int g = 0; g < 16; ++g) a += seg[g][threadIdx.x];
      db3[threadIdx.x] += (float)a;
    }
  }
  // eight lanes per output element: lane q sums rows [q, q+1) * nrows / 8 front to back (four independent accumulators), the eight
  // sums meet in a fixed xor tree -- the same association every run.  (One lane per element walked all 512 rows: 128 dependent
  // L2 round trips, 51 us on 18 blocks.)
  const int oe = o >> 3, q = o & 7;
  const bool live = oe < NH * 9 * XC;
  const int oc = live ? oe : 0;
  const int c = oc % XC, hd = oc / XC, h = hd / 9, d = hd % 9;
  const size_t col = (size_t)(d * 16 + h) * XC + c;
  const int per = nrows >> 3, b0 = q * per;          // (nrows = CB_BLOCKS, a multiple of 32)
  double s0 = 0., s1 = 0., s2 = 0., s3 = 0.;
  for (int b = b0; b < b0 + per; b += 4) {
    s0 += part[(size_t)b * CB_PART + col];
    s1 += part[(size_t)(b + 1) * CB_PART + col];
    s2 += part[(size_t)(b + 2) * CB_PART + col];
    s3 += part[(size_t)(b + 3) * CB_PART + col];
  }
  double t = (s0 + s1) + (s2 + s3);
  t += __shfl_xor(t, 1, 64);
  t += __shfl_xor(t, 2, 64);
  t += __shfl_xor(t, 4, 64);
  if (live && q == 0) dW3[oe] = (float)t;
}

// ------------------------------------------------------------------ gate backward
// With dr[p,ch] = relu'(a[p,ch]) * dpsi[p,head(ch)], every sum the BatchNorm/conv backward needs is linear in
//   Zr[ch,k] = sum_p dr[p,ch] x[p,k],   Sr[ch] = sum_p dr[p,ch],   db2[h] = sum_p dpsi[p,h]
// (u = W1 x + b1 is affine in x), so the reduction pass is two chained MFMAs per 32 channels and the per-channel
// algebra moves to a tiny finalize kernel.  The apply pass splits du[p,ch] = K1 dr + K2 u + K3 into the masked part
// (MFMA against K1*W1) and the affine part x Q + r with Q = W1^T diag(K2) W1 (one 32x32 matrix for all channels).
constexpr int RED_Z = MAXH * HC * XC;            // Zr
constexpr int RED_N = RED_Z + MAXH * HC + MAXH;  // + Sr + db2
constexpr int RED_BLOCKS = 256;                  // partial sums written by the reduction pass
// scratch layout (floats): part[RED_BLOCKS][RED_N] | coef[256][4] (K1, K2, K2*b1+K3, -) | Q[32][32] | r[32]
constexpr size_t SCR_COEF = (size_t)RED_BLOCKS * RED_N;
constexpr size_t SCR_Q = SCR_COEF + MAXH * HC * 4;
constexpr size_t SCR_R = SCR_Q + XC * XC;
constexpr size_t SCR_FLOATS = SCR_R + XC;

__global__ __launch_bounds__(512, 1) void gate_bwd_reduce_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ G,
                                                                 const float* __restrict__ dG, const float* __restrict__ W1,
                                                                 const float* __restrict__ b1, const float* __restrict__ scale,
                                                                 const float* __restrict__ shift, float* __restrict__ part,
                                                                 long long M, int NH) {
  // eight waves = four pixel streams x two channel groups (128 gate channels each, so the Zr accumulators stay at
  // 64 registers); per wave: x tile (2 KB) + dpsi^T tile [16 heads][32 px] f32 (2 KB); the same LDS is reused for
  // the cross-wave reduction at the end
  constexpr int WB = NBLK / 2;
  __shared__ __attribute__((aligned(16))) char sm[NBLK * 16 * 64 * 4 + NBLK * 64 * 4 + MAXH * 4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l31 = lane & 31, half = lane >> 5;
  const int cg = wave & 1, stream = wave >> 1;
  const int nch = NH * HC, nblk = (nch + 31) / 32;
  char* xt = sm + wave * 4096;
  float* dps = (float*)(xt + 2048);
  bf16x8 wB[WB][2], wBl[WB][2];
  float thr[WB];
#pragma unroll
  for (int b = 0; b < WB; ++b) {
    const int ch = (cg * WB + b) * 32 + l31;
    const bool ok = ch < nch;
    const int chc = ok ? ch : nch - 1;            // (unconditional loads on a clamped channel, see load_w1s)
    const float okf = ok ? 1.f : 0.f;
    const float sc = scale[chc] * okf;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      float f[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) f[e] = W1[(size_t)chc * XC + ks * 16 + half * 8 + e] * sc;
      frag8_split(f, wB[b][ks], wBl[b][ks]);
    }
    thr[b] = -(b1[chc] * sc + shift[chc]) * okf;  // a > 0  <=>  W1s x > -bias
  }
  f32x16 Z[WB];
  float sr[WB], dbh[8];
#pragma unroll
  for (int b = 0; b < WB; ++b) Z[b] = zero16(), sr[b] = 0.f;
#pragma unroll
  for (int e = 0; e < 8; ++e) dbh[e] = 0.f;

  const long long ntile = (M + 31) / 32;
  if (cg * WB < nblk) {
    for (long long t = (long long)blockIdx.x * 4 + stream; t < ntile; t += (long long)gridDim.x * 4) {
      const long long p = t * 32 + l31;
      const bool live = p < M;
      Frag x0, x1;
      load_x_frags(x, p, live, half, x0, x1);
      float dpsi[8];
      if (live) {
        const uint4 gq = *(const uint4*)(G + (size_t)p * MAXH + half * 8);
        const float4 d0 = *(const float4*)(dG + (size_t)p * MAXH + half * 8), d1 = *(const float4*)(dG + (size_t)p * MAXH + half * 8 + 4);
        const uint32_t gu[4] = {gq.x, gq.y, gq.z, gq.w};
        const float dg[8] = {d0.x, d0.y, d0.z, d0.w, d1.x, d1.y, d1.z, d1.w};
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float g = ((e & 1) ? hi16f(gu[e >> 1]) : lo16f(gu[e >> 1]));
          dpsi[e] = (half * 8 + e < NH) ? dg[e] * g * (1.f - g) : 0.f;
        }
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) dpsi[e] = 0.f;
      }
      __builtin_amdgcn_wave_barrier();
      *(uint4*)(xt + l31 * 64 + half * 16) = x0.q;
      *(uint4*)(xt + l31 * 64 + 32 + half * 16) = x1.q;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        dbh[e] += dpsi[e];
        dps[(half * 8 + e) * 32 + l31] = dpsi[e];
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      const bf16x8 xT0 = tile_T_frag(xt, 0, lane), xT1 = tile_T_frag(xt, 1, lane);
#pragma unroll
      for (int b = 0; b < WB; ++b) {
        const int blk = cg * WB + b;
        if (blk < nblk) {
          f32x16 a = mfma(x0.v, wB[b][0], zero16());  // a[px, ch]: col = channel l31 of the block, rows = pixels
          a = mfma(x1.v, wB[b][1], a);
          a = mfma(x0.v, wBl[b][0], a);
          a = mfma(x1.v, wBl[b][1], a);
          const float* dp = dps + (blk * 2 + (l31 >> 4)) * 32 + 4 * half;
          float dr[16];
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const float4 d = *(const float4*)(dp + 8 * q);
            dr[4 * q] = a[4 * q] > thr[b] ? d.x : 0.f;
            dr[4 * q + 1] = a[4 * q + 1] > thr[b] ? d.y : 0.f;
            dr[4 * q + 2] = a[4 * q + 2] > thr[b] ? d.z : 0.f;
            dr[4 * q + 3] = a[4 * q + 3] > thr[b] ? d.w : 0.f;
          }
          float s = 0.f;
#pragma unroll
          for (int j = 0; j < 16; ++j) s += dr[j];
          sr[b] += s;
          Z[b] = mfma(frag8(dr), xT0, Z[b]);  // Zr[ch, k]: rows = channels (chained), col = x channel l31
          Z[b] = mfma(frag8(dr + 8), xT1, Z[b]);
        }
      }
    }
  }
  // ---- block reduction of the waves' accumulators, then one partial row per block
  float* zb = (float*)sm;                     // [NBLK][16][64]
  float* sb = zb + NBLK * 16 * 64;            // [NBLK][64]
  float* db = sb + NBLK * 64;                 // [MAXH]
#pragma unroll
  for (int e = 0; e < 8; ++e) {
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) dbh[e] += __shfl_xor(dbh[e], o, 64);
  }
  __syncthreads();
  if (threadIdx.x < MAXH) db[threadIdx.x] = 0.f;
  for (int w = 0; w < 4; ++w) {
    if (stream == w) {
#pragma unroll
      for (int b = 0; b < WB; ++b) {
        const int blk = cg * WB + b;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
          float* q = zb + (blk * 16 + j) * 64 + lane;
          *q = (w == 0 ? 0.f : *q) + Z[b][j];
        }
        float* q = sb + blk * 64 + lane;
        *q = (w == 0 ? 0.f : *q) + sr[b];
      }
    }
    __syncthreads();
  }
  for (int w = 0; w < 4; ++w) {               // (wave order, no LDS atomics: see the accumulators above)
    if (stream == w && l31 == 0 && cg == 0) {
#pragma unroll
      for (int e = 0; e < 8; ++e) db[half * 8 + e] += dbh[e];
    }
    __syncthreads();
  }
  float* o = part + (size_t)blockIdx.x * RED_N;
  for (int e = threadIdx.x; e < NBLK * 16 * 64; e += 512) {
    const int ln = e & 63, j = (e >> 6) & 15, blk = e >> 10;
    o[(blk * 32 + acc_row(j, ln >> 5)) * XC + (ln & 31)] = zb[e];
  }
  if (threadIdx.x < MAXH * HC) {
    const float* q = sb + (threadIdx.x >> 5) * 64 + (threadIdx.x & 31);
    o[RED_Z + threadIdx.x] = q[0] + q[32];
  }
  if (threadIdx.x < MAXH) o[RED_Z + MAXH * HC + threadIdx.x] = db[threadIdx.x];
}

// finalize, one block per gate channel: parameter gradients and the apply-pass coefficients
__global__ __launch_bounds__(64) void gate_bwd_finalize_kernel(const float* __restrict__ part, const double* __restrict__ mom_sum,
                                                               const float* __restrict__ W1, const float* __restrict__ b1,
                                                               const float* __restrict__ scale, const float* __restrict__ shift,
                                                               const float* __restrict__ mean, const float* __restrict__ rstd,
                                                               const float* __restrict__ gamma, const float* __restrict__ W2,
                                                               float* __restrict__ dW1, float* __restrict__ dgamma,
                                                               float* __restrict__ dbeta, float* __restrict__ dW2,
                                                               float* __restrict__ db2, float* __restrict__ coef, double count) {
  __shared__ double zs[XC + 2];
  const int ch = blockIdx.x, t = threadIdx.x;
  if (t < XC + 2) {
    const bool want = t < XC + 1 || (ch % HC) == 0;
    const size_t col = t < XC ? (size_t)ch * XC + t : (t == XC ? (size_t)RED_Z + ch : (size_t)RED_Z + MAXH * HC + ch / HC);
    double s0 = 0., s1 = 0., s2 = 0., s3 = 0.;
    if (want) {
      for (int b = 0; b < RED_BLOCKS; b += 4) {
        s0 += part[(size_t)b * RED_N + col];
        s1 += part[(size_t)(b + 1) * RED_N + col];
        s2 += part[(size_t)(b + 2) * RED_N + col];
        s3 += part[(size_t)(b + 3) * RED_N + col];
      }
    }
    zs[t] = (s0 + s1) + (s2 + s3);
  }
  __syncthreads();
  const float* w = W1 + (size_t)ch * XC;
  const double Sr = zs[XC], w2 = W2[ch], gm = gamma[ch], rs = rstd[ch], mu = mean[ch], bb = b1[ch];
  double wz = 0.;
  for (int k = 0; k < XC; ++k) wz += (double)w[k] * zs[k];
  const double Sb = w2 * Sr;                                  // sum_p da,  da = dr * w2
  const double Sg = rs * (w2 * wz + (bb - mu) * Sb);          // sum_p da * that
  const double c1 = gm * Sb / count, c2 = gm * Sg / count;
  if (t == 0) {
    dgamma[ch] += (float)Sg;
    dbeta[ch] += (float)Sb;
    dW2[ch] += (float)((double)scale[ch] * (wz + bb * Sr) + (double)shift[ch] * Sr);  // sum_p dpsi * relu(a)
    if ((ch % HC) == 0) db2[ch / HC] += (float)zs[XC + 1];
    const double K2 = -rs * rs * c2, K3 = -rs * c1 + mu * rs * rs * c2;
    coef[4 * ch] = (float)(rs * w2 * gm);
    coef[4 * ch + 1] = (float)K2;
    coef[4 * ch + 2] = (float)(K2 * bb + K3);
    coef[4 * ch + 3] = 0.f;
  }
  if (t < XC) {
    double wm2 = 0.;
    for (int j = 0; j < XC; ++j) wm2 += (double)w[j] * mom_sum[XC + j * XC + t];
    const double tx = rs * (wm2 + (bb - mu) * mom_sum[t]);  // sum_p that_p x_p[t]
    dW1[(size_t)ch * XC + t] += (float)(rs * (gm * w2 * zs[t] - c1 * mom_sum[t] - c2 * tx));
  }
}

// Q[j][k] = sum_ch W1[ch,j] K2[ch] W1[ch,k],  r[k] = sum_ch (K2 b1 + K3)[ch] W1[ch,k]
__global__ __launch_bounds__(1024) void gate_bwd_affine_kernel(const float* __restrict__ W1, const float* __restrict__ coef,
                                                               float* __restrict__ Q, float* __restrict__ r, int nch) {
  __shared__ float ws[MAXH * HC * XC];
  __shared__ float k2[MAXH * HC], k3[MAXH * HC];
  for (int e = threadIdx.x; e < nch * XC; e += 1024) ws[e] = W1[e];
  for (int e = threadIdx.x; e < nch; e += 1024) k2[e] = coef[4 * e + 1], k3[e] = coef[4 * e + 2];
  __syncthreads();
  const int j = threadIdx.x >> 5, k = threadIdx.x & 31;
  double q = 0., rr = 0.;
  for (int ch = 0; ch < nch; ++ch) {
    q += (double)ws[ch * XC + j] * k2[ch] * ws[ch * XC + k];
    if (j == 0) rr += (double)k3[ch] * ws[ch * XC + k];
  }
  Q[threadIdx.x] = (float)q;
  if (j == 0) r[k] = (float)rr;
}

// apply pass: dF[p,k] = dXc[p,k] + x[p,:] Q[:,k] + r[k] + sum_ch dr[p,ch] K1[ch] W1[ch,k]
__global__ __launch_bounds__(256, 2) void gate_bwd_apply_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ G,
                                                                const float* __restrict__ dG, const float* __restrict__ dXc,
                                                                const float* __restrict__ W1, const float* __restrict__ b1,
                                                                const float* __restrict__ scale, const float* __restrict__ shift,
                                                                const float* __restrict__ coef, const float* __restrict__ Q,
                                                                const float* __restrict__ r, bf16_t* __restrict__ dF,
                                                                long long M, int NH) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l31 = lane & 31, half = lane >> 5;
  const int nch = NH * HC, nblk = (nch + 31) / 32;
  __shared__ bf16x8 wK[NBLK * 2][64];  // K1*W1 operand, lane-indexed: rows = x channel l31, K slots = chained channels
  __shared__ bf16x8 wAl[NBLK * 2][64];  // lo parts of W1s
  bf16x8 wA[NBLK][2], qh[2], ql[2];
  float biasf[NBLK];
  uint32_t wbias[NBLK];
  load_w1s(W1, b1, scale, shift, nch, lane, wave == 0, wA, wAl, biasf);
#pragma unroll
  for (int blk = 0; blk < NBLK; ++blk) wbias[blk] = half == 0 ? split_bf(biasf[blk]) : 0u;
  for (int f = wave; f < NBLK * 2; f += 4) {
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int c = (f >> 1) * 32 + acc_row(8 * (f & 1) + e, half);
      const int cc = c < nch ? c : nch - 1;
      v[e] = coef[4 * cc] * W1[(size_t)cc * XC + l31] * (c < nch ? 1.f : 0.f);
    }
    wK[f][lane] = frag8(v);
  }
  __syncthreads();
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {  // Q^T as (hi, lo) operands: row = k (l31), K slot = j
    float f[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) f[e] = Q[(ks * 16 + half * 8 + e) * XC + l31];
    frag8_split(f, qh[ks], ql[ks]);
  }
  const uint32_t rw = half == 0 ? split_bf(r[l31]) : 0u;
  const bf16x8 ones = frag_lo(half == 0 ? MVIT_ONE2 : 0u);

  const long long ntile = (M + 31) / 32;
  for (long long t = (long long)blockIdx.x * 4 + wave; t < ntile; t += (long long)gridDim.x * 4) {
    const long long p = t * 32 + l31;
    const bool live = p < M;
    Frag x0, x1;
    load_x_frags(x, p, live, half, x0, x1);
    float dpsi[MAXH];
    if (live) {
      float g[MAXH];
      load_g16(G + (size_t)p * MAXH, g);
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const float4 d = ((const float4*)(dG + (size_t)p * MAXH))[v];
        dpsi[4 * v] = d.x * g[4 * v] * (1.f - g[4 * v]);
        dpsi[4 * v + 1] = d.y * g[4 * v + 1] * (1.f - g[4 * v + 1]);
        dpsi[4 * v + 2] = d.z * g[4 * v + 2] * (1.f - g[4 * v + 2]);
        dpsi[4 * v + 3] = d.w * g[4 * v + 3] * (1.f - g[4 * v + 3]);
      }
    } else {
#pragma unroll
      for (int h = 0; h < MAXH; ++h) dpsi[h] = 0.f;
    }
    f32x16 out = mfma(qh[0], x0.v, zero16());  // out^T[k, px]
    out = mfma(qh[1], x1.v, out);
    out = mfma(ql[0], x0.v, out);
    out = mfma(ql[1], x1.v, out);
    out = mfma(frag_lo(rw), ones, out);
#pragma unroll
    for (int blk = 0; blk < NBLK; ++blk) {
      if (blk < nblk) {
        f32x16 a = mfma(wA[blk][0], x0.v, zero16());  // a[ch, px]
        a = mfma(wA[blk][1], x1.v, a);
        a = mfma(wAl[2 * blk][lane], x0.v, a);
        a = mfma(wAl[2 * blk + 1][lane], x1.v, a);
        a = mfma(frag_lo(wbias[blk]), ones, a);
        float dr[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) dr[j] = a[j] > 0.f ? dpsi[2 * blk + (j >> 3)] : 0.f;
        out = mfma(wK[2 * blk][lane], frag8(dr), out);
        out = mfma(wK[2 * blk + 1][lane], frag8(dr + 8), out);
      }
    }
    if (live) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {  // rows (x channels) 8q + 4*half + {0..3} of this lane's pixel
        const int k = 8 * q + 4 * half;
        const float4 d = *(const float4*)(dXc + (size_t)p * XC + k);
        *(uint2*)(dF + (size_t)p * XC + k) =
            make_uint2(pack2bf(out[4 * q] + d.x, out[4 * q + 1] + d.y), pack2bf(out[4 * q + 2] + d.z, out[4 * q + 3] + d.w));
      }
    }
  }
}

inline int nblk(long long work, int per, int cap) {
  long long b = (work + per - 1) / per;
  return (int)(b < 1 ? 1 : (b > cap ? cap : b));
}

}  // namespace

extern "C" {

MVIT_API int mvit_heads_moments(const void* x, double* mom, long long M, int nslots, mvit_stream_t stream) {
  MVIT_CLEAR_ERROR();
  if (M <= 0 || nslots <= 0) return MVIT_EINVAL;
  hipLaunchKernelGGL(moments_kernel, dim3(nblk(M, 32 * 4 * 4, nslots >= 256 && nslots < 512 ? nslots : 512)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, mom, M,
                     nslots);
  return MVIT_LAUNCH_CHECK();
}

MVIT_API int mvit_heads_bn_from_moments(const double* mom, const float* W1, const float* b1, const float* gamma,
                                        const float* beta, float* running_mean, float* running_var, float* scale,
                                        float* shift, float* mean_out, float* rstd_out, double* mom_sum, int NH, int nslots,
                                        double count, float eps, float momentum, int training, mvit_stream_t stream) {
  MVIT_CLEAR_ERROR();
  if (NH <= 0 || NH > MAXH || (training && (!mom || nslots <= 0 || count <= 0))) return MVIT_EINVAL;
  hipLaunchKernelGGL(bn_from_moments_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, mom, W1, b1, gamma, beta,
                     running_mean, running_var, scale, shift, mean_out, rstd_out, mom_sum, NH * HC, nslots, count, eps,
                     momentum, training);
  return MVIT_LAUNCH_CHECK();
}

MVIT_API int mvit_heads_gate_fwd(const void* x, const float* W1, const float* b1, const float* scale, const float* shift,
                                 const float* W2, const float* b2, void* G, long long M, int NH, mvit_stream_t stream) {
  MVIT_CLEAR_ERROR();
  if (M <= 0 || NH <= 0 || NH > MAXH) return MVIT_EINVAL;
  hipLaunchKernelGGL(gate_fwd_kernel, dim3(nblk(M, 32 * 4 * 4, 512)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, W1, b1,
                     scale, shift, W2, b2, (bf16_t*)G, M, NH);
  return MVIT_LAUNCH_CHECK();
}

MVIT_API int mvit_heads_conv_fwd(const void* x, const void* G, const float* W3, const float* b3, float* out, int B, int H,
                                 int W, int NH, mvit_stream_t stream) {
  MVIT_CLEAR_ERROR();
  if (B <= 0 || H <= 0 || W <= 0 || NH <= 0 || NH > MAXH) return MVIT_EINVAL;
  static mvit_per_device_size raised;
  if (mvit_ensure_dynamic_lds((const void*)conv_fwd_kernel, CF_LDS, raised) != MVIT_OK) return MVIT_EINVAL;
  const long long blocks = (long long)B * ((H + CF_TH - 1) / CF_TH) * ((W + CF_TW - 1) / CF_TW);
  if (blocks >= (1ll << 30)) return MVIT_EINVAL;
  hipLaunchKernelGGL(conv_fwd_kernel, dim3((unsigned)(blocks < 256 ? blocks : 256)), dim3(512), CF_LDS, (hipStream_t)stream,
                     (const bf16_t*)x, (const bf16_t*)G, W3, b3, out, B, H, W, NH);
  return MVIT_LAUNCH_CHECK();
}

MVIT_API long long mvit_heads_conv_bwd_scratch_bytes(long long M) {
  return (long long)(((size_t)M * MAXH * sizeof(bf16_t) + 255) / 256 * 256 + (size_t)CB_BLOCKS * CB_PART * sizeof(float) +
                     (size_t)DZ_BLOCKS * MAXH * sizeof(float));
}

MVIT_API int mvit_heads_conv_bwd(const float* dY, const float* Y, const void* x, const void* G, const float* W3, void* scratch,
                                 long long scratch_bytes, float* dG, float* dXc, float* dW3, float* db3, int B, int H, int W,
                                 int NH, mvit_stream_t stream) {
  MVIT_CLEAR_ERROR();
  if (B <= 0 || H <= 0 || W <= 0 || NH <= 0 || NH > MAXH || !scratch) return MVIT_EINVAL;
  const long long M = (long long)B * H * W;
  if (M >= (1ll << 31) - 64 || scratch_bytes < mvit_heads_conv_bwd_scratch_bytes(M)) return MVIT_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  bf16_t* dzb = (bf16_t*)scratch;
  float* part = (float*)((char*)scratch + ((size_t)M * MAXH * sizeof(bf16_t) + 255) / 256 * 256);
  float* dbpart = part + (size_t)CB_BLOCKS * CB_PART;
  const int dzblocks = nblk(M, 256, DZ_BLOCKS);
  if (NH == MAXH)
    hipLaunchKernelGGL(conv_bwd_dz_kernel<true>, dim3(dzblocks), dim3(256), 0, s, dY, Y, dzb, dbpart, B, (long long)H * W, NH);
  else
    hipLaunchKernelGGL(conv_bwd_dz_kernel<false>, dim3(dzblocks), dim3(256), 0, s, dY, Y, dzb, dbpart, B, (long long)H * W, NH);
  hipLaunchKernelGGL(conv_bwd_kernel, dim3(CB_BLOCKS), dim3(256), 0, s, (const bf16_t*)dzb, (const bf16_t*)x, (const bf16_t*)G, W3,
                     dG, dXc, part, B, H, W, NH);
  static_assert(CB_BLOCKS % 32 == 0, "dw3: eight row segments of whole groups of four");
  hipLaunchKernelGGL(conv_bwd_dw3_kernel, dim3((NH * 9 * XC * 8 + 255) / 256), dim3(256), 0, s, (const float*)part, dW3, NH, CB_BLOCKS,
                     (const float*)dbpart, dzblocks, db3);
  return MVIT_LAUNCH_CHECK();
}

MVIT_API long long mvit_heads_gate_bwd_scratch_bytes(void) { return (long long)(SCR_FLOATS * sizeof(float)); }

MVIT_API int mvit_heads_gate_bwd(const void* x, const void* G, const float* dG, const float* dXc, const float* W1,
                                 const float* b1, const float* scale, const float* shift, const float* mean,
                                 const float* rstd, const float* gamma, const float* W2, const double* mom_sum,
                                 void* scratch, long long scratch_bytes, float* dW1, float* dgamma, float* dbeta, float* dW2,
                                 float* db2, void* dF, long long M, int NH, double count, mvit_stream_t stream) {
  MVIT_CLEAR_ERROR();
  if (M <= 0 || NH <= 0 || NH > MAXH || !scratch || scratch_bytes < (long long)(SCR_FLOATS * sizeof(float))) return MVIT_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  float* part = (float*)scratch;
  float *coef = part + SCR_COEF, *Q = part + SCR_Q, *r = part + SCR_R;
  hipLaunchKernelGGL(gate_bwd_reduce_kernel, dim3(RED_BLOCKS), dim3(512), 0, s, (const bf16_t*)x, (const bf16_t*)G, dG, W1, b1,
                     scale, shift, part, M, NH);
  hipLaunchKernelGGL(gate_bwd_finalize_kernel, dim3(NH * HC), dim3(64), 0, s, part, mom_sum, W1, b1, scale, shift, mean, rstd,
                     gamma, W2, dW1, dgamma, dbeta, dW2, db2, coef, count);
  hipLaunchKernelGGL(gate_bwd_affine_kernel, dim3(1), dim3(1024), 0, s, W1, coef, Q, r, NH * HC);
  hipLaunchKernelGGL(gate_bwd_apply_kernel, dim3(nblk(M, 32 * 4 * 4, 512)), dim3(256), 0, s, (const bf16_t*)x, (const bf16_t*)G,
                     dG, dXc, W1, b1, scale, shift, coef, Q, r, (bf16_t*)dF, M, NH);
  return MVIT_LAUNCH_CHECK();
}

}  // extern "C"
