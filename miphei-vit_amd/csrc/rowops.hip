// Row-wise / skinny kernels of the ViT encoder: LayerNorm fwd/bwd (wave per row, wave-level
// reductions over 64 lanes), rank-r LoRA products, patch gather, prefix tokens, casts.
// All HBM-bound: 16-byte loads per lane, rows kept in registers between the reduction passes.
#include "common.hpp"
#include "../../include/miphei_hip.h"

namespace {

constexpr int LN_MAXV = 8;  // float4 per lane -> D <= 2048
}  // namespace
constexpr int LN_MAXV_GENERIC = 8;
namespace {

// ------------------------------------------------------------------ LayerNorm forward
// NVF > 0: the row is exactly NVF full groups of 64 float4 (D = 256 * NVF: 1536 = 6): no per-group bounds test.  With the test
// (`if (idx < nv) v[i] = xr[idx]` in an unrolled loop) hipcc branches around every load and waits for it before the next one:
// a wave paid one HBM round trip PER GROUP (9 `s_waitcnt vmcnt(0)` for 10 loads in this kernel, 13 in the backward one).
template <int NVF>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                     const float* __restrict__ b, bf16_t* __restrict__ out, int M, int D,
                                                     float eps) {
  // No FMA contraction in the statistics / normalisation: the three LayerNorm forward kernels must give the SAME bits for a row (the fused
  // kernels are checked against ln_fwd bit for bit), and hipcc otherwise contracts a*a + b*b one way in one inlined copy and the other
  // way in the next (1-ulp flips in ~1 of 10^6 outputs, round 4).
#pragma clang fp contract(off)

  constexpr int LN_MAXV = NVF ? NVF : ::LN_MAXV_GENERIC;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= M) return;
  const int nv = NVF ? 64 * NVF : D >> 2;
  const float4* xr = (const float4*)(x + (size_t)row * D);
  float4 v[LN_MAXV];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < LN_MAXV; ++i) {
    const int idx = lane + 64 * i;
    if (NVF || idx < nv) {
      v[i] = xr[idx];
      s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    }
  }
  const float mu = wave_sum(s) / D;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < LN_MAXV; ++i) {
    const int idx = lane + 64 * i;
    if (NVF || idx < nv) {
      const float a = v[i].x - mu, bb = v[i].y - mu, c = v[i].z - mu, d = v[i].w - mu;
      q += (a * a + bb * bb) + (c * c + d * d);
    }
  }
  const float rs = rsqrtf(wave_sum(q) / D + eps);
  uint2* o = (uint2*)(out + (size_t)row * D);
  // all results first, then all stores: a load issued behind a store waits for that store too (vmcnt counts both)
  uint2 r[LN_MAXV];
#pragma unroll
  for (int i = 0; i < LN_MAXV; ++i) {
    const int idx = lane + 64 * i;
    if (NVF || idx < nv) {
      const float4 ww = ((const float4*)w)[idx], bv = ((const float4*)b)[idx];
      r[i].x = pack2bf((v[i].x - mu) * rs * ww.x + bv.x, (v[i].y - mu) * rs * ww.y + bv.y);
      r[i].y = pack2bf((v[i].z - mu) * rs * ww.z + bv.z, (v[i].w - mu) * rs * ww.w + bv.w);
    }
  }
#pragma unroll
  for (int i = 0; i < LN_MAXV; ++i) {
    const int idx = lane + 64 * i;
    if (NVF || idx < nv) o[idx] = r[i];
  }
}

// ------------------------------------------------------------------ LayerNorm forward + LoRA down-projection
// h = LN(x) (bf16) and t = h @ [A_q | A_v]  ([M, R2], R2 = 2*rank <= 16) in one launch: the LoRA products of QkvWithLoRA
// (src/generators/lora.py:16-18,29-33) start from the same LN1 output the qkv GEMM reads, so the separate skinny GEMM (one more
// pass over h, one more launch per block) disappears.
// A block normalises 16 rows, one per wave (row in registers, wave-level statistics, as ln_fwd_kernel), and parks the bf16 rows
// in LDS next to the adapter matrix AcatT [R2, D]; after one barrier four of its waves run the 16-row x 16-column product on
// v_mfma_f32_16x16x32_bf16 -- A fragments (row = token) and B fragments (column = adapter column) are plain 16-byte LDS reads,
// the 32-wide K steps are dealt to the four waves and their partial blocks summed through LDS.  Row images are padded by 16 bytes
// so that the 16 rows of a fragment read fall on distinct banks.
#ifndef MVIT_LNL_ROWS
#define MVIT_LNL_ROWS 16
#endif
constexpr int LNL_ROWS = MVIT_LNL_ROWS;   // rows (= waves) per block: 16, or 8 with the upper half of the MFMA rows idle
template <int NV, bool FULL = false>  // float4 groups per lane: D <= 256 * NV (FULL: D == 256 * NV, no bounds tests, see ln_fwd_kernel)
__global__ __launch_bounds__(64 * LNL_ROWS) void ln_fwd_lora_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                                    const float* __restrict__ b, bf16_t* __restrict__ out,
                                                                    const bf16_t* __restrict__ AcatT, bf16_t* __restrict__ t,
                                                                    int M, int D, float eps, int R2) {
  // No FMA contraction in the statistics / normalisation: the three LayerNorm forward kernels must give the SAME bits for a row (the fused
  // kernels are checked against ln_fwd bit for bit), and hipcc otherwise contracts a*a + b*b one way in one inlined copy and the other
  // way in the next (1-ulp flips in ~1 of 10^6 outputs, round 4).
#pragma clang fp contract(off)

  extern __shared__ __attribute__((aligned(16))) char lds_raw[];
  const int RS = D * 2 + 16;                          // padded row image (bytes)
  char* As = lds_raw;                                 // [16][RS]: AcatT rows (rows >= R2 zero)
  char* Hs = lds_raw + 16 * RS;                       // [LNL_ROWS][RS]: the block's normalised rows
  float* part = (float*)(Hs + LNL_ROWS * RS);         // [3][64][4] partial blocks of waves 1..3
  const int nv = D >> 2;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int row = blockIdx.x * LNL_ROWS + wave;
  float4 v[NV];
  {
    const float4* xr = (const float4*)(x + (size_t)(row < M ? row : M - 1) * D) + lane;   // + 64 * i: immediate offsets
#pragma unroll
    for (int i = 0; i < NV; ++i)
      if (FULL || lane + 64 * i < nv) v[i] = xr[64 * i];
  }
  {
    const int h16 = D >> 3;                           // 16-byte pieces per adapter row (D % 8 == 0 is checked by the host)
    for (int i = threadIdx.x; i < 16 * h16; i += 64 * LNL_ROWS) {
      const int j = i / h16, c = i - j * h16;
      uint4 tq = ((const uint4*)AcatT)[(size_t)(j < R2 ? j : R2 - 1) * h16 + c];   // unconditional load (see ln_fwd_kernel), masked
      const unsigned keep = j < R2 ? 0xffffffffu : 0u;
      tq.x &= keep, tq.y &= keep, tq.z &= keep, tq.w &= keep;
      *(uint4*)(As + j * RS + c * 16) = tq;
    }
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i)
    if (FULL || lane + 64 * i < nv) s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
  const float mu = wave_sum(s) / D;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i)
    if (FULL || lane + 64 * i < nv) {
      const float a = v[i].x - mu, bb = v[i].y - mu, c = v[i].z - mu, d = v[i].w - mu;
      q += (a * a + bb * bb) + (c * c + d * d);
    }
  const float rs = rsqrtf(wave_sum(q) / D + eps);
  {
    uint2* o = (uint2*)(out + (size_t)row * D) + lane;
    uint2* hl = (uint2*)(Hs + wave * RS) + lane;
    const float4* wl = (const float4*)w + lane;
    const float4* bl = (const float4*)b + lane;
    uint2 r[NV];     // all results first, then all stores (see ln_fwd_kernel)
#pragma unroll
    for (int i = 0; i < NV; ++i)
      if (FULL || lane + 64 * i < nv) {
        const float4 ww = wl[64 * i], bv = bl[64 * i];
        r[i].x = pack2bf((v[i].x - mu) * rs * ww.x + bv.x, (v[i].y - mu) * rs * ww.y + bv.y);
        r[i].y = pack2bf((v[i].z - mu) * rs * ww.z + bv.z, (v[i].w - mu) * rs * ww.w + bv.w);
        hl[64 * i] = r[i];
      }
    if (row < M) {
#pragma unroll
      for (int i = 0; i < NV; ++i)
        if (FULL || lane + 64 * i < nv) o[64 * i] = r[i];
    }
  }
  __syncthreads();
  // t block [16 tokens][16 columns]: K steps of 32, step ks handled by wave ks % 4.  Every wave of the block stays alive through
  // the second barrier (waves 4.. only wait there): nothing relies on how the hardware counts exited waves at a barrier.
  const int r16 = lane & 15, kq = lane >> 4;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  if (wave < 4) {
    const char* ha = Hs + (r16 & (LNL_ROWS - 1)) * RS + kq * 16;   // (rows beyond LNL_ROWS repeat: their results are not stored)
    const char* ab = As + r16 * RS + kq * 16;
    const int nks = D >> 5;
    for (int ks = wave; ks < nks; ks += 4) {
      const bf16x8 fa = *(const bf16x8*)(ha + ks * 64);
      const bf16x8 fb = *(const bf16x8*)(ab + ks * 64);
      acc = mvit_mfma16(fa, fb, acc, 0, 0, 0);
    }
    if (wave > 0) {
#pragma unroll
      for (int j = 0; j < 4; ++j) part[((wave - 1) * 64 + lane) * 4 + j] = acc[j];
    }
  }
  __syncthreads();                      // the partial blocks of waves 1..3 are in LDS
  // C/D layout of the 16x16 MFMA: col = lane & 15 (adapter column), row = (lane >> 4) * 4 + reg (token)
  if (wave == 0 && r16 < R2) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int orow = blockIdx.x * LNL_ROWS + kq * 4 + j;
      if (kq * 4 + j >= LNL_ROWS) continue;
      const float val = acc[j] + part[(0 * 64 + lane) * 4 + j] + part[(1 * 64 + lane) * 4 + j] + part[(2 * 64 + lane) * 4 + j];
      if (orow < M) t[(size_t)orow * R2 + r16] = f2bf(val);
    }
  }
}

// Round 4: the same arithmetic on a grid that fits the chip in ONE round.  ln_fwd_lora_kernel runs 16 rows per block with ~100 KB of LDS,
// i.e. one block per CU: M = 5264 rows are 329 blocks = 1.29 rounds on 256 CUs, and a round of this kernel is one dependent chain
// (row loads -> statistics -> stores -> barrier -> MFMA -> barrier -> store): 16.6 us in the training step at 3.0 TB/s where ln_fwd moves
// its rows at 4.7.  Here a block owns a BALANCED share of the rows (ceil(M / blocks) <= 32: 21 at M = 5264 on 251 blocks), every wave
// normalises up to TWO rows whose loads are all requested before the first use, the adapter matrix is staged once per block, and the
// two 16-row groups share one pair of barriers: the product is 8 wave tasks (row group x K quarter) on waves 0-7.
template <int NV, bool FULL = false>
__global__ __launch_bounds__(1024) void ln_fwd_lora2_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                            const float* __restrict__ b, bf16_t* __restrict__ out,
                                                            const bf16_t* __restrict__ AcatT, bf16_t* __restrict__ t,
                                                            int M, int D, float eps, int R2, int rows_per_block) {
  // No FMA contraction in the statistics / normalisation: the three LayerNorm forward kernels must give the SAME bits for a row (the fused
  // kernels are checked against ln_fwd bit for bit), and hipcc otherwise contracts a*a + b*b one way in one inlined copy and the other
  // way in the next (1-ulp flips in ~1 of 10^6 outputs, round 4).
#pragma clang fp contract(off)

  extern __shared__ __attribute__((aligned(16))) char lds_raw[];
  const int RS = D * 2 + 16;                          // padded row image (bytes)
  char* As = lds_raw;                                 // [16][RS]: AcatT rows (rows >= R2 zero)
  char* Hs = lds_raw + 16 * RS;                       // [32][RS]: the block's normalised rows (two groups of 16)
  float* part = (float*)(Hs + 32 * RS);               // [8][64][4] partial blocks (row group x K quarter)
  const int nv = D >> 2;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r_begin = blockIdx.x * rows_per_block, r_end = min(M, r_begin + rows_per_block);
  {
    const int h16 = D >> 3;                           // 16-byte pieces per adapter row
    for (int i = threadIdx.x; i < 16 * h16; i += 1024) {
      const int j = i / h16, c = i - j * h16;
      uint4 tq = ((const uint4*)AcatT)[(size_t)(j < R2 ? j : R2 - 1) * h16 + c];   // unconditional load, masked
      const unsigned keep = j < R2 ? 0xffffffffu : 0u;
      tq.x &= keep, tq.y &= keep, tq.z &= keep, tq.w &= keep;
      *(uint4*)(As + j * RS + c * 16) = tq;
    }
  }
  for (int base = r_begin; base < r_end; base += 32) {
    const int rowA = base + wave, rowB = base + 16 + wave;
    const bool okA = rowA < r_end, okB = rowB < r_end;
    float4 va[NV], vb[NV];
    {
      const float4* xa = (const float4*)(x + (size_t)(okA ? rowA : r_end - 1) * D) + lane;   // + 64 * i: immediate offsets
      const float4* xb = (const float4*)(x + (size_t)(okB ? rowB : r_end - 1) * D) + lane;
#pragma unroll
      for (int i = 0; i < NV; ++i)
        if (FULL || lane + 64 * i < nv) va[i] = xa[64 * i];
#pragma unroll
      for (int i = 0; i < NV; ++i)
        if (FULL || lane + 64 * i < nv) vb[i] = xb[64 * i];
    }
    auto norm_row = [&](float4 (&v)[NV], int row, bool ok, int slot) __attribute__((always_inline)) {
      float s = 0.f;
#pragma unroll
      for (int i = 0; i < NV; ++i)
        if (FULL || lane + 64 * i < nv) s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
      const float mu = wave_sum(s) / D;
      float q = 0.f;
#pragma unroll
      for (int i = 0; i < NV; ++i)
        if (FULL || lane + 64 * i < nv) {
          const float a = v[i].x - mu, bb = v[i].y - mu, c = v[i].z - mu, d = v[i].w - mu;
          q += (a * a + bb * bb) + (c * c + d * d);
        }
      const float rs = rsqrtf(wave_sum(q) / D + eps);
      uint2* o = (uint2*)(out + (size_t)row * D) + lane;
      uint2* hl = (uint2*)(Hs + slot * RS) + lane;
      const float4* wl = (const float4*)w + lane;
      const float4* bl = (const float4*)b + lane;
      uint2 r[NV];     // all results first, then all stores (see ln_fwd_kernel)
#pragma unroll
      for (int i = 0; i < NV; ++i)
        if (FULL || lane + 64 * i < nv) {
          const float4 ww = wl[64 * i], bv = bl[64 * i];
          r[i].x = pack2bf((v[i].x - mu) * rs * ww.x + bv.x, (v[i].y - mu) * rs * ww.y + bv.y);
          r[i].y = pack2bf((v[i].z - mu) * rs * ww.z + bv.z, (v[i].w - mu) * rs * ww.w + bv.w);
          hl[64 * i] = r[i];
        }
      if (ok) {
#pragma unroll
        for (int i = 0; i < NV; ++i)
          if (FULL || lane + 64 * i < nv) o[64 * i] = r[i];
      }
    };
    norm_row(va, rowA, okA, wave);
    norm_row(vb, rowB, okB, 16 + wave);
    __syncthreads();
    // t blocks [16 tokens][16 columns] of the two row groups: K steps of 32, wave task = (group, K quarter) on waves 0-7
    const int r16 = lane & 15, kq = lane >> 4;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    if (wave < 8) {
      const int grp = wave >> 2, kw = wave & 3;
      const char* ha = Hs + (grp * 16 + r16) * RS + kq * 16;
      const char* ab = As + r16 * RS + kq * 16;
      const int nks = D >> 5;
      for (int ks = kw; ks < nks; ks += 4) {
        const bf16x8 fa = *(const bf16x8*)(ha + ks * 64);
        const bf16x8 fb = *(const bf16x8*)(ab + ks * 64);
        acc = mvit_mfma16(fa, fb, acc, 0, 0, 0);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) part[(wave * 64 + lane) * 4 + j] = acc[j];
    }
    __syncthreads();                      // the partial blocks are in LDS
    // C/D layout of the 16x16 MFMA: col = lane & 15 (adapter column), row = (lane >> 4) * 4 + reg (token)
    if ((wave == 0 || wave == 4) && r16 < R2) {
      const int grp = wave >> 2;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int orow = base + grp * 16 + kq * 4 + j;
        float val = 0.f;
#pragma unroll
        for (int k4 = 0; k4 < 4; ++k4) val += part[((grp * 4 + k4) * 64 + lane) * 4 + j];   // fixed order: run-to-run identical
        if (orow < r_end) t[(size_t)orow * R2 + r16] = f2bf(val);
      }
    }
    if (base + 32 < r_end) __syncthreads();   // (more rows: the row images and partial blocks are rewritten)
  }
}

// ------------------------------------------------------------------ LayerNorm backward (input grad only)
// dx (+)= rstd * (g - mean(g) - xhat * mean(g*xhat)),  g = dh * w.  Statistics are recomputed from x.
// Optionally emits dy = bf16(gamma_next * dx_total): the LayerScale-scaled gradient the next dgrad GEMM consumes.
// (A leaner variant -- gradient row kept packed, running gradient fetched after the reductions, 80 registers = 6 waves per SIMD so
// that all 5264 rows of the training shape are resident in one round -- measured 23.3 us like this one in isolation and 0.5 % slower
// inside the step, 433.4 vs 435.8 tiles/s same box: the late loads of the running gradient are exposed there.)
template <int NV, bool FULL = false>  // float4 groups per lane (D <= 256 * NV; FULL: D == 256 * NV, no bounds tests -- see
                                       // ln_fwd_kernel); the three row images below are 12 * NV registers
__global__ __launch_bounds__(256) void ln_bwd_kernel(const bf16_t* __restrict__ dh, const float* __restrict__ x,
                                                     const float* __restrict__ w, float* __restrict__ dx,
                                                     const float* __restrict__ gamma_next, bf16_t* __restrict__ dy,
                                                     int M, int D, float eps, int accumulate,
                                                     const float* __restrict__ rowscale_next) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= M) return;
  const float rsn = rowscale_next ? rowscale_next[row] : 1.f;   // DropPath factor of the branch dy feeds (per sample)
  const int nv = FULL ? 64 * NV : D >> 2;
  const float4* xr = (const float4*)(x + (size_t)row * D);
  const uint2* gr = (const uint2*)(dh + (size_t)row * D);
  float4 v[NV], g[NV], o[NV];
  float4* dxr = (float4*)(dx + (size_t)row * D);
  float s = 0.f;
  // every operand of the row is requested before anything is used (the running gradient too: read after the reductions it would
  // sit behind each store of the last loop -- same array -- one exposed HBM latency per element group)
  uint2 gp[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int idx = lane + 64 * i;
    if (FULL || idx < nv) {
      v[i] = xr[idx];
      gp[i] = gr[idx];
      g[i] = ((const float4*)w)[idx];
      if (accumulate) o[i] = dxr[idx];
    }
  }
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int idx = lane + 64 * i;
    if (FULL || idx < nv) {
      if (!accumulate) o[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
      g[i].x *= lo16f(gp[i].x);
      g[i].y *= hi16f(gp[i].x);
      g[i].z *= lo16f(gp[i].y);
      g[i].w *= hi16f(gp[i].y);
    }
  }
  const float mu = wave_sum(s) / D;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int idx = lane + 64 * i;
    if (FULL || idx < nv) {
      v[i].x -= mu; v[i].y -= mu; v[i].z -= mu; v[i].w -= mu;
      q += (v[i].x * v[i].x + v[i].y * v[i].y) + (v[i].z * v[i].z + v[i].w * v[i].w);
    }
  }
  const float rs = rsqrtf(wave_sum(q) / D + eps);
  float c1 = 0.f, c2 = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int idx = lane + 64 * i;
    if (FULL || idx < nv) {
      v[i].x *= rs; v[i].y *= rs; v[i].z *= rs; v[i].w *= rs;  // xhat
      c1 += (g[i].x + g[i].y) + (g[i].z + g[i].w);
      c2 += (g[i].x * v[i].x + g[i].y * v[i].y) + (g[i].z * v[i].z + g[i].w * v[i].w);
    }
  }
  c1 = wave_sum(c1) / D;
  c2 = wave_sum(c2) / D;
  uint2* dyr = dy ? (uint2*)(dy + (size_t)row * D) : nullptr;
  // the results replace the running-gradient image, the LayerScale vector of the next product is fetched for the whole row, and
  // only then do the stores start: a load between two stores waits for the store in front of it (vmcnt counts both)
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int idx = lane + 64 * i;
    if (FULL || idx < nv) {
      o[i].x += rs * (g[i].x - c1 - v[i].x * c2);
      o[i].y += rs * (g[i].y - c1 - v[i].y * c2);
      o[i].z += rs * (g[i].z - c1 - v[i].z * c2);
      o[i].w += rs * (g[i].w - c1 - v[i].w * c2);
      if (dyr) g[i] = ((const float4*)gamma_next)[idx];
    }
  }
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int idx = lane + 64 * i;
    if (FULL || idx < nv) {
      dxr[idx] = o[i];
      if (dyr) {
        uint2 p;
        p.x = pack2bf(o[i].x * g[i].x * rsn, o[i].y * g[i].y * rsn);
        p.y = pack2bf(o[i].z * g[i].z * rsn, o[i].w * g[i].w * rsn);
        dyr[idx] = p;
      }
    }
  }
}

// ------------------------------------------------------------------ out[M,R] = X[M,K] @ W[R,K]^T, R <= 16
// LoRA down-projection (x @ A) and its adjoint (dq @ B^T): M is large, R tiny.  One block per 16 rows on
// v_mfma_f32_16x16x32_bf16 (N = 16 exactly); the block's four wavefronts interleave the 32-wide k-steps (M/16 blocks
// of one wave would leave the chip at ~1 wave per CU, latency-bound), fragments are loaded straight from global memory
// (X is streamed once, W is L2-resident) with up to 8 k-steps of loads in flight per wave, and the four partial
// 16x16 blocks are summed through LDS.
struct SkinnyPair {
  const bf16_t* X[2];
  const bf16_t* W[2];
  bf16_t* out[2];
};
__global__ __launch_bounds__(256) void skinny_xw_kernel(const SkinnyPair pr, int ldx, int ldw, int ldo, int M, int K, int R) {
  const bf16_t* __restrict__ X = pr.X[blockIdx.y];
  const bf16_t* __restrict__ W = pr.W[blockIdx.y];
  bf16_t* __restrict__ out = pr.out[blockIdx.y];
  __shared__ float part[3][64][4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r16 = lane & 15, kq = lane >> 4;
  const int row = blockIdx.x * 16 + r16;
  const bool rok = row < M, wok = r16 < R;
  const bf16_t* xp = X + (size_t)(rok ? row : 0) * ldx + kq * 8;
  const bf16_t* wp = W + (size_t)(wok ? r16 : 0) * ldw + kq * 8;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  const uint4 zero = make_uint4(0, 0, 0, 0);
  const unsigned xmask = rok ? 0xffffffffu : 0u, wmask = wok ? 0xffffffffu : 0u;
  // This wave's k-steps come in PAIRS that share a 128-byte line of every row (steps 2j and 2j + 1 of line j = wave, wave + 4, ...): a
  // wave's load instruction covers 64 bytes of 16 rows, so with the former interleave (step = wave, wave + 4, ...) every line of X was
  // fetched by two different waves, half each (round 4: 12.5 -> see profiles/r04_small_kernels.txt)
  int k = wave * 64;
  constexpr int U = 6;                     // k-steps in flight per wave (three lines)
  for (; k + (U / 2 - 1) * 256 + 64 <= K; k += (U / 2) * 256) {
    // unconditional loads (the pointers of rows / adapter columns outside the problem were clamped to row 0 above), masked
    // afterwards: `rok ? load : zero` makes hipcc branch around each load and wait for it (three round trips per six k-steps)
    uint4 xa[U], wb[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      xa[u] = *(const uint4*)(xp + k + 256 * (u >> 1) + 32 * (u & 1));
      wb[u] = *(const uint4*)(wp + k + 256 * (u >> 1) + 32 * (u & 1));
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      xa[u].x &= xmask, xa[u].y &= xmask, xa[u].z &= xmask, xa[u].w &= xmask;
      wb[u].x &= wmask, wb[u].y &= wmask, wb[u].z &= wmask, wb[u].w &= wmask;
      acc = mvit_mfma16(*(bf16x8*)&xa[u], *(bf16x8*)&wb[u], acc, 0, 0, 0);
    }
  }
  for (; k < K; k += 256) {
#pragma unroll
    for (int h2 = 0; h2 < 2; ++h2) {
      const int kk = k + 32 * h2;
      const bool kok = kk + kq * 8 < K;
      const uint4 xa = (rok && kok) ? *(const uint4*)(xp + kk) : zero;
      const uint4 wb = (wok && kok) ? *(const uint4*)(wp + kk) : zero;
      acc = mvit_mfma16(*(const bf16x8*)&xa, *(const bf16x8*)&wb, acc, 0, 0, 0);
    }
  }
  if (wave > 0) {
#pragma unroll
    for (int j = 0; j < 4; ++j) part[wave - 1][lane][j] = acc[j];
  }
  __syncthreads();
  // C/D layout of the 16x16 MFMA: col = lane&15, row = (lane>>4)*4 + reg
  if (wave == 0 && r16 < R) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int orow = blockIdx.x * 16 + kq * 4 + j;
      const float v = acc[j] + part[0][lane][j] + part[1][lane][j] + part[2][lane][j];
      if (orow < M) out[(size_t)orow * ldo + r16] = f2bf(v);
    }
  }
}

// ------------------------------------------------------------------ prefix tokens (cls + registers) of the token matrix
__global__ __launch_bounds__(256) void prefix_tokens_kernel(float* __restrict__ x, const float* __restrict__ cls,
                                                            const float* __restrict__ reg, int B, int ntok, int D, int R) {
  const int total = B * (1 + R) * D;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
    const int d = i % D, t = (i / D) % (1 + R), b = i / (D * (1 + R));
    x[((size_t)b * ntok + t) * D + d] = t == 0 ? cls[d] : reg[(size_t)(t - 1) * D + d];
  }
}

__global__ __launch_bounds__(256) void cast_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst, long long n) {
  for (long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += (long long)gridDim.x * 1024) {
    if (i + 3 < n) {
      const float4 v = *(const float4*)(src + i);
      uint2 r;
      r.x = pack2bf(v.x, v.y);
      r.y = pack2bf(v.z, v.w);
      *(uint2*)(dst + i) = r;
    } else {
      for (long long j = i; j < n; ++j) dst[j] = f2bf(src[j]);
    }
  }
}

__global__ __launch_bounds__(256) void scale_cols_cast_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                              bf16_t* __restrict__ out, int M, int D,
                                                              const float* __restrict__ rowscale) {
  const long long nv = (long long)M * D / 4;
  const int dv = D >> 2;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < nv; i += (long long)gridDim.x * 256) {
    const float4 v = ((const float4*)x)[i];
    const float4 g = ((const float4*)gamma)[i % dv];
    const float rs = rowscale ? rowscale[i / dv] : 1.f;
    uint2 r;
    r.x = pack2bf(v.x * g.x * rs, v.y * g.y * rs);
    r.y = pack2bf(v.z * g.z * rs, v.w * g.w * rs);
    ((uint2*)out)[i] = r;
  }
}

inline int nblocks(long long work, int per_block, int cap = 8192) {
  long long b = (work + per_block - 1) / per_block;
  return (int)(b < 1 ? 1 : (b > cap ? cap : b));
}

}  // namespace

extern "C" {

MVIT_API int mvit_layernorm_fwd(const float* x, const float* w, const float* b, void* out, int M, int D, float eps,
                                mvit_stream_t stream) {
  MVIT_CLEAR_ERROR();
  if (M <= 0 || D <= 0 || (D & 3) || D > 256 * LN_MAXV) return MVIT_EINVAL;
  auto launch = [&](auto kern) {
    hipLaunchKernelGGL(kern, dim3((M + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, w, b, (bf16_t*)out, M, D, eps);
  };
  if (D == 1536) launch(ln_fwd_kernel<6>);
  else if (D == 1024) launch(ln_fwd_kernel<4>);
  else if (D == 2048) launch(ln_fwd_kernel<8>);
  else if (D == 512) launch(ln_fwd_kernel<2>);
  else launch(ln_fwd_kernel<0>);
  return MVIT_LAUNCH_CHECK();
}

MVIT_API int mvit_layernorm_lora_fwd(const float* x, const float* w, const float* b, void* out, const void* AcatT, void* t,
                                     int M, int D, float eps, int R2, mvit_stream_t stream) {
  MVIT_CLEAR_ERROR();
  if (M <= 0 || D <= 0 || (D & 31) || D > 256 * LN_MAXV || R2 <= 0 || R2 > 16 || !AcatT || !t) return MVIT_EINVAL;
  // two 16-row groups per block on a balanced grid (ln_fwd_lora2_kernel) where the 48 row images fit the LDS
  const size_t lds2 = 48 * ((size_t)D * 2 + 16) + 8 * 64 * 4 * sizeof(float);
#ifndef MVIT_LNL_TWO_GROUPS
#define MVIT_LNL_TWO_GROUPS 1
#endif
  if (MVIT_LNL_TWO_GROUPS && lds2 <= 160 * 1024) {
    int blocks = (M + 15) / 16;
    if (blocks > mvit_num_cus()) blocks = mvit_num_cus();
    const int rpb = (M + blocks - 1) / blocks;
    blocks = (M + rpb - 1) / rpb;
    auto launch2 = [&](auto kern, mvit_per_device_size& raised) {
      if (mvit_ensure_dynamic_lds((const void*)kern, lds2, raised) != MVIT_OK) return (int)MVIT_EINVAL;
      hipLaunchKernelGGL(kern, dim3(blocks), dim3(1024), lds2, (hipStream_t)stream, x, w, b, (bf16_t*)out, (const bf16_t*)AcatT,
                         (bf16_t*)t, M, D, eps, R2, rpb);
      return MVIT_LAUNCH_CHECK();
    };
    static mvit_per_device_size g2, g4, g6, h2, h4, h6;
    if (D == 1536) return launch2(ln_fwd_lora2_kernel<6, true>, h6);
    if (D == 1024) return launch2(ln_fwd_lora2_kernel<4, true>, h4);
    if (D == 512) return launch2(ln_fwd_lora2_kernel<2, true>, h2);
    if (D <= 512) return launch2(ln_fwd_lora2_kernel<2>, g2);
    if (D <= 1024) return launch2(ln_fwd_lora2_kernel<4>, g4);
    return launch2(ln_fwd_lora2_kernel<6>, g6);
  }
  const size_t lds = (16 + LNL_ROWS) * ((size_t)D * 2 + 16) + 3 * 64 * 4 * sizeof(float);
  auto launch = [&](auto kern, mvit_per_device_size& raised) {
    if (mvit_ensure_dynamic_lds((const void*)kern, lds, raised) != MVIT_OK) return (int)MVIT_EINVAL;
    hipLaunchKernelGGL(kern, dim3((M + LNL_ROWS - 1) / LNL_ROWS), dim3(64 * LNL_ROWS), lds, (hipStream_t)stream, x, w, b,
                       (bf16_t*)out, (const bf16_t*)AcatT, (bf16_t*)t, M, D, eps, R2);
    return MVIT_LAUNCH_CHECK();
  };
  static mvit_per_device_size r2, r4, r6, r8, f2, f4, f6, f8;
  if (D == 1536) return launch(ln_fwd_lora_kernel<6, true>, f6);
  if (D == 1024) return launch(ln_fwd_lora_kernel<4, true>, f4);
  if (D == 2048) return launch(ln_fwd_lora_kernel<8, true>, f8);
  if (D == 512) return launch(ln_fwd_lora_kernel<2, true>, f2);
  if (D <= 512) return launch(ln_fwd_lora_kernel<2>, r2);
  if (D <= 1024) return launch(ln_fwd_lora_kernel<4>, r4);
  if (D <= 1536) return launch(ln_fwd_lora_kernel<6>, r6);
  return launch(ln_fwd_lora_kernel<8>, r8);
}

MVIT_API int mvit_layernorm_bwd(const void* dh, const float* x, const float* w, float* dx, const float* gamma_next,
                                void* dy, int M, int D, float eps, int accumulate, const float* rowscale_next,
                                mvit_stream_t stream) {
  MVIT_CLEAR_ERROR();
  if (M <= 0 || D <= 0 || (D & 3) || D > 256 * LN_MAXV) return MVIT_EINVAL;
  if ((dy != nullptr) != (gamma_next != nullptr)) return MVIT_EINVAL;
  auto launch = [&](auto kern) {
    hipLaunchKernelGGL(kern, dim3((M + 3) / 4), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)dh, x, w, dx, gamma_next,
                       (bf16_t*)dy, M, D, eps, accumulate, rowscale_next);
    return MVIT_LAUNCH_CHECK();
  };
  if (D == 1536) return launch(ln_bwd_kernel<6, true>);
  if (D == 1024) return launch(ln_bwd_kernel<4, true>);
  if (D == 2048) return launch(ln_bwd_kernel<8, true>);
  if (D == 512) return launch(ln_bwd_kernel<2, true>);
  if (D <= 512) return launch(ln_bwd_kernel<2>);
  if (D <= 1024) return launch(ln_bwd_kernel<4>);
  if (D <= 1536) return launch(ln_bwd_kernel<6>);
  return launch(ln_bwd_kernel<8>);
}

MVIT_API int mvit_skinny_xw(const void* X, int ldx, const void* W, int ldw, void* out, int ldo, int M, int K, int R,
                            mvit_stream_t stream) {
  MVIT_CLEAR_ERROR();
  if (M <= 0 || K <= 0 || R <= 0 || R > 16 || (K & 7) || (ldx & 7) || (ldw & 7)) return MVIT_EINVAL;
  SkinnyPair pr{{(const bf16_t*)X, nullptr}, {(const bf16_t*)W, nullptr}, {(bf16_t*)out, nullptr}};
  hipLaunchKernelGGL(skinny_xw_kernel, dim3((M + 15) / 16, 1), dim3(256), 0, (hipStream_t)stream, pr, ldx, ldw, ldo, M, K, R);
  return MVIT_LAUNCH_CHECK();
}

MVIT_API int mvit_skinny_xw2(const void* X0, const void* W0, void* out0, const void* X1, const void* W1, void* out1, int ldx,
                             int ldw, int ldo, int M, int K, int R, mvit_stream_t stream) {
  MVIT_CLEAR_ERROR();
  if (M <= 0 || K <= 0 || R <= 0 || R > 16 || (K & 7) || (ldx & 7) || (ldw & 7) || !X1 || !W1 || !out1) return MVIT_EINVAL;
  SkinnyPair pr{{(const bf16_t*)X0, (const bf16_t*)X1}, {(const bf16_t*)W0, (const bf16_t*)W1}, {(bf16_t*)out0, (bf16_t*)out1}};
  hipLaunchKernelGGL(skinny_xw_kernel, dim3((M + 15) / 16, 2), dim3(256), 0, (hipStream_t)stream, pr, ldx, ldw, ldo, M, K, R);
  return MVIT_LAUNCH_CHECK();
}

MVIT_API int mvit_prefix_tokens(float* x, const float* cls, const float* reg, int B, int ntok, int D, int R,
                                mvit_stream_t stream) {
  MVIT_CLEAR_ERROR();
  if (B <= 0 || D <= 0 || R < 0 || ntok < 1 + R) return MVIT_EINVAL;
  hipLaunchKernelGGL(prefix_tokens_kernel, dim3(nblocks((long long)B * (1 + R) * D, 256)), dim3(256), 0,
                     (hipStream_t)stream, x, cls, reg, B, ntok, D, R);
  return MVIT_LAUNCH_CHECK();
}

MVIT_API int mvit_cast_f32_bf16(const float* src, void* dst, long long n, mvit_stream_t stream) {
  MVIT_CLEAR_ERROR();
  if (n <= 0) return MVIT_EINVAL;
  hipLaunchKernelGGL(cast_kernel, dim3(nblocks(n, 1024)), dim3(256), 0, (hipStream_t)stream, src, (bf16_t*)dst, n);
  return MVIT_LAUNCH_CHECK();
}

MVIT_API int mvit_scale_cols_cast(const float* x, const float* gamma, void* out, int M, int D, const float* rowscale,
                                  mvit_stream_t stream) {
  MVIT_CLEAR_ERROR();
  if (M <= 0 || D <= 0 || (D & 3)) return MVIT_EINVAL;
  hipLaunchKernelGGL(scale_cols_cast_kernel, dim3(nblocks((long long)M * D / 4, 256)), dim3(256), 0, (hipStream_t)stream, x,
                     gamma, (bf16_t*)out, M, D, rowscale);
  return MVIT_LAUNCH_CHECK();
}

}  // extern "C"
