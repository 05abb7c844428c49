// "TN" MFMA GEMM for weight gradients:  C[i,j] += sum_m A[m,i] * B[m,j]   (f32 atomic accumulate)
//
// Both operands are stored m-major (activations / gradients as produced by the forward and backward passes), so the
// contraction index is the ROW index of both.  Tiles [64 m][64 cols] are DMA'd into LDS as they lie in memory and the
// MFMA fragments (8 consecutive m per lane for a fixed column) are gathered with the gfx950 transpose read
// ds_read_b64_tr_b16 -- no transposed copies of the operands, no im2col buffer:
//   * decoder conv weight gradients: A = virtual im2col of the NHWC input (3x3 window gather, as the forward loader),
//     B = dY [pixels, Cout]   ->  dW^T [(ky,kx,c), Cout]
//   * LoRA dA = dt^T h, dB = t^T dq and the head conv dW3 = E^T x (A dense).
// 4 waves, tile 64(i) x 128(j) or 128(i) x 32(j), 64 rows of m per step, 2 LDS stages, split over m (blockIdx.z).
#include "common.hpp"
#include "../../include/miphei_hip.h"

namespace {

typedef short v4s __attribute__((ext_vector_type(4)));
constexpr int BLK_BYTES = 64 * 128;  // one [64 m][64 col] bf16 block

__device__ __forceinline__ int swz(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }

__device__ __forceinline__ v4s tr_read4(const char* tile, int row_base, int col_base16, int lane) {
  const int i = lane & 15;
  const int row = row_base + (i >> 2);
  const int col = col_base16 + 4 * (i & 3);
  const char* p = tile + swz(row, col >> 3) + ((col & 7) << 1);
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s*)p);
}
__device__ __forceinline__ bf16x8 join(v4s a, v4s b) {
  union { struct { v4s lo, hi; } s; bf16x8 v; } u;
  u.s.lo = a;
  u.s.hi = b;
  return u.v;
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* ptr) {
  const unsigned long long v = (unsigned long long)ptr;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
  return __builtin_amdgcn_make_buffer_rsrc((void*)(((unsigned long long)hi << 32) | lo), 0, 0x7fffffff, 0x00020000);
}

// TR: the accumulator tiles are computed transposed (MFMA operands swapped), so that the 32 lanes of an output instruction run along i --
// for outputs whose i index is the contiguous one (ldci == 1: LoRA dA, stored [D][r]) an atomic instruction then touches one or two
// 128-byte lines instead of 32 (the products are split over m: the atomics are what such a launch waits for)
// NST: LDS stages.  2 = one step of DMA in flight per block, three blocks per CU.  3 (-DMVIT_TN_STAGES=3, batched LoRA launches only) keeps two
// steps in flight but only two blocks per CU fit: measured SLOWER (dB 68 -> 74 us, dA 51 -> 57 us per launch of 10 blocks), kept at 2
template <int IT, int JT, int AMODE, bool TR = false, int NST = 2>
__global__ __launch_bounds__(256) void gemm_tn_kernel(const mvit_gemm_tn_args p) {
  constexpr int WI = IT / 32 >= 4 ? 4 : IT / 32;  // waves along i
  constexpr int WJ = 4 / WI;
  constexpr int WJT = JT / WJ;                     // columns per wave
  constexpr int TN = WJT / 32;
  constexpr int ABLK = IT / 64, BBLK = (JT + 63) / 64;
  constexpr int STAGE = (ABLK + BBLK) * BLK_BYTES;
  typedef __attribute__((address_space(3))) void* lds_ptr;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const int wave_i = wave / WJ, wave_j = wave % WJ;
  const int half = lane >> 5, l31 = lane & 31;
  // Block coordinates.  Round 5: the hardware places the workgroup of linear index L (x fastest, then y, then z) on XCD L % 8, and the
  // gx * gy blocks of one (product, m slice) read the SAME rows of both operands -- as hardware coordinates they sat on gx * gy different
  // XCDs (the conv weight gradients: 164 MB of fabric traffic per launch against 38 MB of operands, profiles/r04_pmc_traffic.json).
  // The XCD-contiguous renumbering of gemm_ws.hip's TileOrder puts them on one XCD (x fastest inside a slice).
  const int gx = gridDim.x, gy = gridDim.y, gxy = gx * gy, nblk = gxy * (int)gridDim.z;
  const int Lb = (int)blockIdx.x + gx * ((int)blockIdx.y + gy * (int)blockIdx.z);
  const int xq_ = nblk >> 3, xr_ = nblk & 7, xcd_ = Lb & 7;
  const int wg_ = (xcd_ < xr_ ? xcd_ * (xq_ + 1) : xr_ * (xq_ + 1) + (xcd_ - xr_) * xq_) + (Lb >> 3);
  const int bz = wg_ / gxy, rem_ = wg_ - bz * gxy, by = rem_ / gx, bx = rem_ - by * gx;
  const int i0 = bx * IT, j0 = by * JT;
  const long long nsteps = ((long long)p.M + 63) / 64;
  const int nb = p.batch > 1 ? p.batch : 1;
  const int nz = gridDim.z / nb, bi = bz / nz, zi = bz - bi * nz;   // (product, slice of m)
  const long long per = (nsteps + nz - 1) / nz;
  const long long s_begin = zi * per, s_end = min(nsteps, s_begin + per);
  if (s_begin >= s_end) return;
  if (p.C2 && p.jlo2 > 0 && j0 >= p.j1 && j0 + JT <= p.jlo2) return;  // column block owned by neither output
  const bf16_t* Ap = (const bf16_t*)p.A + (size_t)bi * p.strideA;
  const bf16_t* Bp = (const bf16_t*)p.B + (size_t)bi * p.strideB;
  const __amdgpu_buffer_rsrc_t rsA = make_rsrc(AMODE == MVIT_A_DENSE ? (const void*)(Ap + (size_t)s_begin * 64 * p.lda) : (const void*)Ap);
  const __amdgpu_buffer_rsrc_t rsB = make_rsrc(Bp + (size_t)s_begin * 64 * p.ldb);

  auto issue = [&](long long st, int buf) {
    char* a = smem + buf * STAGE;
    char* b = a + ABLK * BLK_BYTES;
    const int mrel0 = (int)((st - s_begin) * 64);  // row offset relative to the descriptor base
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int row = 8 * wave_u + 32 * j + (lane >> 3);
      const int c = (lane & 7) ^ ((row >> 1) & 7);
      const long long m = st * 64 + row;
      const bool mok = m < p.M;
      // ---- A blocks
      int pb = 0, py = 0, px = 0;
      if (AMODE != MVIT_A_DENSE) {
        const long long mm = mok ? m : 0;
        px = (int)(mm % p.conv_OW);
        const long long t = mm / p.conv_OW;
        py = (int)(t % p.conv_OH);
        pb = (int)(t / p.conv_OH);
      }
#pragma unroll
      for (int ab = 0; ab < ABLK; ++ab) {
        const int icol = i0 + ab * 64 + c * 8;
        unsigned off = 0x80000000u;
        if (mok && icol < p.I) {
          if (AMODE == MVIT_A_DENSE) {
            off = ((unsigned)(mrel0 + row) * (unsigned)p.lda + (unsigned)icol) * 2u;
          } else {
            const int tap = icol / p.conv_C, ch = icol - tap * p.conv_C;
            const int ky = tap / 3, kx = tap - ky * 3;
            const int iy = py * p.conv_stride + ky - 1, ix = px * p.conv_stride + kx - 1;
            if (iy >= 0 && iy < p.conv_H && ix >= 0 && ix < p.conv_W)
              off = (((unsigned)(pb * p.conv_H + iy) * (unsigned)p.conv_W + (unsigned)ix) * (unsigned)p.conv_ld + (unsigned)ch) * 2u;
          }
        }
        asm volatile("" : "+v"(off));
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_ptr)(a + ab * BLK_BYTES + (8 * wave_u + 32 * j) * 128), 16, off, 0, 0, 0);
      }
#pragma unroll
      for (int bb = 0; bb < BBLK; ++bb) {
        const int jcol = j0 + bb * 64 + c * 8;
        unsigned off = (mok && jcol < p.J) ? ((unsigned)(mrel0 + row) * (unsigned)p.ldb + (unsigned)jcol) * 2u : 0x80000000u;
        asm volatile("" : "+v"(off));
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lds_ptr)(b + bb * BLK_BYTES + (8 * wave_u + 32 * j) * 128), 16, off, 0, 0, 0);
      }
    }
  };

  f32x16 acc[TN];
#pragma unroll
  for (int t = 0; t < TN; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  // fragment addressing: this lane's column inside its 64-column block
  const int il = wave_i * 32;                       // first i of the wave inside the tile
  const char* const a_off = (const char*)0 + (il / 64) * BLK_BYTES;
  const int a_cb = (il % 64) + 16 * ((lane >> 4) & 1);

  // Stage st + NST - 1 is requested at the top of step st (behind the barrier that says every wave is done with the stage it
  // overwrites); a step waits for its own stage with a counted vmcnt -- the younger stages stay in flight -- and for this wave's LDS
  // reads of the previous step (lgkmcnt: s_barrier does not), then meets the other waves.
  constexpr int DMAS = 2 * (ABLK + BBLK);              // DMA instructions per wave and stage
#pragma unroll
  for (int i = 0; i < NST - 1; ++i)
    if (s_begin + i < s_end) issue(s_begin + i, i);
  int cur = 0;
  for (long long st = s_begin; st < s_end; ++st) {
    if (NST == 3 && st + 1 < s_end)
      asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(DMAS) : "memory");
    else
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (st + NST - 1 < s_end) issue(st + NST - 1, cur == 0 ? NST - 1 : cur - 1);
    const char* a = smem + cur * STAGE + (size_t)a_off;
    const char* b = smem + cur * STAGE + ABLK * BLK_BYTES;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int mb = 16 * s + 4 * half;
      const bf16x8 fa = join(tr_read4(a, mb, a_cb, lane), tr_read4(a, mb + 8, a_cb, lane));
#pragma unroll
      for (int t = 0; t < TN; ++t) {
        const int jl = wave_j * WJT + t * 32;
        const char* bt = b + (jl / 64) * BLK_BYTES;
        const int b_cb = (jl % 64) + 16 * ((lane >> 4) & 1);
        const bf16x8 fb = join(tr_read4(bt, mb, b_cb, lane), tr_read4(bt, mb + 8, b_cb, lane));
        acc[t] = TR ? mvit_mfma32(fb, fa, acc[t], 0, 0, 0)
                    : mvit_mfma32(fa, fb, acc[t], 0, 0, 0);
      }
    }
    cur = cur + 1 == NST ? 0 : cur + 1;
  }
  // D[i][j]: col j = lane&31, row i = (r&3) + 8*(r>>2) + 4*half   (TR: col i = lane&31, row j = ...)
  // split_stride > 0: slice zi of the m range adds into its own copy of the output (C + zi * split_stride) -- every element
  // then has exactly one contributing block and the caller sums the copies in a fixed order (run-to-run identical results)
  float* C = (float*)p.C + (size_t)bi * p.strideC + (size_t)zi * p.split_stride;
  float* C2 = p.C2 ? p.C2 + (size_t)bi * p.strideC + (size_t)zi * p.split_stride : nullptr;
#pragma unroll
  for (int t = 0; t < TN; ++t) {
    const int jb = j0 + wave_j * WJT + t * 32, ib = i0 + il;
    if (!TR && jb + l31 >= p.J) continue;
    if (TR && ib + l31 >= p.I) continue;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int rr = (r & 3) + 8 * (r >> 2) + 4 * half;
      const int i = TR ? ib + l31 : ib + rr;
      const int j = TR ? jb + rr : jb + l31;
      if (TR ? j >= p.J : i >= p.I) continue;
      if (!p.C2) {
        atomicAdd(C + (size_t)i * p.ldci + (size_t)j * p.ldcj, acc[t][r]);
      } else if (i < p.isplit) {   // two outputs from one pass (see mvit_gemm_tn_args)
        if (p.jlo2 == 0 || j < p.j1) atomicAdd(C + (size_t)i * p.ldci + (size_t)j * p.ldcj, acc[t][r]);
      } else if (j >= p.jlo2) {
        atomicAdd(C2 + (size_t)(i - p.isplit) * p.ldci + (size_t)(j - p.jlo2) * p.ldcj, acc[t][r]);
      }
    }
  }
}

template <int IT, int JT>
int launch(const mvit_gemm_tn_args& a, hipStream_t s) {
  const long long nsteps = ((long long)a.M + 63) / 64;
  int split = a.msplit > 0 ? a.msplit : 1;
  if (split > nsteps) split = (int)nsteps;
  const int nb = a.batch > 1 ? a.batch : 1;
  if ((long long)split * nb > 65535) return MVIT_EINVAL;
  dim3 grid((a.I + IT - 1) / IT, (a.J + JT - 1) / JT, split * nb);
  const size_t lds2 = 2 * (size_t)(IT / 64 + (JT + 63) / 64) * BLK_BYTES;
  int rc = MVIT_OK;
  auto go = [&](auto kern, size_t lds, mvit_per_device_size& raised) {
    rc = mvit_ensure_dynamic_lds((const void*)kern, lds, raised);
    if (rc == MVIT_OK) hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, a);
  };
#ifndef MVIT_TN_STAGES
#define MVIT_TN_STAGES 2
#endif
  constexpr int NSL = (IT == 64 && JT == 128) ? MVIT_TN_STAGES : 2;     // stages of the batched LoRA launches
  if (a.amode == MVIT_A_DENSE && a.batch > 1 && NSL == 3) {
    static mvit_per_device_size r0, r1;                 // (per kernel: the attribute belongs to the function)
    if (a.ldci == 1 && a.ldcj > 1)
      go(gemm_tn_kernel<IT, JT, MVIT_A_DENSE, true, NSL>, lds2 / 2 * 3, r0);
    else
      go(gemm_tn_kernel<IT, JT, MVIT_A_DENSE, false, NSL>, lds2 / 2 * 3, r1);
  } else if (a.amode == MVIT_A_DENSE && a.ldci == 1 && a.ldcj > 1 && IT == 64 && JT == 128) {
    static mvit_per_device_size r2;
    go(gemm_tn_kernel<IT, JT, MVIT_A_DENSE, true>, lds2, r2);
  } else if (a.amode == MVIT_A_DENSE) {
    static mvit_per_device_size r3;
    go(gemm_tn_kernel<IT, JT, MVIT_A_DENSE>, lds2, r3);
  } else {
    static mvit_per_device_size r4;
    go(gemm_tn_kernel<IT, JT, MVIT_A_CONV3>, lds2, r4);
  }
  if (rc != MVIT_OK) return rc;
  return MVIT_LAUNCH_CHECK();
}

}  // namespace

extern "C" MVIT_API int mvit_gemm_tn_bf16(const mvit_gemm_tn_args* args, mvit_stream_t stream) {
  MVIT_CLEAR_ERROR();
  if (!args) return MVIT_EINVAL;
  const mvit_gemm_tn_args& a = *args;
  if (a.M <= 0 || a.I <= 0 || a.J <= 0 || (a.I & 7) || (a.J & 7) || (a.ldb & 7)) return MVIT_EINVAL;
  if (a.C2 && (a.isplit <= 0 || a.isplit >= a.I || a.jlo2 < 0 || (a.jlo2 > 0 && (a.j1 <= 0 || a.j1 > a.jlo2)))) return MVIT_EINVAL;
  if (a.split_stride < 0) return MVIT_EINVAL;
  if (a.batch < 0 || (a.batch > 1 && (a.amode != MVIT_A_DENSE || ((a.strideA | a.strideB) & 7)))) return MVIT_EINVAL;
  if (a.amode == MVIT_A_DENSE) {
    if (a.lda & 7) return MVIT_EINVAL;
  } else if (a.amode == MVIT_A_CONV3) {
    if ((a.conv_C & 7) || (a.conv_ld & 7) || a.I != 9 * a.conv_C) return MVIT_EINVAL;
  } else {
    return MVIT_EINVAL;
  }
  hipStream_t s = (hipStream_t)stream;
  // tile choice: narrow outputs keep JT at the padded J (a 128-wide tile on J = 64 would spend half its DMA and MFMA work
  // on zero columns); tall outputs take 128 rows (fewer operand bytes per flop); the LoRA products (I = 16) stay on 64 rows
  if (a.J <= 32) return launch<128, 32>(a, s);
  if (a.J <= 64) return a.I >= 128 ? launch<128, 64>(a, s) : launch<64, 128>(a, s);
  return a.I >= 512 ? launch<128, 128>(a, s) : launch<64, 128>(a, s);
}
