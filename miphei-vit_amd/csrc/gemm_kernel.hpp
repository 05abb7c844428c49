// bf16 MFMA GEMM / implicit-GEMM 3x3 convolution for gfx950 (MI355X).
//
//   C[M,N] = A[M,K] * B[N,K]^T (+ A2 * B2^T)      f32 accumulate on v_mfma_f32_32x32x16_bf16
//
// Structure
//  * 4 or 8 wavefronts (64 lanes), each owning a 64x64 / 32x32 / 64x32 sub-tile of a BMxBN block tile, BK = 64.
//  * Operands go global -> LDS by DMA (buffer_load ... lds, 16 B per lane, no VGPR round trip).  LDS rows are 128 B
//    (64 bf16); the 16-byte chunk index is XOR-swizzled with (row>>1)&7 ON THE SOURCE ADDRESS so every ds_read_b128
//    lane group touches 16 distinct slots (conflict-free).  M / K tails and the zero padding of the 3x3 window come
//    from out-of-range offsets (hardware returns 0).
//  * NSTAGE LDS buffers with NSTAGE-1 K tiles of DMA in flight: counted s_waitcnt vmcnt + raw s_barrier, never a
//    __syncthreads() inside the K loop (it would drain the DMA queue).
//  * Persistent blocks: a block walks tiles blockIdx.x, +gridDim.x, ...; before running the epilogue of tile i it has
//    already issued the first NSTAGE-1 K tiles of tile i+1, so launch, descriptor setup and first-tile DMA latency
//    are paid once per block, not once per tile.  Tile order is XCD-aware and grouped (GROUP_M tile rows x all
//    columns) so the ~32 blocks sharing one XCD's 4 MiB L2 reuse each other's A / B panels.
//  * Epilogue: 16-row accumulator slabs go through a wave-private f32 LDS panel (inside the LDS buffer consumed
//    last) and leave as 16-byte stores (8 consecutive columns per lane).
// The A operand loader is dense or a 3x3 window gather from an NHWC activation (implicit-GEMM convolution forward
// and adjoint), so the decoder convolutions run on the same mainloop.
#pragma once
#include <type_traits>
#include "common.hpp"
#include "../../include/miphei_hip.h"

namespace mvit_gemm {

constexpr int BK = 64;

struct ConvRow {
  int b, y, x;
  bool ok;
};

// LDS stages: the 8-wave 256x128 tile runs one block per CU and keeps two K tiles of DMA in flight
constexpr int gemm_stages(int bm, int bn) { return (bm + bn) * 128 * 3 <= 160 * 1024 && bm >= 256 ? 3 : 2; }

template <int BM, int BN, int WAVES_M, int WAVES_N>
constexpr size_t gemm_lds_bytes() {
  return (size_t)gemm_stages(BM, BN) * (BM + BN) * 128;
}

#ifndef MVIT_GEMM_TRANS   // 1: dense tiles accumulate C^T (MFMA operands swapped) and store row-per-lane, no LDS panel
#define MVIT_GEMM_TRANS 1
#endif
#ifndef MVIT_GEMM_LATE_CONST   // 1: measurement build, epilogue column constants fetched at the head of the epilogue (the former order)
#define MVIT_GEMM_LATE_CONST 0
#endif
#ifndef MVIT_GEMM_LATE_PROLOGUE   // 1: measurement build, the next tile's prologue DMA issued behind the first epilogue slab
#define MVIT_GEMM_LATE_PROLOGUE 0
#endif
#ifndef MVIT_GEMM_MI16    // bit 0: the 8-wave 256x128 tile runs on v_mfma_f32_16x16x32_bf16 (two sub-steps per K tile) instead of 32x32x16;
#define MVIT_GEMM_MI16 1  // measurement builds: bit 1 the 8-wave 256x256 tile too, bit 2 the 4-wave 256-row tiles too
#endif

template <int BM, int BN, int WAVES_M, int WAVES_N, int AMODE, int EPI>
__global__ __launch_bounds__(64 * WAVES_M * WAVES_N) void gemm_kernel(const mvit_gemm_args p) {
  constexpr int NW = WAVES_M * WAVES_N;
  constexpr int NT = 64 * NW;                 // threads
  constexpr int RPI = NT / 8;                 // tile rows filled per DMA instruction of the block
  constexpr int NSTAGE = gemm_stages(BM, BN);
  constexpr int WTM = BM / WAVES_M, WTN = BN / WAVES_N;
  // MFMA shape.  v_mfma_f32_16x16x32_bf16 does the same flops per cycle as 32x32x16 with a quarter of the accumulator traffic per
  // instruction: on random operands at the socket power cap a bare loop of it sustains 2.03 PFLOP/s against 1.82
  // (tools/probes/mfma_power.hip), and the dense loops here are power-limited.  Measured in the training step,
  // same box: 8-wave 256x128 tile +3.9 % (the product choice); 8-wave 256x256 tile -0.4 % (fc1 + SwiGLU 153.5 vs 145.2 us; batch-64
  // inference within noise, embedding extraction -1.6 %); 4-wave 256x128 tile -11 %.
  constexpr bool MI16 = BM == 256 && (((MVIT_GEMM_MI16 & 1) && WAVES_M * WAVES_N == 8 && BN == 128) ||
                                      ((MVIT_GEMM_MI16 & 2) && WAVES_M * WAVES_N == 8 && BN == 256) ||
                                      ((MVIT_GEMM_MI16 & 4) && WAVES_M * WAVES_N == 4));
  constexpr int FR = MI16 ? 16 : 32;          // rows of an operand fragment = rows / columns of an accumulator block
  constexpr int NSUB = MI16 ? 2 : 4;          // MFMA sub-steps per K tile (K = 32 / 16 per instruction)
  constexpr int CPS = 8 / NSUB;               // 16-byte K chunks per sub-step
  constexpr int AR = MI16 ? 4 : 16;           // accumulator registers per block
  constexpr int FSTRIDE = FR * 128;           // LDS bytes between consecutive fragments of a wave's sub-tile
  using acc_t = std::conditional_t<MI16, f32x4, f32x16>;
  constexpr int TM = WTM / FR, TN = WTN / FR;
  // 128-row wave sub-tiles on the 16x16x32 shape: the A fragments of a sub-step are consumed in two groups of TM / 2, so that the
  // register-double-buffered fragments stay at 2 x (4 + 4) (with all 8 + 4 of a sub-step double-buffered the kernel spills)
  constexpr int AHALF = (MI16 && TM > 4) ? 2 : 1;
  constexpr int TMH = TM / AHALF;
  constexpr int A_CH = BM / RPI, B_CH = BN / RPI;
  constexpr int LPT = A_CH + B_CH;            // DMA instructions per thread per K tile
  constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, BUF_BYTES = A_BYTES + B_BYTES;
  constexpr int SLD = WTN + 4;                // epilogue panel row stride (floats)
  constexpr int SLAB = 16 * SLD;              // 16-row slab per wave
  // register-double-buffered fragment pipeline with an explicit instruction order: one wave per SIMD (4 waves, 128-row
  // sub-tiles) and the 8-wave 256-row tiles
  constexpr bool PIPE = (NW == 4 && WTM == 128) || (NW == 8 && BM == 256);
  // Transposed accumulation: acc[i][j] holds D^T (lane = row of C, registers = 4-column groups), so the epilogue needs no
  // transposition through LDS: every lane post-processes and stores pieces of its own row.
  // (Measured and dropped for the epilogues with per-element operands -- residual, SwiGLU, d(SwiGLU): a lane per row means 64
  // different cache lines per load / store instruction, and proj + residual went 39 -> 48 us, dfc2 + d(SwiGLU) 84 -> 100 us.)
  // (Not on the 16x16x32 tiles: a v_permlane16_swap version of the row-per-lane store -- 64-byte row segments per instruction -- was
  // measured at 81.2 vs 77.6 us for qkv and -0.25 % on the step against the LDS panel.)
  constexpr bool TRANS = MVIT_GEMM_TRANS && AMODE == MVIT_A_DENSE && EPI == MVIT_EPI_STORE && !MI16;
  static_assert(!MI16 || PIPE, "the 16x16x32 K step exists in the explicitly ordered pipeline only");
  static_assert(A_CH >= 1 && B_CH >= 1, "tile too small");
  static_assert((size_t)NW * SLAB * 4 + (size_t)WAVES_M * BN * 8 <= (size_t)BUF_BYTES, "epilogue panel must fit one LDS buffer");
  typedef __attribute__((address_space(3))) void* lds_ptr;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const int wave_m = wave / WAVES_N, wave_n = wave % WAVES_N;
  const int frag_row = lane & (FR - 1), frag_half = lane / FR;   // fragment row and K chunk (inside a sub-step) of this lane
  const int c8 = tid & 7;         // 16-byte slot inside the 64-wide K slice
  const int row_base = tid >> 3;  // 0..RPI-1, +RPI per chunk index
  const int c8s = c8 ^ ((row_base >> 1) & 7);  // source chunk for this lane's LDS slot (same for every +RPI row)
  constexpr unsigned OOB = 0x80000000u;

  const int tiles_m = (p.M + BM - 1) / BM, tiles_n = (p.N + BN - 1) / BN;
  const int ntiles = tiles_m * tiles_n;

  // ---- K range (split-K over blockIdx.z), identical for every tile
  const int nk1 = (p.K + BK - 1) / BK;
  const int nk2 = p.A2 ? (p.K2 + BK - 1) / BK : 0;
  const int nk = nk1 + nk2;
  int t_begin = 0, t_end = nk;
  if (p.ksplit > 1) {
    const int per = (nk + p.ksplit - 1) / p.ksplit;
    t_begin = blockIdx.z * per;
    t_end = min(nk, t_begin + per);
    if (t_begin >= t_end) return;
  }

  const bf16_t* __restrict__ Ap = (const bf16_t*)p.A;
  const bf16_t* __restrict__ Bp = (const bf16_t*)p.B;
  const bf16_t* __restrict__ A2p = (const bf16_t*)p.A2;
  const bf16_t* __restrict__ B2p = (const bf16_t*)p.B2;

  auto make_rsrc = [](const bf16_t* ptr) __attribute__((always_inline)) {
    // wave-uniform by construction (kernel arguments + blockIdx arithmetic); readfirstlane makes it provable so the
    // descriptor stays in SGPRs and hipcc does not wrap every buffer_load in a waterfall loop
    const unsigned long long v = (unsigned long long)ptr;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return __builtin_amdgcn_make_buffer_rsrc((void*)(((unsigned long long)hi << 32) | lo), 0, 0x7fffffff, 0x00020000);
  };

  // ---- per-tile context
  int m0 = 0, n0 = 0;
  __amdgpu_buffer_rsrc_t rsA, rsB, rsA2, rsB2;
  ConvRow crow[A_CH];
  unsigned baseA = 0, baseB = 0;  // dense operands: this lane's byte offset of chunk row 0 in K tile 0
  int validA = 0, validB = 0;     // bit j: chunk row j lies inside M / N

  auto setup_tile = [&](int vt) __attribute__((always_inline)) {
    // virtual tile id -> XCD-aware, grouped tile coordinates (bijective for any tile count):
    // block b runs on XCD b%8, so ids congruent mod 8 form one XCD's contiguous chunk of the grouped order
    int wg;
    {
      const int q = ntiles >> 3, r = ntiles & 7, xcd = vt & 7;
      wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (vt >> 3);
    }
    // tile rows per group of the tile order (256-row tiles): 8 since round 3 -- re-measured on the 16x16x32 kernels, same box, three
    // pairs of runs each: 8 vs 4 +0.45 % (451.5 vs 449.4 tiles/s), 16 vs 8 -0.1 %
#ifndef MVIT_GEMM_GROUP_M
#define MVIT_GEMM_GROUP_M 8
#endif
    constexpr int GROUP_M = BM >= 256 ? MVIT_GEMM_GROUP_M : 8;
    const int per_group = GROUP_M * tiles_n;
    const int first_m = (wg / per_group) * GROUP_M;
    const int gsz = min(tiles_m - first_m, GROUP_M);
    const int tile_m = first_m + (wg % per_group) % gsz, tile_n = (wg % per_group) / gsz;
    m0 = tile_m * BM;
    n0 = tile_n * BN;
    rsA = make_rsrc((AMODE == MVIT_A_DENSE) ? Ap + (size_t)m0 * p.lda : Ap);
    rsB = make_rsrc(Bp + (size_t)n0 * p.ldb);
    rsA2 = make_rsrc(A2p ? A2p + (size_t)m0 * p.lda2 : Ap);
    rsB2 = make_rsrc(B2p ? B2p + (size_t)n0 * p.ldb2 : Bp);
    baseA = ((unsigned)row_base * (unsigned)p.lda + (unsigned)c8s * 8u) * 2u;
    baseB = ((unsigned)row_base * (unsigned)p.ldb + (unsigned)c8s * 8u) * 2u;
    validA = validB = 0;
#pragma unroll
    for (int j = 0; j < A_CH; ++j) validA |= (m0 + row_base + RPI * j < p.M) ? (1 << j) : 0;
#pragma unroll
    for (int j = 0; j < B_CH; ++j) validB |= (n0 + row_base + RPI * j < p.N) ? (1 << j) : 0;
    if (AMODE != MVIT_A_DENSE) {
#pragma unroll
      for (int j = 0; j < A_CH; ++j) {
        const int gm = m0 + row_base + RPI * j;
        crow[j].ok = gm < p.M;
        const int g = crow[j].ok ? gm : 0;
        crow[j].x = g % p.conv_OW;
        const int t = g / p.conv_OW;
        crow[j].y = t % p.conv_OH;
        crow[j].b = t / p.conv_OH;
        if (AMODE == MVIT_A_PATCH) {   // row = (image, patch row, patch column): keep the window's first pixel
          crow[j].y *= p.conv_stride;
          crow[j].x *= p.conv_stride;
        }
      }
    }
  };

  // ---- operand staging: one K tile (A: BM x 64, B: BN x 64) by DMA into LDS buffer `buf`
  auto issue_tile_impl = [&](int t, int buf, auto ext_tag) __attribute__((always_inline)) {
    constexpr bool ext = decltype(ext_tag)::value;
    char* a = smem + buf * BUF_BYTES + wave_u * 8 * 128;
    char* b = a + A_BYTES;
    const int k0 = (ext ? t - nk1 : t) * BK + c8s * 8;
    const int klim = ext ? p.K2 : p.K;
    const bool kok = k0 < klim;
    if constexpr (AMODE == MVIT_A_DENSE || ext) {
      const int ld = ext ? p.lda2 : p.lda;
#pragma unroll
      for (int j = 0; j < A_CH; ++j) {
        const int r = row_base + RPI * j;
        const unsigned lin = ((unsigned)r * (unsigned)ld + (unsigned)k0) * 2u;
        unsigned off = (kok && m0 + r < p.M) ? lin : OOB;
        asm volatile("" : "+v"(off));  // keep ONE unconditional DMA per chunk (the vmcnt accounting counts them)
        if constexpr (ext)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA2, (lds_ptr)(a + j * RPI * 128), 16, off, 0, 0, 0);
        else
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_ptr)(a + j * RPI * 128), 16, off, 0, 0, 0);
      }
    } else if constexpr (AMODE == MVIT_A_PATCH) {
      // non-overlapping patch windows (patch-embed convolution, kernel = stride = conv_stride) of an NHWC bf16 image whose pixels
      // are one 16-byte piece (conv_ld = 8 channels, the unused ones zero): k = (dy * patch + dx) * 8 + c, so the piece of this
      // lane's K slot is pixel (dy, dx) of the window; rows of the image beyond the last whole patch are never addressed
      const int q = k0 >> 3;
      const int dy = q / p.conv_stride, dx = q - dy * p.conv_stride;
#pragma unroll
      for (int j = 0; j < A_CH; ++j) {
        const bool ok = kok && crow[j].ok;
        const unsigned lin =
            (((unsigned)(crow[j].b * p.conv_H + crow[j].y + dy) * (unsigned)p.conv_W + (unsigned)(crow[j].x + dx)) * (unsigned)p.conv_ld) * 2u;
        unsigned off = ok ? lin : OOB;
        asm volatile("" : "+v"(off));
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_ptr)(a + j * RPI * 128), 16, off, 0, 0, 0);
      }
    } else {
      const int tap = k0 / p.conv_C, ch = k0 - tap * p.conv_C;
      const int ky = tap / 3, kx = tap - ky * 3;
#pragma unroll
      for (int j = 0; j < A_CH; ++j) {
        int iy, ix;
        bool ok = kok && crow[j].ok;
        if (AMODE == MVIT_A_CONV3) {
          iy = crow[j].y * p.conv_stride + ky - 1;
          ix = crow[j].x * p.conv_stride + kx - 1;
        } else {
          const int ty = crow[j].y + 1 - ky, tx = crow[j].x + 1 - kx;
          ok = ok && ty >= 0 && tx >= 0;
          iy = ty / p.conv_stride;
          ix = tx / p.conv_stride;
          ok = ok && (iy * p.conv_stride == ty) && (ix * p.conv_stride == tx);
        }
        ok = ok && iy >= 0 && iy < p.conv_H && ix >= 0 && ix < p.conv_W;
        const unsigned lin =
            (((unsigned)(crow[j].b * p.conv_H + iy) * (unsigned)p.conv_W + (unsigned)ix) * (unsigned)p.conv_ld + (unsigned)ch) * 2u;
        unsigned off = ok ? lin : OOB;
        asm volatile("" : "+v"(off));
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_ptr)(a + j * RPI * 128), 16, off, 0, 0, 0);
      }
    }
    {
      const int ld = ext ? p.ldb2 : p.ldb;
#pragma unroll
      for (int j = 0; j < B_CH; ++j) {
        const int r = row_base + RPI * j;
        const unsigned lin = ((unsigned)r * (unsigned)ld + (unsigned)k0) * 2u;
        unsigned off = (kok && n0 + r < p.N) ? lin : OOB;
        asm volatile("" : "+v"(off));
        if constexpr (ext)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB2, (lds_ptr)(b + j * RPI * 128), 16, off, 0, 0, 0);
        else
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lds_ptr)(b + j * RPI * 128), 16, off, 0, 0, 0);
      }
    }
  };
  // full K tiles of dense main operands: the K advance rides on the scalar offset operand of the DMA and the per-lane
  // offsets are per-tile constants -> a K step costs LPT x (s_mov m0 + buffer_load), no address VALU at all.
  // (generic lambda on purpose: the DMA builtin only exists for the device target, and the host pass of hipcc must
  // not instantiate the body or it silently drops the kernel's host stub)
  auto issue_tile_fast = [&](int t, int buf, auto) __attribute__((always_inline)) {
    char* a = smem + buf * BUF_BYTES + wave_u * 8 * 128;
    char* b = a + A_BYTES;
    const int soff = t * (BK * 2);
    const unsigned sa = (unsigned)(RPI * 2) * (unsigned)p.lda, sb = (unsigned)(RPI * 2) * (unsigned)p.ldb;
#pragma unroll
    for (int j = 0; j < A_CH; ++j) {
      unsigned off = (validA >> j) & 1 ? baseA + (unsigned)j * sa : OOB;
      asm volatile("" : "+v"(off));
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_ptr)(a + j * RPI * 128), 16, off, soff, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < B_CH; ++j) {
      unsigned off = (validB >> j) & 1 ? baseB + (unsigned)j * sb : OOB;
      asm volatile("" : "+v"(off));
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lds_ptr)(b + j * RPI * 128), 16, off, soff, 0, 0);
    }
  };
  // pieces [p0, p1) of the same tile (piece j < A_CH is an A chunk, the rest are B chunks): lets the one-wave-per-SIMD
  // loop spread the DMA issue of a K tile over its MFMA sub-steps
  auto issue_pieces_fast = [&](int t, int buf, auto p0_tag, auto p1_tag) __attribute__((always_inline)) {
    constexpr int P0 = decltype(p0_tag)::value, P1 = decltype(p1_tag)::value;
    char* a = smem + buf * BUF_BYTES + wave_u * 8 * 128;
    char* b = a + A_BYTES;
    const int soff = t * (BK * 2);
    const unsigned sa = (unsigned)(RPI * 2) * (unsigned)p.lda, sb = (unsigned)(RPI * 2) * (unsigned)p.ldb;
#pragma unroll
    for (int j = P0; j < P1; ++j) {
      if (j < A_CH) {
        unsigned off = (validA >> j) & 1 ? baseA + (unsigned)j * sa : OOB;
        asm volatile("" : "+v"(off));
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_ptr)(a + j * RPI * 128), 16, off, soff, 0, 0);
      } else {
        const int jb = j - A_CH;
        unsigned off = (validB >> jb) & 1 ? baseB + (unsigned)jb * sb : OOB;
        asm volatile("" : "+v"(off));
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lds_ptr)(b + jb * RPI * 128), 16, off, soff, 0, 0);
      }
    }
  };
  auto issue_tile = [&](int t, int buf) __attribute__((always_inline)) {
    if (AMODE == MVIT_A_DENSE && (t + 1) * BK <= p.K)
      issue_tile_fast(t, buf, 0);
    else if (t < nk1)
      issue_tile_impl(t, buf, std::false_type{});
    else
      issue_tile_impl(t, buf, std::true_type{});
  };
  auto issue_prologue = [&](int first_buf) __attribute__((always_inline)) {
    int b = first_buf;
#pragma unroll
    for (int s_ = 0; s_ < NSTAGE - 1; ++s_) {
      if (t_begin + s_ < t_end) issue_tile(t_begin + s_, b);
      b = b + 1 == NSTAGE ? 0 : b + 1;
    }
  };

  const bool out_f32 = p.flags & MVIT_OUT_F32;
  const bool atomic = p.flags & MVIT_ATOMIC;
  const bool scalar_io = p.flags & 0x400;  // set by the dispatcher when a pointer / leading dimension is not vector-aligned
  float* Cf = (float*)p.C;
  bf16_t* Cb = (bf16_t*)p.C;

  int vt = blockIdx.x;
  setup_tile(vt);
  int cb = 0;  // LDS buffer holding the first K tile of the current output tile
  issue_prologue(cb);

  for (;;) {
    acc_t acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < AR; ++r) acc[i][j][r] = 0.f;
    // Column constants of this tile's epilogue (bias, LayerScale), requested BEFORE the main loop where the registers allow it
    // (EARLY_CONST): fetched at the head of the epilogue they sit behind the next tile's prologue DMA in the VM queue, and the
    // `s_waitcnt vmcnt(0)` in front of their first use made every epilogue wait for that prologue to land (loads return in order).
    constexpr int V = 8;
    constexpr int CPR = (EPI == MVIT_EPI_SWIGLU) ? 32 / V : WTN / V;  // lanes per output row
    constexpr bool EARLY_CONST = !MVIT_GEMM_LATE_CONST && PIPE && BM * BN <= 256 * 128 && !TRANS;
    const int lc = (lane % CPR) * V;
    float bias[V], gam[V], bias2[V];
    auto load_col_consts = [&](int tile_n0) __attribute__((always_inline)) {
      const int col_ = tile_n0 + wave_n * WTN + lc;
      const int nv_ = (EPI == MVIT_EPI_SWIGLU) ? V : min(V, p.N - col_);
#pragma unroll
      for (int e = 0; e < V; ++e) {
        if constexpr (EPI == MVIT_EPI_SWIGLU) {
          bias[e] = p.bias ? p.bias[col_ + e] : 0.f;
          bias2[e] = p.bias ? p.bias[col_ + 32 + e] : 0.f;
          gam[e] = 1.f;
        } else {
          bias[e] = (p.bias && e < nv_) ? p.bias[col_ + e] : 0.f;
          gam[e] = (p.gamma && e < nv_) ? p.gamma[col_ + e] : 1.f;
          bias2[e] = 0.f;
        }
      }
    };
    if constexpr (EARLY_CONST) load_col_consts(n0);
    auto mfma1 = [](const bf16x8& x, const bf16x8& y, const acc_t& c) __attribute__((always_inline)) {
      if constexpr (MI16)
        return mvit_mfma16(x, y, c, 0, 0, 0);
      else
        return mvit_mfma32(x, y, c, 0, 0, 0);
    };

    // ---------------------------------------------------------------- main loop
    int ib = cb + NSTAGE - 1 >= NSTAGE ? cb - 1 : cb + NSTAGE - 1;  // buffer receiving K tile t + NSTAGE - 1
    if constexpr (PIPE) {
      // One wave per SIMD (4 waves, 128-row sub-tiles): nothing else hides this wave's LDS latency, so the operand
      // fragments are double-buffered in registers - sub-step s+1 is read while the MFMAs of sub-step s run - and
      // the K-tile hand-over (wait for the next tile's DMA, barrier, refill of the buffer just consumed, first
      // fragments of the next tile) sits in front of the LAST sub-step's MFMAs instead of between two K tiles.
      bf16x8 fa[2][TMH], fb[2][TN];
      // fragment addresses: one lane-dependent offset per sub-step and operand; the 32-row fragment stride (4096 B)
      // does not touch the swizzle bits and rides on the ds_read immediate
      unsigned aoff[NSUB], boff[NSUB];
#pragma unroll
      for (int s_ = 0; s_ < NSUB; ++s_) {
        const unsigned sw = (unsigned)(((s_ * CPS + frag_half) ^ ((frag_row >> 1) & 7)) << 4);
        aoff[s_] = (unsigned)(wave_m * WTM + frag_row) * 128u + sw;
        boff[s_] = (unsigned)A_BYTES + (unsigned)(wave_n * WTN + frag_row) * 128u + sw;
      }
      auto load_frags = [&](const char* a, const char*, int s_, bf16x8 (&xa)[TMH], bf16x8 (&xb)[TN]) __attribute__((always_inline)) {
        const char* pa = a + aoff[s_];
        const char* pb = a + boff[s_];
#pragma unroll
        for (int i = 0; i < TMH; ++i) xa[i] = *(const bf16x8*)(pa + i * FSTRIDE);
#pragma unroll
        for (int j = 0; j < TN; ++j) xb[j] = *(const bf16x8*)(pb + j * FSTRIDE);
      };
      // (AHALF == 2) the A fragments of group h of sub-step s_, alone
      auto load_a_half = [&](const char* a, int s_, int h, bf16x8 (&xa)[TMH]) __attribute__((always_inline)) {
        const char* pa = a + aoff[s_] + h * TMH * FSTRIDE;
#pragma unroll
        for (int i = 0; i < TMH; ++i) xa[i] = *(const bf16x8*)(pa + i * FSTRIDE);
      };
      auto mma = [&](const bf16x8 (&xa)[TMH], const bf16x8 (&xb)[TN], int h = 0) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < TMH; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[h * TMH + i][j] = TRANS ? mfma1(xb[j], xa[i], acc[h * TMH + i][j]) : mfma1(xa[i], xb[j], acc[h * TMH + i][j]);
      };
      auto wait_tile = [&](int tn) __attribute__((always_inline)) {  // this lane's DMA pieces of K tile tn have landed
        if (NSTAGE == 3 && tn + 1 < t_end)
          asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LPT) : "memory");
        else
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      };
      // The refill of a step goes to the buffer consumed in the PREVIOUS step (free since that step's hand-over), so its
      // DMA pieces can be spread over the first three MFMA sub-steps instead of bursting after the hand-over; the
      // hand-over (wait for tile t+1, barrier, first fragments of tile t+1) sits in front of the last sub-step.
      // `tight` steps: the refill is a full dense tile (straight-line DMA code), the whole step is two scheduling
      // regions and the instruction mix is pinned with sched_group_barrier.
      wait_tile(t_begin);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (the previous tile's panel reads: finished, not just issued -- once per tile)
      __builtin_amdgcn_s_barrier();
      load_frags(smem + cb * BUF_BYTES, smem + cb * BUF_BYTES + A_BYTES, 0, fa[0], fb[0]);
      constexpr int PG = (LPT + 2) / 3;  // DMA pieces per sub-step
      // Explicit instruction order for the tight steps: every MFMA is followed by at most one or two companion operations
      // (a fragment read for the next sub-step, in the order the next sub-step consumes them, or one DMA piece of the
      // refill), and sched_barrier(0) keeps hipcc from regrouping them into bursts.
      // this lane's source offset of every DMA piece of a dense K tile (tile constants; the K advance rides on the scalar
      // offset): kept in registers across the tight loop so that a piece costs s_mov m0 + buffer_load and nothing else
      unsigned poff[LPT];
      auto set_piece_offsets = [&](auto) __attribute__((always_inline)) {  // generic: see issue_tile_fast
        const unsigned sa = (unsigned)(RPI * 2) * (unsigned)p.lda, sb = (unsigned)(RPI * 2) * (unsigned)p.ldb;
#pragma unroll
        for (int j = 0; j < LPT; ++j) {
          poff[j] = j < A_CH ? ((validA >> j) & 1 ? baseA + (unsigned)j * sa : OOB)
                             : ((validB >> (j - A_CH)) & 1 ? baseB + (unsigned)(j - A_CH) * sb : OOB);
          asm volatile("" : "+v"(poff[j]));
        }
      };
      auto issue_piece = [&](int t, int buf, int j, auto) __attribute__((always_inline)) {
        char* a = smem + buf * BUF_BYTES + wave_u * 8 * 128;
        const int soff = t * (BK * 2);
        const unsigned off = poff[j];  // (a local on purpose: with the subscript as the builtin's argument the host pass drops the kernel stub)
        if (j < A_CH) {
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_ptr)(a + j * RPI * 128), 16, off, soff, 0, 0);
        } else {
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lds_ptr)(a + A_BYTES + (j - A_CH) * RPI * 128), 16, off, soff, 0, 0);
        }
      };
      auto sub_seq = [&](const bf16x8 (&ca)[TMH], const bf16x8 (&cbf)[TN], bf16x8 (&na)[TMH], bf16x8 (&nbf)[TN],
                         const char* rbase, int rs, int dt, int dbuf, int p0, int p1, int mf0, int mf1, bool reads_first, auto tag,
                         int ch = 0, int rh = 0, bool rb = true) __attribute__((always_inline)) {
        // dt / dbuf: K tile and LDS buffer of the DMA pieces [p0, p1)
        // MFMAs [mf0, mf1) of the sub-step; the companions (all NR reads when mf1 is the end, DMA pieces [p0, p1)) are spread over them
        // (AHALF == 2) ch: A group the MFMAs work on; rh: A group of sub-step rs that is read; rb: whether the B fragments are read too
        const int NR = TMH + (rb ? TN : 0);
        constexpr int NM = TMH * TN;
        const int ND = p1 - p0, C = (mf1 == NM ? NR : 0) + ND, NMr = mf1 - mf0;
        const char* pa = rbase + aoff[rs] + rh * TMH * FSTRIDE;
        const char* pb = rbase + boff[rs];
#pragma unroll
        for (int m = mf0; m < mf1; ++m) {
          const int i = m / TN, j = m % TN, ia = ch * TMH + i;
          acc[ia][j] = TRANS ? mfma1(cbf[j], ca[i], acc[ia][j]) : mfma1(ca[i], cbf[j], acc[ia][j]);
#pragma unroll
          for (int c = 0; c < TMH + TN + LPT; ++c) {
            if (c < C && c * NMr / C == m - mf0) {
              // DMA pieces evenly between the reads, or (last sub-step before the hand-over) after all of them
              const int before = reads_first ? (c < NR ? 0 : c - NR) : c * ND / C;
              if (reads_first ? c >= NR : (c + 1) * ND / C > before) {
                issue_piece(dt, dbuf, p0 + before, tag);
              } else {
                const int r = c - before;
                if (r == 0)
                  na[0] = *(const bf16x8*)(pa);
                else if (!rb)
                  na[r] = *(const bf16x8*)(pa + r * FSTRIDE);
                else if (r <= TN)
                  nbf[r - 1] = *(const bf16x8*)(pb + (r - 1) * FSTRIDE);
                else
                  na[r - TN] = *(const bf16x8*)(pa + (r - TN) * FSTRIDE);
              }
              __builtin_amdgcn_sched_barrier(0);
            }
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      };
      // `pre`: the first piece group of this step's refill was already issued behind the previous step's hand-over;
      // `nxt`: issue the first group of the NEXT step's refill behind this step's hand-over (into the buffer the hand-over has
      // just freed).  The refill of a step thus runs from the previous hand-over to the end of sub-step 1 instead of over
      // sub-steps 0-2: half a K step more lead, which the two-stage 256x256 tile needs (its refill must land within the step).
      auto kstep_seq = [&](int t, auto pre_tag, auto nxt_tag) __attribute__((always_inline)) {
        constexpr bool pre = decltype(pre_tag)::value, nxt = decltype(nxt_tag)::value;
        const char* a = smem + cb * BUF_BYTES;
        constexpr int Q1 = PG < LPT ? PG : LPT, Q2 = 2 * PG < LPT ? 2 * PG : LPT;
        constexpr int NM = TMH * TN;
        // MFMAs of the last sub-step issued before the hand-over, so that its fragment reads have landed at the barrier and the
        // next tile's first fragments still get some MFMAs of lead (measured: 1 of 4 and 3 of 8 are the best splits)
        constexpr int HO = NM * 3 / 8;
        const int rt = t + NSTAGE - 1;  // tile of this step's refill, into buffer ib
        if constexpr (MI16 && AHALF == 2) {
          // two sub-steps x two A groups = four phases of TMH x TN MFMAs; the B fragments change with the sub-step only:
          //   phase 0 (s0, g0): read A(s0, g1)            phase 1 (s0, g1): read A(s1, g0) and B(s1)
          //   phase 2 (s1, g0): read A(s1, g1)            phase 3 (s1, g1): hand-over, read A(s0, g0) and B(s0) of the next tile
          sub_seq(fa[0], fb[0], fa[1], fb[1], a, 0, rt, ib, pre ? Q1 : 0, pre ? Q2 : Q1, 0, NM, false, 0, 0, 1, false);
          sub_seq(fa[1], fb[0], fa[0], fb[1], a, 1, rt, ib, pre ? Q2 : Q1, pre ? LPT : Q2, 0, NM, false, 0, 1, 0, true);
          sub_seq(fa[0], fb[1], fa[1], fb[0], a, 1, rt, ib, pre ? 0 : Q2, pre ? 0 : LPT, 0, NM, !pre, 0, 0, 1, false);
        } else if constexpr (MI16) {
          // two sub-steps of TM x TN MFMAs: the reads of sub-step 1 and the whole refill ride on sub-step 0
          sub_seq(fa[0], fb[0], fa[1], fb[1], a, 1, rt, ib, pre ? Q1 : 0, LPT, 0, NM, false, 0);
        } else if constexpr (pre) {
          sub_seq(fa[0], fb[0], fa[1], fb[1], a, 1, rt, ib, Q1, Q2, 0, NM, false, 0);
          sub_seq(fa[1], fb[1], fa[0], fb[0], a, 2, rt, ib, Q2, LPT, 0, NM, false, 0);
          sub_seq(fa[0], fb[0], fa[1], fb[1], a, 3, rt, ib, 0, 0, 0, NM, false, 0);
        } else {
          sub_seq(fa[0], fb[0], fa[1], fb[1], a, 1, rt, ib, 0, Q1, 0, NM, false, 0);
          sub_seq(fa[1], fb[1], fa[0], fb[0], a, 2, rt, ib, Q1, Q2, 0, NM, false, 0);
          sub_seq(fa[0], fb[0], fa[1], fb[1], a, 3, rt, ib, Q2, LPT, 0, NM, true, 0);
        }
        if (HO > 0) sub_seq(fa[1], fb[1], fa[0], fb[0], a, 0, rt, ib, 0, 0, 0, HO, false, 0, AHALF - 1);
        const int nb = cb + 1 == NSTAGE ? 0 : cb + 1;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (NSTAGE == 3)
          asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LPT) : "memory");
        else
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        // behind the hand-over: first fragments of the next tile, then (nxt) the first pieces of tile t + NSTAGE into the buffer
        // every wave has just finished reading (the one this step consumed)
        sub_seq(fa[1], fb[1], fa[0], fb[0], smem + nb * BUF_BYTES, 0, t + NSTAGE, cb, 0, nxt ? Q1 : 0, HO, NM, true, 0, AHALF - 1, 0, true);
        cb = nb;
        ib = ib + 1 == NSTAGE ? 0 : ib + 1;
      };
      auto kstep = [&](int t, auto tight_tag) __attribute__((always_inline)) {
        constexpr bool tight = decltype(tight_tag)::value;
        const char* a = smem + cb * BUF_BYTES;
        const char* b = a + A_BYTES;
        if constexpr (tight) {
          issue_pieces_fast(t + NSTAGE - 1, ib, std::integral_constant<int, 0>{}, std::integral_constant<int, PG>{});
        } else {
          if (t + NSTAGE - 1 < t_end) issue_tile(t + NSTAGE - 1, ib);
        }
        if constexpr (AHALF == 2) {   // four phases, as in kstep_seq
          load_a_half(a, 0, 1, fa[1]);
          mma(fa[0], fb[0], 0);
          load_frags(a, b, 1, fa[0], fb[1]);
          mma(fa[1], fb[0], 1);
          load_a_half(a, 1, 1, fa[1]);
          mma(fa[0], fb[1], 0);
        } else {
        load_frags(a, b, 1, fa[1], fb[1]);
        mma(fa[0], fb[0]);
        }
        if constexpr (NSUB == 4) {
          if constexpr (tight)
            issue_pieces_fast(t + NSTAGE - 1, ib, std::integral_constant<int, PG>{}, std::integral_constant<int, (2 * PG < LPT ? 2 * PG : LPT)>{});
          load_frags(a, b, 2, fa[0], fb[0]);
          mma(fa[1], fb[1]);
          if constexpr (tight)
            issue_pieces_fast(t + NSTAGE - 1, ib, std::integral_constant<int, (2 * PG < LPT ? 2 * PG : LPT)>{}, std::integral_constant<int, LPT>{});
          load_frags(a, b, 3, fa[1], fb[1]);
          mma(fa[0], fb[0]);
        }
        if constexpr (tight) {
#pragma unroll
          for (int g = 0; g < 3 * (TM + TN); ++g) {
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                      // 1 LDS read
            __builtin_amdgcn_sched_group_barrier(0x008, (TM * TN) / (TM + TN), 0);  // MFMAs
            if (g * LPT / (3 * (TM + TN)) != (g + 1) * LPT / (3 * (TM + TN)))
              __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);                    // 1 DMA piece
          }
        }
        const int nb = cb + 1 == NSTAGE ? 0 : cb + 1;
        if (tight || t + 1 < t_end) {
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // this wave's last reads of buffer cb are back
          if constexpr (tight) {
            if (NSTAGE == 3)
              asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LPT) : "memory");
            else
              asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          } else {
            wait_tile(t + 1);
          }
          __builtin_amdgcn_s_barrier();
          load_frags(smem + nb * BUF_BYTES, smem + nb * BUF_BYTES + A_BYTES, 0, fa[0], fb[0]);
        }
        mma(fa[1], fb[1], AHALF - 1);
        if constexpr (tight) {
#pragma unroll
          for (int g = 0; g < TM + TN; ++g) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 1);            // MFMA
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 1);            // LDS read
            __builtin_amdgcn_sched_group_barrier(0x008, (TM * TN) / (TM + TN) - 1, 1);
          }
        }
        cb = nb;
        ib = ib + 1 == NSTAGE ? 0 : ib + 1;
      };
      int t = t_begin;
      if (AMODE == MVIT_A_DENSE) {
        const int t_tight = min(t_end, p.K / BK) - (NSTAGE - 1);
        if (t < t_tight) {
          set_piece_offsets(0);
          if constexpr (NSTAGE == 2) {  // (three stages have the lead anyway: measured 2 % slower there)
            if (t + 1 < t_tight) {
              kstep_seq(t++, std::false_type{}, std::true_type{});
              for (; t + 1 < t_tight; ++t) kstep_seq(t, std::true_type{}, std::true_type{});
              kstep_seq(t++, std::true_type{}, std::false_type{});
            }
          }
          for (; t < t_tight; ++t) kstep_seq(t, std::false_type{}, std::false_type{});
        }
      }
      for (; t < t_end; ++t) kstep(t, std::false_type{});
    } else {
    for (int t = t_begin; t < t_end; ++t) {
      if (NSTAGE == 3 && t + 1 < t_end)
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LPT) : "memory");
      else
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the previous step's fragment reads are BACK (its MFMAs may be scheduled below the
      __builtin_amdgcn_s_barrier();                        //  barrier; s_barrier does not wait for LDS reads): their buffer is refilled next
      if (t + NSTAGE - 1 < t_end) issue_tile(t + NSTAGE - 1, ib);
      const char* a = smem + cb * BUF_BYTES;
      const char* b = a + A_BYTES;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        bf16x8 fa[TM], fb[TN];
        const int ch = s * 2 + frag_half;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          const int row = wave_m * WTM + i * 32 + frag_row;
          fa[i] = *(const bf16x8*)(a + row * 128 + ((ch ^ ((row >> 1) & 7)) << 4));
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const int row = wave_n * WTN + j * 32 + frag_row;
          fb[j] = *(const bf16x8*)(b + row * 128 + ((ch ^ ((row >> 1) & 7)) << 4));
        }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = TRANS ? mvit_mfma32(fb[j], fa[i], acc[i][j], 0, 0, 0)
                              : mvit_mfma32(fa[i], fb[j], acc[i][j], 0, 0, 0);
      }
      cb = cb + 1 == NSTAGE ? 0 : cb + 1;
      ib = ib + 1 == NSTAGE ? 0 : ib + 1;
    }
    }
    // every wave is done reading the last K tile: its buffer becomes the epilogue panel, the other NSTAGE-1
    // buffers (starting at cb) receive the first K tiles of the next output tile while the epilogue runs
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // (reads finished, not just issued: other waves write their panels next)
    __builtin_amdgcn_s_barrier();
    const int last = cb == 0 ? NSTAGE - 1 : cb - 1;
    const int em0 = m0, en0 = n0;
    const int vt_next = vt + gridDim.x;
    const bool has_next = vt_next < ntiles;
    // hipcc puts `s_waitcnt vmcnt(0)` in front of the first LDS read that follows an LDS-DMA builtin (it cannot tell the DMA's target
    // from the panel), so with the prologue issued here the first slab of the epilogue waits for it to land.  Issuing it behind the
    // first slab instead (LATE_PROLOGUE) was measured: qkv 75.6 vs 79.6 us and fc1 137.8 vs 141.6 us alone, but 427.8 vs 430.0
    // tiles/s inside the step (same box) -- the earlier request is worth more there than the wait costs.  Off.
    constexpr bool LATE_PROLOGUE = MVIT_GEMM_LATE_PROLOGUE && !TRANS;
    if (has_next && (!LATE_PROLOGUE || (p.flags & 0x800))) {
      setup_tile(vt_next);
      issue_prologue(cb);
    }

    // ---------------------------------------------------------------- epilogue of tile (em0, en0)
    // C/D layout of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5).  16-row slabs of the wave's
    // sub-tile are parked in a wave-private f32 panel and re-read row-wise: each lane applies the epilogue to 8
    // consecutive columns of one row and issues 16-byte stores; CPR lanes cover one contiguous row segment.
    constexpr int RPP = 64 / CPR;                                      // rows per pass
    constexpr int NPASS = RPP >= 16 ? 1 : 16 / RPP;
    float* stg = (float*)(smem + last * BUF_BYTES) + (size_t)wave * SLAB;
    const int col_l = lane & (FR - 1);
    const int lr = lane / CPR;
    const int colw = en0 + wave_n * WTN;  // first column of this wave's panel

    auto ld8bf = [&](const bf16_t* q, float (&o)[V], int nv) __attribute__((always_inline)) {
      if (nv == V && !scalar_io) {
        const uint4 t = *(const uint4*)q;
        const uint32_t u[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) o[2 * e] = lo16f(u[e]), o[2 * e + 1] = hi16f(u[e]);
      } else {
#pragma unroll
        for (int e = 0; e < V; ++e) o[e] = e < nv ? bf2f(q[e]) : 0.f;
      }
    };
    auto st8bf = [&](bf16_t* q, const float (&o)[V], int nv) __attribute__((always_inline)) {
      if (nv == V && !scalar_io) {
        uint4 t;
        t.x = pack2bf(o[0], o[1]), t.y = pack2bf(o[2], o[3]), t.z = pack2bf(o[4], o[5]), t.w = pack2bf(o[6], o[7]);
        *(uint4*)q = t;
      } else {
        for (int e = 0; e < nv; ++e) q[e] = f2bf(o[e]);
      }
    };
    auto ld8f = [&](const float* q, float (&o)[V], int nv) __attribute__((always_inline)) {
      if (nv == V && !scalar_io) {
        const float4 t0 = ((const float4*)q)[0], t1 = ((const float4*)q)[1];
        o[0] = t0.x, o[1] = t0.y, o[2] = t0.z, o[3] = t0.w, o[4] = t1.x, o[5] = t1.y, o[6] = t1.z, o[7] = t1.w;
      } else {
#pragma unroll
        for (int e = 0; e < V; ++e) o[e] = e < nv ? q[e] : 0.f;
      }
    };
    auto st8f = [&](float* q, const float (&o)[V], int nv) __attribute__((always_inline)) {
      if (nv == V && !scalar_io) {
        ((float4*)q)[0] = make_float4(o[0], o[1], o[2], o[3]);
        ((float4*)q)[1] = make_float4(o[4], o[5], o[6], o[7]);
      } else {
        for (int e = 0; e < nv; ++e) q[e] = o[e];
      }
    };
    auto panel8 = [&](int rl, int c0, float (&o)[V]) __attribute__((always_inline)) {
      const float4 t0 = *(const float4*)(stg + rl * SLD + c0), t1 = *(const float4*)(stg + rl * SLD + c0 + 4);
      o[0] = t0.x, o[1] = t0.y, o[2] = t0.z, o[3] = t0.w, o[4] = t1.x, o[5] = t1.y, o[6] = t1.z, o[7] = t1.w;
    };

    float st_s[V], st_q[V];
#pragma unroll
    for (int e = 0; e < V; ++e) st_s[e] = st_q[e] = 0.f;

    // column-dependent epilogue constants
    const int col = colw + lc;
    const int nv = (EPI == MVIT_EPI_SWIGLU) ? V : min(V, p.N - col);  // valid columns of this lane's group
    if constexpr (!EARLY_CONST) load_col_consts(en0);

    // Epilogues that READ per-element operands (residual stream, saved SwiGLU pre-activation) fetch them one slab
    // ahead: otherwise every row pass would expose a full HBM round trip behind its dependent store.
    constexpr bool AUX_PF = (EPI == MVIT_EPI_RESID || EPI == MVIT_EPI_DSWIGLU);
    const bool vec_ok = nv == V && !scalar_io;
    uint4 pre[2][NPASS][2];
    auto issue_aux = [&](int slab, int q) __attribute__((always_inline)) {
      if constexpr (AUX_PF) {
#pragma unroll
        for (int it = 0; it < NPASS; ++it) {
          const int rl = it * RPP + lr;
          const int row = em0 + wave_m * WTM + (slab >> 1) * 32 + 16 * (slab & 1) + rl;
          const bool ok = vec_ok && row < p.M && (RPP <= 16 || rl < 16);
          pre[q][it][0] = pre[q][it][1] = make_uint4(0, 0, 0, 0);
          if (ok) {
            if constexpr (EPI == MVIT_EPI_RESID) {
              const float* rp = p.aux ? (const float*)p.aux + (size_t)row * p.ldaux + col : Cf + (size_t)row * p.ldc + col;
              pre[q][it][0] = ((const uint4*)rp)[0];
              pre[q][it][1] = ((const uint4*)rp)[1];
            } else {
              const bf16_t* u = (const bf16_t*)p.aux + (size_t)row * p.ldaux + (((col >> 5) << 6) + (col & 31));
              pre[q][it][0] = *(const uint4*)u;
              pre[q][it][1] = *(const uint4*)(u + 32);
            }
          }
        }
      }
    };
    auto un8bf = [&](const uint4& t, float (&o)[V]) __attribute__((always_inline)) {
      const uint32_t u[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) o[2 * e] = lo16f(u[e]), o[2 * e + 1] = hi16f(u[e]);
    };

    if constexpr (TRANS) {
      // D^T layout of the 32x32 MFMA: lane & 31 = row of C inside the 32-row block, register r = column (r&3) + 8*(r>>2) + 4*half.
      // bf16: the two half-waves hold interleaved 4-column groups of the same rows; v_permlane32_swap pairs them into 8
      // consecutive columns per lane -> 16-byte stores (cdna_hip_programming.md T21).
      if (!(p.flags & 0x800)) {
        float4 b4[TN][4];
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int c0 = en0 + wave_n * WTN + j * 32 + 8 * q + 4 * frag_half;
            b4[j][q] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (p.bias) {
              if (c0 + 3 < p.N && !scalar_io) {
                b4[j][q] = *(const float4*)(p.bias + c0);
              } else {
                b4[j][q].x = c0 < p.N ? p.bias[c0] : 0.f, b4[j][q].y = c0 + 1 < p.N ? p.bias[c0 + 1] : 0.f;
                b4[j][q].z = c0 + 2 < p.N ? p.bias[c0 + 2] : 0.f, b4[j][q].w = c0 + 3 < p.N ? p.bias[c0 + 3] : 0.f;
              }
            }
          }
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          const int row = em0 + wave_m * WTM + i * 32 + frag_row;
          const bool rok = row < p.M;
#pragma unroll
          for (int j = 0; j < TN; ++j) {
            const int cb0 = en0 + wave_n * WTN + j * 32;
            // (C += with bf16 output takes the element path below: ONE rounding of old + new, as every other epilogue of this file
            //  and of gemm_ws.hip -- the packed path would round the new value before the addition and the sum again)
            const bool fast = !scalar_io && !atomic && !out_f32 && cb0 + 32 <= p.N && !(p.flags & MVIT_ACCUM_BF16);
            if (fast) {
#pragma unroll
              for (int pr = 0; pr < 2; ++pr) {      // group pairs (0,1) and (2,3)
                const int q0 = 2 * pr, q1 = 2 * pr + 1;
                unsigned a0 = pack2bf(acc[i][j][4 * q0] + b4[j][q0].x, acc[i][j][4 * q0 + 1] + b4[j][q0].y);
                unsigned a1 = pack2bf(acc[i][j][4 * q0 + 2] + b4[j][q0].z, acc[i][j][4 * q0 + 3] + b4[j][q0].w);
                unsigned c0 = pack2bf(acc[i][j][4 * q1] + b4[j][q1].x, acc[i][j][4 * q1 + 1] + b4[j][q1].y);
                unsigned c1 = pack2bf(acc[i][j][4 * q1 + 2] + b4[j][q1].z, acc[i][j][4 * q1 + 3] + b4[j][q1].w);
                const auto s0 = __builtin_amdgcn_permlane32_swap(a0, c0, false, false);
                const auto s1 = __builtin_amdgcn_permlane32_swap(a1, c1, false, false);
                // lanes 0-31: columns 16*pr + 0..7, lanes 32-63: columns 16*pr + 8..15 of the block
                uint4 o4 = make_uint4(s0[0], s1[0], s0[1], s1[1]);
                if (rok) {
                  bf16_t* dst = Cb + (size_t)row * p.ldc + cb0 + 16 * pr + 8 * frag_half;
                  *(uint4*)dst = o4;
                }
              }
            } else if (rok) {
#pragma unroll
              for (int q = 0; q < 4; ++q) {
                const int c0 = cb0 + 8 * q + 4 * frag_half;
                const float v[4] = {acc[i][j][4 * q] + b4[j][q].x, acc[i][j][4 * q + 1] + b4[j][q].y, acc[i][j][4 * q + 2] + b4[j][q].z,
                                    acc[i][j][4 * q + 3] + b4[j][q].w};
                const size_t o = (size_t)row * p.ldc + c0;
                if (atomic) {
                  for (int e = 0; e < 4; ++e)
                    if (c0 + e < p.N) atomicAdd(Cf + o + e, v[e]);
                } else if (out_f32) {
                  if (c0 + 3 < p.N && !scalar_io) {
                    *(float4*)(Cf + o) = make_float4(v[0], v[1], v[2], v[3]);
                  } else {
                    for (int e = 0; e < 4; ++e)
                      if (c0 + e < p.N) Cf[o + e] = v[e];
                  }
                } else {
                  for (int e = 0; e < 4; ++e)
                    if (c0 + e < p.N) {
                      float t_ = v[e];
                      if (p.flags & MVIT_ACCUM_BF16) t_ += bf2f(Cb[o + e]);
                      Cb[o + e] = f2bf(t_);
                    }
                }
              }
            }
          }
        }
      }
    } else
    if (!(p.flags & 0x800)) {  // (0x800: debug, no epilogue)
      issue_aux(0, 0);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int g = 0; g < FR / 16; ++g) {
          constexpr int NSLAB = WTM / 16;
          const int slab = i * (FR / 16) + g;
          if (slab + 1 < NSLAB) issue_aux(slab + 1, (slab + 1) & 1);
          // park rows FR*i + 16*g .. +15 of the wave's sub-tile: accumulator registers 8g .. 8g+7 of the 32x32 blocks, all four of
          // the 16x16 blocks (C/D layout there: col = lane & 15, row = 4 * (lane >> 4) + register)
#pragma unroll
          for (int j = 0; j < TN; ++j) {
            if constexpr (MI16) {
#pragma unroll
              for (int r4 = 0; r4 < 4; ++r4) stg[(r4 + 4 * frag_half) * SLD + j * 16 + col_l] = acc[i][j][r4];
            } else {
#pragma unroll
              for (int r8 = 0; r8 < 8; ++r8)
                stg[((r8 & 3) + 8 * (r8 >> 2) + 4 * frag_half) * SLD + j * 32 + col_l] = acc[i][j][8 * g + r8];
            }
          }
#pragma unroll
          for (int it = 0; it < NPASS; ++it) {
            const int rl = it * RPP + lr;  // row inside the slab
            if (RPP > 16 && rl >= 16) continue;
            const int row = em0 + wave_m * WTM + i * FR + 16 * g + rl;
            if constexpr (EPI == MVIT_EPI_SWIGLU) {
              bf16_t* aux = (bf16_t*)p.aux;
              const int ca = col, cbb = col + 32, cg = (colw >> 1) + lc;
              float a_[V], b_[V], g_[V];
              panel8(rl, lc, a_);
              panel8(rl, lc + 32, b_);
              if (row >= p.M) continue;
#pragma unroll
              for (int e = 0; e < V; ++e) a_[e] += bias[e], b_[e] += bias2[e];
              if (aux) {
                st8bf(aux + (size_t)row * p.ldaux + ca, a_, V);
                st8bf(aux + (size_t)row * p.ldaux + cbb, b_, V);
              }
#pragma unroll
              for (int e = 0; e < V; ++e) g_[e] = a_[e] * sigmoidf_(a_[e]) * b_[e];
              st8bf(Cb + (size_t)row * p.ldc + cg, g_, V);
            } else {
              float v[V];
              panel8(rl, lc, v);
              if (row >= p.M || nv <= 0) continue;
#pragma unroll
              for (int e = 0; e < V; ++e) v[e] += bias[e];
              if constexpr (EPI == MVIT_EPI_STORE) {
                const size_t o = (size_t)row * p.ldc + col;
                if constexpr (AMODE == MVIT_A_CONV3) {   // (folded BatchNorm + ReLU of the eval-mode ConvStream; the dispatcher admits the flag here only)
                  if (p.flags & MVIT_RELU) {
#pragma unroll
                    for (int e = 0; e < V; ++e) v[e] = fmaxf(v[e], 0.f);
                  }
                }
                if (atomic) {
                  for (int e = 0; e < nv; ++e) atomicAdd(Cf + o + e, v[e]);
                } else if (out_f32) {
                  st8f(Cf + o, v, nv);
                } else {
                  if (p.flags & MVIT_ACCUM_BF16) {
                    float old[V];
                    ld8bf(Cb + o, old, nv);
#pragma unroll
                    for (int e = 0; e < V; ++e) v[e] += old[e];
                  }
                  st8bf(Cb + o, v, nv);
                }
              } else if constexpr (EPI == MVIT_EPI_GELU) {
                if (p.aux) st8bf((bf16_t*)p.aux + (size_t)row * p.ldaux + col, v, nv);
                float g_[V];
#pragma unroll
                for (int e = 0; e < V; ++e) g_[e] = gelu_erf(v[e]);
                st8bf(Cb + (size_t)row * p.ldc + col, g_, nv);
              } else if constexpr (EPI == MVIT_EPI_RESID) {
                const size_t o = (size_t)row * p.ldc + col;
                float r_[V];
                if (vec_ok) {
                  const uint4 t0 = pre[slab & 1][it][0], t1 = pre[slab & 1][it][1];
                  r_[0] = __uint_as_float(t0.x), r_[1] = __uint_as_float(t0.y), r_[2] = __uint_as_float(t0.z), r_[3] = __uint_as_float(t0.w);
                  r_[4] = __uint_as_float(t1.x), r_[5] = __uint_as_float(t1.y), r_[6] = __uint_as_float(t1.z), r_[7] = __uint_as_float(t1.w);
                } else {
                  ld8f(p.aux ? (const float*)p.aux + (size_t)row * p.ldaux + col : Cf + o, r_, nv);
                }
                const float rsc = p.rowscale ? p.rowscale[row] : 1.f;   // DropPath factor of this row's sample (uniform branch)
#pragma unroll
                for (int e = 0; e < V; ++e) r_[e] += rsc * gam[e] * v[e];
                st8f(Cf + o, r_, nv);
              } else if constexpr (EPI == MVIT_EPI_PATCH) {
                const int img = row / p.patch_P, pp = row - img * p.patch_P;
                const size_t o = (size_t)(img * p.patch_ntok + p.patch_prefix + pp) * p.ldc + col;
                float pe[V];
                ld8f(p.pos + (size_t)pp * p.N + col, pe, nv);
#pragma unroll
                for (int e = 0; e < V; ++e) v[e] += pe[e];
                st8f(Cf + o, v, nv);
              } else if constexpr (EPI == MVIT_EPI_STATS) {
                st8bf(Cb + (size_t)row * p.ldc + col, v, nv);
#pragma unroll
                for (int e = 0; e < V; ++e)
                  if (e < nv) st_s[e] += v[e], st_q[e] += v[e] * v[e];
              } else if constexpr (EPI == MVIT_EPI_DSWIGLU) {
                // gate columns col..col+7 live at packed positions ca.. (a) and ca+32.. (b) of the saved pre-activation
                const bf16_t* u = (const bf16_t*)p.aux;
                const int ca = ((col >> 5) << 6) + (col & 31);
                float a_[V], b_[V], da[V], db[V];
                if (vec_ok) {
                  un8bf(pre[slab & 1][it][0], a_);
                  un8bf(pre[slab & 1][it][1], b_);
                } else {
                  ld8bf(u + (size_t)row * p.ldaux + ca, a_, nv);
                  ld8bf(u + (size_t)row * p.ldaux + ca + 32, b_, nv);
                }
#pragma unroll
                for (int e = 0; e < V; ++e) {
                  const float sg = sigmoidf_(a_[e]);
                  da[e] = v[e] * b_[e] * sg * (1.f + a_[e] * (1.f - sg));
                  db[e] = v[e] * a_[e] * sg;
                }
                st8bf(Cb + (size_t)row * p.ldc + ca, da, nv);
                st8bf(Cb + (size_t)row * p.ldc + ca + 32, db, nv);
              } else if constexpr (EPI == MVIT_EPI_DGELU) {
                float u_[V], o_[V];
                ld8bf((const bf16_t*)p.aux + (size_t)row * p.ldaux + col, u_, nv);
#pragma unroll
                for (int e = 0; e < V; ++e) o_[e] = v[e] * gelu_erf_grad(u_[e]);
                st8bf(Cb + (size_t)row * p.ldc + col, o_, nv);
              }
            }
          }
          if (LATE_PROLOGUE && slab == 0 && has_next) {   // (see above: the next tile's first K tiles, behind the first slab)
            setup_tile(vt_next);
            issue_prologue(cb);
          }
        }
    }

    if constexpr (EPI == MVIT_EPI_STATS) {
      // lanes with equal lane % CPR own the same V columns; then across the WAVES_M waves through LDS
      float* red = (float*)(smem + last * BUF_BYTES) + (size_t)NW * SLAB;  // [WAVES_M][BN][2], behind the panels
#pragma unroll
      for (int e = 0; e < V; ++e) {
#pragma unroll
        for (int o = CPR; o < 64; o <<= 1) {
          st_s[e] += __shfl_xor(st_s[e], o, 64);
          st_q[e] += __shfl_xor(st_q[e], o, 64);
        }
      }
      if (lane < CPR) {
#pragma unroll
        for (int e = 0; e < V; ++e) {
          const int c = wave_n * WTN + lc + e;
          red[(wave_m * BN + c) * 2 + 0] = st_s[e];
          red[(wave_m * BN + c) * 2 + 1] = st_q[e];
        }
      }
      __syncthreads();
      if (tid < BN && en0 + tid < p.N) {
        double s_ = 0., q_ = 0.;
#pragma unroll
        for (int w = 0; w < WAVES_M; ++w) {
          s_ += red[(w * BN + tid) * 2 + 0];
          q_ += red[(w * BN + tid) * 2 + 1];
        }
        // slot: one writer block per slot when the caller provides at least gridDim.x slots (sums are then added in program
        // order only: run-to-run identical statistics); otherwise slots are shared and merely spread the atomic traffic
        double* st = p.stats + (size_t)(p.nslots >= (int)gridDim.x ? (int)blockIdx.x : (int)((blockIdx.x + vt) % p.nslots)) * 2 * p.N;
        atomicAdd(st + en0 + tid, s_);
        atomicAdd(st + p.N + en0 + tid, q_);
      }
    }
    if (!has_next) break;
    vt = vt_next;
  }
}

int gemm_num_cus();

template <int BM, int BN, int WAVES_M, int WAVES_N, int AMODE, int EPI>
int launch_one(const mvit_gemm_args& a, hipStream_t s) {
  const int tiles = ((a.M + BM - 1) / BM) * ((a.N + BN - 1) / BN);
  const size_t lds = gemm_lds_bytes<BM, BN, WAVES_M, WAVES_N>();
  // persistent grid: as many blocks as fit the chip at once (LDS-limited), each walking several tiles
  int per_cu = (int)((160 * 1024) / lds);
  per_cu = per_cu < 1 ? 1 : (per_cu > 4 ? 4 : per_cu);
  int gx = gemm_num_cus() * per_cu;
  if (gx > tiles) gx = tiles;
  if (EPI == MVIT_EPI_STATS && a.nslots >= 256 && gx > a.nslots) gx = a.nslots;   // (>= 256 slots: one writer per slot, see the epilogue)
  dim3 grid(gx, 1, a.ksplit > 1 ? a.ksplit : 1);
  auto kern = gemm_kernel<BM, BN, WAVES_M, WAVES_N, AMODE, EPI>;
  static mvit_per_device_size raised;  // per instantiation, per device
  if (mvit_ensure_dynamic_lds((const void*)kern, lds, raised) != MVIT_OK) return MVIT_EINVAL;
  hipLaunchKernelGGL(kern, grid, dim3(64 * WAVES_M * WAVES_N), lds, s, a);
  return MVIT_LAUNCH_CHECK();
}

// one translation unit per (tile, A-mode): keeps hipcc compile times parallel
template <int BM, int BN, int WAVES_M, int WAVES_N>
int launch_dense(const mvit_gemm_args& a, hipStream_t s);
template <int BM, int BN, int WAVES_M, int WAVES_N>
int launch_conv(const mvit_gemm_args& a, hipStream_t s);

#define MVIT_GEMM_DENSE_UNIT(BM, BN, WM, WN)                                                      \
  template <>                                                                                     \
  int launch_dense<BM, BN, WM, WN>(const mvit_gemm_args& a, hipStream_t s) {                      \
    switch (a.epi) {                                                                              \
      case MVIT_EPI_STORE: return launch_one<BM, BN, WM, WN, MVIT_A_DENSE, MVIT_EPI_STORE>(a, s); \
      case MVIT_EPI_GELU: return launch_one<BM, BN, WM, WN, MVIT_A_DENSE, MVIT_EPI_GELU>(a, s);   \
      case MVIT_EPI_SWIGLU: return launch_one<BM, BN, WM, WN, MVIT_A_DENSE, MVIT_EPI_SWIGLU>(a, s); \
      case MVIT_EPI_RESID: return launch_one<BM, BN, WM, WN, MVIT_A_DENSE, MVIT_EPI_RESID>(a, s); \
      case MVIT_EPI_PATCH: return launch_one<BM, BN, WM, WN, MVIT_A_DENSE, MVIT_EPI_PATCH>(a, s); \
      case MVIT_EPI_DSWIGLU: return launch_one<BM, BN, WM, WN, MVIT_A_DENSE, MVIT_EPI_DSWIGLU>(a, s); \
      case MVIT_EPI_DGELU: return launch_one<BM, BN, WM, WN, MVIT_A_DENSE, MVIT_EPI_DGELU>(a, s); \
      default: return MVIT_EINVAL;                                                                \
    }                                                                                             \
  }

#define MVIT_GEMM_CONV_UNIT(BM, BN, WM, WN)                                                       \
  template <>                                                                                     \
  int launch_conv<BM, BN, WM, WN>(const mvit_gemm_args& a, hipStream_t s) {                       \
    if (a.amode == MVIT_A_CONV3) {                                                                \
      if (a.epi == MVIT_EPI_STATS) return launch_one<BM, BN, WM, WN, MVIT_A_CONV3, MVIT_EPI_STATS>(a, s); \
      if (a.epi == MVIT_EPI_STORE) return launch_one<BM, BN, WM, WN, MVIT_A_CONV3, MVIT_EPI_STORE>(a, s); \
    } else if (a.amode == MVIT_A_CONV3_T) {                                                       \
      if (a.epi == MVIT_EPI_STORE) return launch_one<BM, BN, WM, WN, MVIT_A_CONV3_T, MVIT_EPI_STORE>(a, s); \
    } else if (a.amode == MVIT_A_PATCH) {                                                         \
      if (a.epi == MVIT_EPI_PATCH) return launch_one<BM, BN, WM, WN, MVIT_A_PATCH, MVIT_EPI_PATCH>(a, s); \
    }                                                                                             \
    return MVIT_EINVAL;                                                                           \
  }

}  // namespace mvit_gemm
