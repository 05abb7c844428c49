// Wave-specialised bf16 MFMA GEMM, ONE consumer wave per SIMD (round 5): C[M,N] = A[M,K] * B[N,K]^T (+ A2 * B2^T), 256x128x64 tiles,
// plain bf16 store epilogue only.
//
// Same producer side, LDS image, ring protocol and tile order as gemm_ws.hip; what changes is the consumer geometry: 4 consumer waves
// (2 x 2, 128x64 sub-tiles, 128 accumulator + 96 double-buffered fragment registers at two waves per SIMD = 256 registers) instead of
// 8 (4 x 2, 64x64).  Why: tools/ws_timing.py puts the K loop of gemm_ws.hip at 1210-1280 cycles per K tile against 1024 of MFMA issue,
// and every measurement of round 5 says the loop is paced by the bytes a CU moves per flop, not by the matrix pipe: per K tile the
// eight 64x64 waves read 128 KB of fragments from LDS (each A row block twice, each B column block four times) beside the 48 KB the
// DMA writes.  A 128x64 sub-tile reads (128 + 64) x 128 B = 24 KB per wave, 96 KB per K tile (-25 %), and 8 instead of 12 waves meet
// in the K-tile barrier.  The price is one MFMA stream per SIMD (no partner wave to fill its bubbles) -- which is why the vendor's
// 256x128 kernel has this shape and why this file exists as a measured alternative, selected per problem by the dispatcher
// (mvit_gemm::ws4_supported; MVIT_GEMM_WS4 in the dbg library).
#include <type_traits>
#include "common.hpp"
#include "../../include/miphei_hip.h"

namespace mvit_gemm {
int gemm_num_cus();

namespace ws4 {
constexpr int BM = 256, BN = 128, BK = 64, NSTAGE = 3;
constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, BUF_BYTES = A_BYTES + B_BYTES;
constexpr int NCW = 4, NPW = 4;                      // consumer / producer waves
constexpr int PPW = (BM + BN) / 8 / NPW;             // DMA pieces (8 rows x 128 B) per producer wave per K tile = 12
constexpr int PA = BM / 8 / NPW, PB = BN / 8 / NPW;  // of which A / B pieces: 8 + 4
constexpr int WTM = 128, WTN = 64, TM = 8, TN = 4;
constexpr int SLDW = WTN + 4;                        // packed-store panel row (one row PAIR) in dwords
constexpr unsigned OOB = 0x80000000u;
#ifndef MVIT_WS_GROUP_M
#define MVIT_WS_GROUP_M 4
#endif
typedef __attribute__((address_space(3))) void* lds_ptr;
static_assert(PPW == PA + PB, "piece split");

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* ptr, unsigned bytes) {
  const unsigned long long v = (unsigned long long)ptr;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
  return __builtin_amdgcn_make_buffer_rsrc((void*)(((unsigned long long)hi << 32) | lo), 0,
                                           (int)__builtin_amdgcn_readfirstlane(bytes), 0x00020000);
}
struct WsExtra {
  unsigned grid_magic, pg_magic;
};
__device__ __forceinline__ int div_small(int x, int g) {
  return g == 4 ? x >> 2 : g == 2 ? x >> 1 : g == 1 ? x : (int)__umulhi((unsigned)x, 0x55555556u);
}
struct TileOrder {   // as gemm_ws.hip
  int tiles_m, tiles_n, ntiles;
  unsigned pg_magic;
  __device__ __forceinline__ void get(int vt, int& m0, int& n0) const {
    const int q = ntiles >> 3, r = ntiles & 7, xcd = vt & 7;
    const int wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (vt >> 3);
    constexpr int GROUP_M = MVIT_WS_GROUP_M;
    static_assert(GROUP_M <= 4, "div_small covers group heights 1 .. 4");
    const int per_group = GROUP_M * tiles_n;
    const int grp = (int)mvit_fast_div((unsigned)wg, (unsigned)per_group, pg_magic);
    const int first_m = grp * GROUP_M;
    const int gsz = min(tiles_m - first_m, GROUP_M);
    const int in_grp = wg - grp * per_group;
    const int cg = div_small(in_grp, gsz);
    m0 = (first_m + (in_grp - cg * gsz)) * BM;
    n0 = cg * BN;
  }
};

__global__ __launch_bounds__(64 * (NCW + NPW)) void gemm_ws4_kernel(const mvit_gemm_args p, const WsExtra xp) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  TileOrder ord;
  ord.tiles_m = (p.M + BM - 1) / BM;
  ord.tiles_n = p.N / BN;
  ord.ntiles = ord.tiles_m * ord.tiles_n;
  ord.pg_magic = xp.pg_magic;
  const int nk1 = p.K / BK;
  const int nk2 = p.A2 ? (p.K2 + BK - 1) / BK : 0;
  const int nk = nk1 + nk2;
  const int my_tiles = ((int)blockIdx.x < ord.ntiles) ? (int)mvit_fast_div((unsigned)(ord.ntiles - 1 - (int)blockIdx.x), gridDim.x, xp.grid_magic) + 1 : 0;
  const int G = my_tiles * nk;

  if (wave >= NCW) {
    // ================================================================ producers (gemm_ws.hip's, without band items / pseudo tiles)
    const int pw = wave - NCW;
    unsigned voA[PA], voB[PB], voA2[PA], voB2[PB];
    const int rl = lane >> 3, c8 = lane & 7;
#pragma unroll
    for (int j = 0; j < PA; ++j) {
      const int row = (pw * PA + j) * 8 + rl;
      const int cs = c8 ^ ((row >> 1) & 7);
      voA[j] = (unsigned)row * (unsigned)p.lda * 2u + (unsigned)cs * 16u;
      voA2[j] = (cs * 8 < p.K2) ? (unsigned)row * (unsigned)p.lda2 * 2u + (unsigned)cs * 16u : OOB;
    }
#pragma unroll
    for (int j = 0; j < PB; ++j) {
      const int row = (pw * PB + j) * 8 + rl;
      const int cs = c8 ^ ((row >> 1) & 7);
      voB[j] = (unsigned)row * (unsigned)p.ldb * 2u + (unsigned)cs * 16u;
      voB2[j] = (cs * 8 < p.K2) ? (unsigned)row * (unsigned)p.ldb2 * 2u + (unsigned)cs * 16u : OOB;
    }
    __amdgpu_buffer_rsrc_t rsA, rsB, rsA2, rsB2;
    auto set_unit = [&](int idx) __attribute__((always_inline)) {
      int m0, n0;
      ord.get(blockIdx.x + idx * gridDim.x, m0, n0);
      const unsigned vm = (unsigned)min(BM, p.M - m0), vn = (unsigned)min(BN, p.N - n0);
      rsA = make_rsrc((const bf16_t*)p.A + (size_t)m0 * p.lda, vm * (unsigned)p.lda * 2u);
      rsB = make_rsrc((const bf16_t*)p.B + (size_t)n0 * p.ldb, vn * (unsigned)p.ldb * 2u);
      rsA2 = make_rsrc(p.A2 ? (const bf16_t*)p.A2 + (size_t)m0 * p.lda2 : (const bf16_t*)p.A, p.A2 ? vm * (unsigned)p.lda2 * 2u : 0u);
      rsB2 = make_rsrc(p.B2 ? (const bf16_t*)p.B2 + (size_t)n0 * p.ldb2 : (const bf16_t*)p.B, p.B2 ? vn * (unsigned)p.ldb2 * 2u : 0u);
    };
    // (generic lambda: the DMA builtin exists for the device target only, see gemm_kernel.hpp)
    auto issue = [&](int k, int stage, auto) __attribute__((always_inline)) {
      char* a = smem + stage * BUF_BYTES + pw * PA * 1024;
      char* b = smem + stage * BUF_BYTES + A_BYTES + pw * PB * 1024;
      const int soff = k < nk1 ? k * (BK * 2) : 0;
      if (k < nk1) {
#pragma unroll
        for (int j = 0; j < PA; ++j) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_ptr)(a + j * 1024), 16, voA[j], soff, 0, 0);
#pragma unroll
        for (int j = 0; j < PB; ++j) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lds_ptr)(b + j * 1024), 16, voB[j], soff, 0, 0);
      } else {
#pragma unroll
        for (int j = 0; j < PA; ++j) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA2, (lds_ptr)(a + j * 1024), 16, voA2[j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < PB; ++j) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB2, (lds_ptr)(b + j * 1024), 16, voB2[j], 0, 0, 0);
      }
    };
    int l_unit = 0, l_k = 0, l_stage = 0, l_g = 0;
    if (G > 0) set_unit(0);
    auto issue_next = [&](auto tag) __attribute__((always_inline)) {
      issue(l_k, l_stage, tag);
      ++l_g;
      l_stage = l_stage + 1 == NSTAGE ? 0 : l_stage + 1;
      if (++l_k == nk) {
        l_k = 0;
        ++l_unit;
        if (l_g < G) set_unit(l_unit);
      }
    };
    if (G > 0) issue_next(0);
    if (G > 1) {
      issue_next(0);
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPW) : "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();                        // B(-1): K tile 0 has landed [no LDS reads pending]: producer waves never read LDS
    int g = 0;
    for (int t = 0; t < my_tiles; ++t) {
      for (int k = 0; k < nk; ++k, ++g) {
        if (g + 2 < G) {
          issue_next(0);
          asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPW) : "memory");     // K tile g + 1 is in LDS
        } else {
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();                    // B(g) [no LDS reads pending]: producer
      }
      __builtin_amdgcn_s_barrier();                      // B'(unit): the consumers are done with the epilogue panel [no LDS reads pending]: producer
    }
    return;
  }

  // ================================================================== consumers: 2 x 2 waves, 128 x 64 sub-tiles
  const int wave_m = wave >> 1, wave_n = wave & 1;
  const int fr = lane & 15, fh = lane >> 4;
  bf16_t* Cb = (bf16_t*)p.C;
  __builtin_amdgcn_s_barrier();                          // B(-1) [no LDS reads pending]: before the first fragment read
  int stage = 0;
  int vt = blockIdx.x;
  for (int t = 0; t < my_tiles; ++t, vt += gridDim.x) {
    int m_base, n0;
    ord.get(vt, m_base, n0);
    unsigned aoff[2], boff[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const unsigned sw = (unsigned)(((s * 4 + fh) ^ ((fr >> 1) & 7)) << 4);
      aoff[s] = (unsigned)(wave_m * WTM + fr) * 128u + sw;
      boff[s] = (unsigned)A_BYTES + (unsigned)(wave_n * WTN + fr) * 128u + sw;
    }
    if (m_base + wave_m * WTM >= p.M) {
      // this wave's rows lie entirely beyond M: it only keeps the block's barriers
      for (int k = 0; k < nk; ++k) {
        __builtin_amdgcn_s_barrier();                    // B(g) [no LDS reads pending]: a wave without rows reads nothing
        stage = stage + 1 == NSTAGE ? 0 : stage + 1;
      }
      __builtin_amdgcn_s_barrier();                      // B'(unit) [no LDS reads pending]
      continue;
    }
    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    bf16x8 fa[2][TM], fb[2][TN];
    {
      const char* cur = smem + stage * BUF_BYTES;
#pragma unroll
      for (int i = 0; i < TM; ++i) fa[0][i] = *(const bf16x8*)(cur + aoff[0] + i * 2048);
#pragma unroll
      for (int j = 0; j < TN; ++j) fb[0][j] = *(const bf16x8*)(cur + boff[0] + j * 2048);
    }
    // read r of a sub-step in the order the MFMAs (i major, j minor) consume them: a0, b0..b3, a1..a7
    auto read_sub = [&](const char* base, int s, int r, bf16x8 (&xa)[TM], bf16x8 (&xb)[TN]) __attribute__((always_inline)) {
      if (r == 0)
        xa[0] = *(const bf16x8*)(base + aoff[s]);
      else if (r <= TN)
        xb[r - 1] = *(const bf16x8*)(base + boff[s] + (r - 1) * 2048);
      else
        xa[r - TN] = *(const bf16x8*)(base + aoff[s] + (r - TN) * 2048);
    };
    constexpr int NM = TM * TN, NR = TM + TN;            // 32 MFMAs / 12 fragment reads per sub-step
    auto kstep = [&](auto more_tag) __attribute__((always_inline)) {
      constexpr bool more = decltype(more_tag)::value;
      const char* cur = smem + stage * BUF_BYTES;
#pragma unroll
      for (int m = 0; m < NM; ++m) {
        const int i = m / TN, j = m % TN;
        acc[i][j] = mvit_mfma16(fa[0][i], fb[0][j], acc[i][j], 0, 0, 0);
#pragma unroll
        for (int r = 0; r < NR; ++r)
          if (r >= (m * NR + NM - 1) / NM && r < ((m + 1) * NR + NM - 1) / NM) read_sub(cur, 1, r, fa[1], fb[1]);
        __builtin_amdgcn_sched_barrier(0);
      }
#ifndef MVIT_WS4_HO
#define MVIT_WS4_HO 12
#endif
      constexpr int HO = MVIT_WS4_HO;                    // MFMAs of sub-step 1 ahead of the hand-over
#pragma unroll
      for (int m = 0; m < HO; ++m) {
        const int i = m / TN, j = m % TN;
        acc[i][j] = mvit_mfma16(fa[1][i], fb[1][j], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();                      // B(g)
      __builtin_amdgcn_sched_barrier(0);
      stage = stage + 1 == NSTAGE ? 0 : stage + 1;
      const char* nxt = smem + stage * BUF_BYTES;
#pragma unroll
      for (int m = HO; m < NM; ++m) {
        const int i = m / TN, j = m % TN;
        acc[i][j] = mvit_mfma16(fa[1][i], fb[1][j], acc[i][j], 0, 0, 0);
        if (more) {
          constexpr int NM2 = NM - HO;
#pragma unroll
          for (int r = 0; r < NR; ++r)
            if (r >= ((m - HO) * NR + NM2 - 1) / NM2 && r < ((m - HO + 1) * NR + NM2 - 1) / NM2) read_sub(nxt, 0, r, fa[0], fb[0]);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    for (int k = 0; k + 1 < nk; ++k) kstep(std::true_type{});
    kstep(std::false_type{});
    const int last = stage == 0 ? NSTAGE - 1 : stage - 1;

    // ---------------------------------------------------------------- epilogue: bf16 store through the half-size panel (gemm_ws.hip)
    uint32_t* stgw = (uint32_t*)(smem + last * BUF_BYTES) + (size_t)wave * (8 * SLDW);
    const int colw = n0 + wave_n * WTN;
    float bc[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) bc[j] = p.bias ? p.bias[colw + j * 16 + fr] : 0.f;
    const int P = lane >> 3, k8 = lane & 7;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int h = 0; h < 2; ++h)
          stgw[(2 * fh + h) * SLDW + j * 16 + fr] = pack2bf(acc[i][j][2 * h] + bc[j], acc[i][j][2 * h + 1] + bc[j]);
      if (i == 0) __builtin_amdgcn_s_waitcnt(0x0F70);    // vmcnt(0): the bias loads (from here on only stores are pending)
      const uint4 t0 = *(const uint4*)(stgw + P * SLDW + k8 * 8), t1 = *(const uint4*)(stgw + P * SLDW + k8 * 8 + 4);
      const uint4 lo = make_uint4(__builtin_amdgcn_perm(t0.y, t0.x, 0x05040100u), __builtin_amdgcn_perm(t0.w, t0.z, 0x05040100u),
                                  __builtin_amdgcn_perm(t1.y, t1.x, 0x05040100u), __builtin_amdgcn_perm(t1.w, t1.z, 0x05040100u));
      const uint4 hi = make_uint4(__builtin_amdgcn_perm(t0.y, t0.x, 0x07060302u), __builtin_amdgcn_perm(t0.w, t0.z, 0x07060302u),
                                  __builtin_amdgcn_perm(t1.y, t1.x, 0x07060302u), __builtin_amdgcn_perm(t1.w, t1.z, 0x07060302u));
      const int row = m_base + wave_m * WTM + i * 16 + 2 * P;
      bf16_t* dst = Cb + (size_t)row * p.ldc + colw + k8 * 8;
      if (row < p.M) *(uint4*)dst = lo;
      if (row + 1 < p.M) *(uint4*)(dst + p.ldc) = hi;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (panel reads finished: the producers refill this stage behind B')
    __builtin_amdgcn_s_barrier();                        // B'(unit)
  }
}

}  // namespace ws4

// problems the one-wave-per-SIMD kernel takes: what gemm_ws.hip's packed bf16 store path takes, without band mode
bool ws_supported(const mvit_gemm_args& a);
bool ws_band_mode(const mvit_gemm_args& a);
bool ws4_supported(const mvit_gemm_args& a) {
  if (!ws_supported(a) || a.epi != MVIT_EPI_STORE) return false;
  if (a.flags & (MVIT_OUT_F32 | MVIT_ACCUM_BF16)) return false;
  return !ws_band_mode(a);
}

int launch_ws4(const mvit_gemm_args& a, hipStream_t s) {
  const int tiles = ((a.M + ws4::BM - 1) / ws4::BM) * (a.N / ws4::BN);
  const size_t lds = (size_t)ws4::NSTAGE * ws4::BUF_BYTES;
  int gx = gemm_num_cus();
  if (gx > tiles) gx = tiles;
  auto kern = ws4::gemm_ws4_kernel;
  static mvit_per_device_size raised;
  if (mvit_ensure_dynamic_lds((const void*)kern, lds, raised) != MVIT_OK) return MVIT_EINVAL;
  ws4::WsExtra xp;
  xp.grid_magic = mvit_div_magic((unsigned)gx);
  xp.pg_magic = mvit_div_magic((unsigned)(MVIT_WS_GROUP_M * (a.N / ws4::BN)));
  hipLaunchKernelGGL(kern, dim3(gx), dim3(64 * (ws4::NCW + ws4::NPW)), lds, s, a, xp);
  return MVIT_LAUNCH_CHECK();
}

}  // namespace mvit_gemm
