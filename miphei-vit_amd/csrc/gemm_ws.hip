// Wave-specialised bf16 MFMA GEMM for gfx950 (round 4): C[M,N] = A[M,K] * B[N,K]^T (+ A2 * B2^T), 256x128x64 tiles.
//
// Why a second dense kernel.  In gemm_kernel.hpp every wave of a block is loader, MFMA issuer and store issuer at once, and the three
// roles meet in ONE per-wave counter: `s_waitcnt vmcnt` counts the operand DMA (buffer_load ... lds) and the epilogue's global stores
// alike (gfx9 has no separate store counter, and loads / stores complete out of order with respect to each other, so hipcc waits for
// zero whenever both kinds are pending).  Consequences measured in rounds 1-3: (i) the next tile's first K step cannot start before the
// previous tile's stores have drained (fc1 + SwiGLU 139 vs 123 us bare, dfc2 + d(SwiGLU) 96 vs 64 us, proj / fc2 + residual 50 vs
// 42 us: ~2.6 ms of a 36.5 ms step with the MFMA pipe idle); (ii) each of the 6 DMA pieces a wave issues per K tile costs that wave
// ~100 issue cycles in the middle of its MFMA stream.
// Here the roles are split over the waves of a 768-thread block:
//   * waves 0-7  (consumers): 64x64 sub-tiles on v_mfma_f32_16x16x32_bf16, fragments double-buffered in registers, NO vector-memory
//     instruction in the K loop and no vmcnt wait anywhere on the way from one tile's epilogue into the next tile's K loop: their
//     stores drain while the next tile computes;
//   * waves 8-11 (producers): issue all operand DMA (12 pieces of 1 KiB per K tile each), wait for it with counted vmcnt and publish a
//     landed K tile through the block barrier the K step already has.  They run two K tiles ahead, across tile boundaries.
// One s_barrier per K tile (all 12 waves), one more per output tile (the epilogue panel lives in the LDS stage consumed last).
// Three waves per SIMD -> at most 168 VGPRs per lane (MI355X_MICROARCH.md, register table).
// Band mode (BAND instantiations): when the partly empty last tile row would cost a whole extra round of tiles, the whole 256-row tile
// rows run as whole rounds and the ragged band runs as 64-row x 128-column items, one per block, behind the block's last tile (same
// pipeline, 16-row wave sub-tiles): fc1 + SwiGLU of the batch-16 step 136.4 -> 129.8 us in the step.
//
// LDS image of a stage, fragment addressing, the XOR swizzle on the DMA source address and the tile order are those of
// gemm_kernel.hpp (so are the epilogue formulas: STORE / SWIGLU / RESID / DSWIGLU, vector paths only; everything else -- unaligned
// operands, N % 128, K % 64, split-K, convolution gathers -- stays on gemm_kernel.hpp, see mvit_gemm::ws_supported).
#include <type_traits>
#include "common.hpp"
#include "../../include/miphei_hip.h"

namespace mvit_gemm {
int gemm_num_cus();

namespace ws {
constexpr int BM = 256, BN = 128, BK = 64, NSTAGE = 3;
constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, BUF_BYTES = A_BYTES + B_BYTES;
constexpr int NCW = 8, NPW = 4;                      // consumer / producer waves
constexpr int PPW = (BM + BN) / 8 / NPW;             // DMA pieces (8 rows x 128 B) per producer wave per K tile = 12
constexpr int PA = BM / 8 / NPW, PB = BN / 8 / NPW;  // of which A / B pieces: 8 + 4
constexpr int WTM = 64, WTN = 64, TM = 4, TN = 4;
// The 16 MFMAs of a sub-step run in boustrophedon order over the wave's 4 x 4 accumulator blocks: every instruction shares one operand
// fragment with its predecessor (A along a row, B at the row turns -- in plain row-major order both operands change at a turn).  The
// loop is power-bound (docs/rounds/round5.md section 14): same cycles per K tile, launches 0.5-1 % shorter, step +0.4 % (section 17).
constexpr bool SNAKE = true;
constexpr int SLD = WTN + 4, SLAB = 16 * SLD;        // wave-private epilogue panel: 16 rows x 68 floats
constexpr int V = 8;
constexpr unsigned OOB = 0x80000000u;
// Single-round residual epilogue (round 5, flag 0x4000, see launch_ws): the accumulator tile is parked whole in LDS (the operand
// stages are dead by then), [BM][PARK_LD] floats; the producer waves, which fetched the tile's residual rows into their otherwise
// idle registers during the K loop, finish it
constexpr int PARK_LD = BN + 4;                      // row stride in floats: 16-byte aligned rows, conflict-free b128 reads
constexpr int RES_PASSES = BM / NPW / 2;             // a producer wave owns BM / NPW = 64 rows, two rows (2 x 128 floats) per pass = 32
#ifndef MVIT_WS_RES_PER_STEP
#define MVIT_WS_RES_PER_STEP 4
#endif
constexpr int RES_PER_STEP = MVIT_WS_RES_PER_STEP, RES_STEPS = RES_PASSES / RES_PER_STEP;   // residual loads ride on the first 32 / RES_PER_STEP K steps
static_assert((size_t)BM * PARK_LD * 4 <= (size_t)NSTAGE * BUF_BYTES, "parked tile must fit the operand stages");
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
// d(SwiGLU) epilogue operand as PSEUDO K TILES (round 5).  The saved pre-activation of a tile (256 rows x 256 packed bf16 = 128 KB)
// used to be requested by the consumer waves at the start of the epilogue, the matrix pipe idle until it arrived (tools/ws_timing.py:
// 13.5 k cycles per tile against 3.1 k for the plain store; more when the tensor comes from HBM, as it does in the backward pass).
// Now the producer waves DMA it into the operand ring behind the tile's last K tile, as NPSEUDO more "K tiles" -- one per 16-row
// accumulator slab, the slab's rows of all eight consumer waves (8 x 16 rows x 256 B = 32 KB of the 48 KB stage) -- two steps ahead
// like every other request, published through the same per-step barrier; the consumers' epilogue becomes NPSEUDO ring steps that
// read the operand from LDS and never wait on the vector-memory counter.  The wave-private accumulator panels (16 x 64 floats,
// XOR-swizzled instead of padded: exactly 4 KB) live in the 16 KB of the stage the pseudo tile leaves free (waves 0-3) and in
// 16 KB behind the ring (waves 4-7): 160 KB of LDS in all.
constexpr int NPSEUDO = 4, PS_WAVE_BYTES = 16 * 256, PS_BYTES = NCW * PS_WAVE_BYTES, PS_PPW = PS_BYTES / 1024 / NPW;   // 8 pieces per producer wave
constexpr int PANEL_BYTES = 16 * 64 * 4;
static_assert(PS_BYTES + 4 * PANEL_BYTES <= BUF_BYTES, "pseudo tile + four panels must fit a stage");
// -DMVIT_WS_TIMING (measurement build, tools/ws_timing.py): wave 0 (consumer) and wave 8 (producer) of every block stamp the phases
// of the block's FIRST work unit with s_memtime (shader cycles) and the block's begin / end with s_memrealtime (100 MHz, common to
// all CUs) into p.stats (16 x 8 bytes per block): where the fixed cost of a launch goes (launch skew, operand cold start, epilogue)
#ifdef MVIT_WS_TIMING
#define WS_STAMP(k, expr) { if (lane == 0 && prof) prof[k] = (long long)(expr); }
#define WS_CYC() __builtin_readcyclecounter()
#define WS_RT() __builtin_amdgcn_s_memrealtime()
#else
#define WS_STAMP(k, expr)
#endif
typedef __attribute__((address_space(3))) void* lds_ptr;
static_assert(PPW == PA + PB, "piece split");
static_assert((size_t)NCW * SLAB * 4 <= (size_t)BUF_BYTES, "epilogue panels must fit one stage");

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* ptr, unsigned bytes) {
  const unsigned long long v = (unsigned long long)ptr;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
  return __builtin_amdgcn_make_buffer_rsrc((void*)(((unsigned long long)hi << 32) | lo), 0,
                                           (int)__builtin_amdgcn_readfirstlane(bytes), 0x00020000);
}

// host-built constants of the block map (round 5: no run-time integer division in the prologue -- tools/ws_timing.py: ~2000 cycles
// from kernel entry to the first operand request, three divisions among them)
struct WsExtra {
  unsigned grid_magic;       // mvit_div_magic(gridDim.x)
  unsigned pg_magic;         // mvit_div_magic(GROUP_M * tiles_n), tiles_n = N / BN (band mode: same tiles_n)
};
// x / g for the group heights 1 .. 4
__device__ __forceinline__ int div_small(int x, int g) {
  return g == 4 ? x >> 2 : g == 2 ? x >> 1 : g == 1 ? x : (int)__umulhi((unsigned)x, 0x55555556u);
}
struct TileOrder {   // virtual tile id -> XCD-aware, grouped (8 tile rows x all columns) coordinates, as gemm_kernel.hpp
  int tiles_m, tiles_n, ntiles;
  unsigned pg_magic;
  __device__ __forceinline__ void get(int vt, int& m0, int& n0) const {
    const int q = ntiles >> 3, r = ntiles & 7, xcd = vt & 7;
    const int wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (vt >> 3);
#ifndef MVIT_WS_GROUP_M
#define MVIT_WS_GROUP_M 4
#endif
    constexpr int GROUP_M = MVIT_WS_GROUP_M;   // tile rows per group of the walk: an XCD's 32 concurrent tiles are a 4 x 8 patch (4 A panels + 8 B panels per K tile = the least L2-miss bytes); same-box step 482.4 / 481.2 (4) vs 477.7 / 475.4 (8) vs 464.1 / 463.2 (16) tiles/s
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

// BAND: the kernel also takes the 64-row items of the ragged band (see the work plan below); instantiated for the epilogues whose
// shapes need it (store, SwiGLU) -- the residual kernel sits at 168 VGPRs without it
template <int EPI, bool BAND>
__global__ __launch_bounds__(64 * (NCW + NPW)) void gemm_ws_kernel(const mvit_gemm_args p, const WsExtra xp) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // Work plan of a block: its tiles of the persistent walk (vt = blockIdx.x, + gridDim.x, ...) and, in BAND mode (flag 0x2000, set by
  // the dispatcher when the partly empty last tile row would cost a whole extra round of tiles: fc1 of the batch-16 step, 21 x 64 =
  // 1344 tiles = 5.25 rounds on 256 CUs), at most one 64-row x 128-column ITEM of the ragged band at the end of the walk: the full
  // tile rows are then exactly the whole rounds, and the band (144 rows at M = 5264: 3 x 64 items per tile column, 192 items) runs as
  // one short extra step per block -- a quarter of a tile's MFMAs, half of its DMA bytes -- instead of a sixth round on 64 of 256 CUs.
  const bool band = BAND && (p.flags & 0x2000) != 0;
  const int rows_full = band ? p.M / BM * BM : 0;                          // rows covered by whole 256-row tiles in band mode
  const int nq = band ? (p.M - rows_full + 63) / 64 : 0;                   // 64-row items per tile column of the band
  TileOrder ord;
  ord.tiles_m = band ? p.M / BM : (p.M + BM - 1) / BM;
  ord.tiles_n = p.N / BN;
  ord.ntiles = ord.tiles_m * ord.tiles_n;
  ord.pg_magic = xp.pg_magic;
  const int nk1 = p.K / BK;
  const int nk2 = p.A2 ? (p.K2 + BK - 1) / BK : 0;
  const int nk = nk1 + nk2;
  constexpr int NPS = (EPI == MVIT_EPI_DSWIGLU && !BAND) ? NPSEUDO : 0;   // pseudo K tiles behind every unit's real ones
  const int nks = nk + NPS;                                                // ring steps per unit
  const int my_tiles = ((int)blockIdx.x < ord.ntiles) ? (int)mvit_fast_div((unsigned)(ord.ntiles - 1 - (int)blockIdx.x), gridDim.x, xp.grid_magic) + 1 : 0;
  int item_id = (int)blockIdx.x;
  // band items in XCD-CONTIGUOUS order (block L runs on XCD L % 8): the nq items of one tile column (same B panel) on one XCD.  Measured
  // (fc1 + SwiGLU at batch 16): 122.4-128.8 us with the items spread evenly (item L on block L), 116.8-118.2 contiguous; ranking the XCDs
  // by measured speed on top of that added nothing (round 5) and its per-device global state is gone (round 6)
  if (BAND && (gridDim.x & 7) == 0) item_id = (int)(blockIdx.x & 7) * (int)(gridDim.x >> 3) + (int)(blockIdx.x >> 3);
  const bool has_item = BAND && item_id < nq * ord.tiles_n;                        // band item of this block: column id / nq, quarter id % nq
  const int item_col = div_small(item_id, nq > 0 ? nq : 1);                        // (nq <= 4: the band is lower than a tile)
  const int item_m = rows_full + (item_id - item_col * (nq > 0 ? nq : 1)) * 64, item_n = item_col * BN;
  const int G = (my_tiles + (has_item ? 1 : 0)) * nks;  // ring steps this block walks (global step index g)
#ifdef MVIT_WS_TIMING
  long long* prof = p.stats ? (long long*)p.stats + (size_t)blockIdx.x * 16 + (wave >= NCW ? 8 : 0) : nullptr;
  if (wave != 0 && wave != NCW) prof = nullptr;
  WS_STAMP(0, WS_RT())
  WS_STAMP(1, WS_CYC())
#endif

  if (wave >= NCW) {
    // ================================================================ producers
    const int pw = wave - NCW;
    // per-lane source offsets of this wave's pieces (tile independent: the tile enters through the descriptor's base; the K advance
    // rides on the scalar offset, which is outside the range check -- every row is a whole multiple of 128 B wide here)
    constexpr int PAQ = 64 / 8 / NPW;                    // A pieces per producer wave of a 64-row band item = 2
    unsigned voA[PA], voB[PB], voA2[PA], voB2[PB], voAq[PAQ], voAq2[PAQ];
    const int rl = lane >> 3, c8 = lane & 7;
#pragma unroll
    for (int j = 0; j < PA; ++j) {
      const int row = (pw * PA + j) * 8 + rl;
      const int cs = c8 ^ ((row >> 1) & 7);
      voA[j] = (unsigned)row * (unsigned)p.lda * 2u + (unsigned)cs * 16u;
      voA2[j] = (cs * 8 < p.K2) ? (unsigned)row * (unsigned)p.lda2 * 2u + (unsigned)cs * 16u : OOB;   // (K2 < 64: the valid chunks)
    }
#pragma unroll
    for (int j = 0; j < PAQ; ++j) {
      const int row = (pw * PAQ + j) * 8 + rl;
      const int cs = c8 ^ ((row >> 1) & 7);
      voAq[j] = (unsigned)row * (unsigned)p.lda * 2u + (unsigned)cs * 16u;
      voAq2[j] = (cs * 8 < p.K2) ? (unsigned)row * (unsigned)p.lda2 * 2u + (unsigned)cs * 16u : OOB;
    }
#pragma unroll
    for (int j = 0; j < PB; ++j) {
      const int row = (pw * PB + j) * 8 + rl;
      const int cs = c8 ^ ((row >> 1) & 7);
      voB[j] = (unsigned)row * (unsigned)p.ldb * 2u + (unsigned)cs * 16u;
      voB2[j] = (cs * 8 < p.K2) ? (unsigned)row * (unsigned)p.ldb2 * 2u + (unsigned)cs * 16u : OOB;
    }
    // d(SwiGLU) instantiation: the same offsets as two per-lane bases (even / odd piece: the swizzle term (row >> 1) & 7 only depends
    // on the piece's parity) + a uniform step per piece pair -- 4 registers instead of 12, the rest go to the operand prefetch
    constexpr bool COMPACT = EPI == MVIT_EPI_DSWIGLU;
    const unsigned stepA = 16u * (unsigned)p.lda * 2u, stepB = 16u * (unsigned)p.ldb * 2u;
    // (the step passes through an empty asm at every use: otherwise the adds are hoisted out of the step loop into eight registers again)
    auto opaque = [&](unsigned v) __attribute__((always_inline)) { asm volatile("" : "+s"(v)); return v; };
    auto offA = [&](int j) __attribute__((always_inline)) { return COMPACT ? voA[j & 1] + (unsigned)(j >> 1) * opaque(stepA) : voA[j]; };
    auto offB = [&](int j) __attribute__((always_inline)) { return COMPACT ? voB[j & 1] + (unsigned)(j >> 1) * opaque(stepB) : voB[j]; };
    // pseudo tiles: piece j of this wave = rows 4 q .. 4 q + 3 (q = j % 4) of consumer wave cw = 2 pw + j / 4's slab; lane l holds
    // row l / 16, 16-byte chunk l % 16 of the wave's 256 packed bytes (the slab's first row enters through the vector offset too:
    // the scalar offset is outside the range check)
    // (kept as two per-lane bases + a uniform row step: 2 registers instead of 8 -- the operand prefetch below needs them)
    unsigned voPb[2] = {0u, 0u};
    const unsigned voPq = 4u * (unsigned)p.ldaux * 2u;
    if constexpr (NPS > 0) {
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const int cw = 2 * pw + c;
        voPb[c] = (unsigned)((cw >> 1) * WTM + (lane >> 4)) * (unsigned)p.ldaux * 2u + (unsigned)(cw & 1) * 256u + (unsigned)(lane & 15) * 16u;
      }
    }
    auto voP = [&](int j) __attribute__((always_inline)) { unsigned q_ = voPq; asm volatile("" : "+s"(q_)); return voPb[j / 4] + (unsigned)(j % 4) * q_; };
    __amdgpu_buffer_rsrc_t rsA, rsB, rsA2, rsB2, rsP, rsPcur;
    // descriptors of work unit `idx` of this block (idx < my_tiles: a tile of the walk; idx == my_tiles: the band item)
    auto set_unit = [&](int idx) __attribute__((always_inline)) {
      int m0, n0, rows;
      if (idx < my_tiles) {
        ord.get(blockIdx.x + idx * gridDim.x, m0, n0);
        rows = BM;
      } else {
        m0 = item_m, n0 = item_n, rows = 64;
      }
      const unsigned vm = (unsigned)min(rows, p.M - m0), vn = (unsigned)min(BN, p.N - n0);
      // num_records = the valid rows of this unit: rows beyond M / N read as zero (hardware range check on the vector offset)
      rsA = make_rsrc((const bf16_t*)p.A + (size_t)m0 * p.lda, vm * (unsigned)p.lda * 2u);
      rsB = make_rsrc((const bf16_t*)p.B + (size_t)n0 * p.ldb, vn * (unsigned)p.ldb * 2u);
      rsA2 = make_rsrc(p.A2 ? (const bf16_t*)p.A2 + (size_t)m0 * p.lda2 : (const bf16_t*)p.A, p.A2 ? vm * (unsigned)p.lda2 * 2u : 0u);
      rsB2 = make_rsrc(p.B2 ? (const bf16_t*)p.B2 + (size_t)n0 * p.ldb2 : (const bf16_t*)p.B, p.B2 ? vn * (unsigned)p.ldb2 * 2u : 0u);
      if constexpr (NPS > 0)     // saved pre-activation rows of this tile, from its first packed column (2 n0): rows beyond M read as zero
        rsP = make_rsrc((const bf16_t*)p.aux + (size_t)m0 * p.ldaux + 2 * (size_t)n0, vm * (unsigned)p.ldaux * 2u);
    };
    // (generic lambda: the DMA builtin exists for the device target only, see gemm_kernel.hpp)
    auto issue = [&](int k, int stage, bool item, auto) __attribute__((always_inline)) {
      if constexpr (NPS > 0) {
        if (k >= nk) {           // pseudo tile k - nk: the 16-row slab (k - nk) of every consumer wave
          char* d = smem + stage * BUF_BYTES + pw * (2 * PS_WAVE_BYTES);
          const unsigned slab = (unsigned)(k - nk) * 16u * (unsigned)p.ldaux * 2u;
#pragma unroll
          for (int j = 0; j < PS_PPW; ++j) {
            const unsigned off = voP(j) + slab;   // (a local on purpose: with the expression as the builtin's argument the host pass of hipcc silently drops the kernel's stub)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsP, (lds_ptr)(d + j * 1024), 16, off, 0, 0, 0);
          }
          return;
        }
      }
      char* b = smem + stage * BUF_BYTES + A_BYTES + pw * PB * 1024;
      const int soff = k < nk1 ? k * (BK * 2) : 0;
      constexpr bool HAS_K2 = EPI != MVIT_EPI_DSWIGLU;      // (d(SwiGLU) never carries a second K range: ws_supported; its registers go to the operand prefetch)
      // second K range (k >= nk1; LoRA: A2 = t [M, 2r], B2 = [N, 2r]): chunks at or beyond K2 are zero (OOB offsets above; K2 <= 64)
      if (!BAND || !item) {
        char* a = smem + stage * BUF_BYTES + pw * PA * 1024;
        if (!HAS_K2 || k < nk1) {
#pragma unroll
          for (int j = 0; j < PA; ++j) {
            const unsigned off = offA(j);    // (a local: see the note at the pseudo-tile requests)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_ptr)(a + j * 1024), 16, off, soff, 0, 0);
          }
        } else {
#pragma unroll
          for (int j = 0; j < PA; ++j) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA2, (lds_ptr)(a + j * 1024), 16, voA2[j], 0, 0, 0);
        }
      } else {
        char* a = smem + stage * BUF_BYTES + pw * PAQ * 1024;      // the item's 64 A rows are rows 0..63 of the stage's A image
        if (k < nk1) {
#pragma unroll
          for (int j = 0; j < PAQ; ++j) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_ptr)(a + j * 1024), 16, voAq[j], soff, 0, 0);
        } else {
#pragma unroll
          for (int j = 0; j < PAQ; ++j) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA2, (lds_ptr)(a + j * 1024), 16, voAq2[j], 0, 0, 0);
        }
      }
      if (!HAS_K2 || k < nk1) {
#pragma unroll
        for (int j = 0; j < PB; ++j) {
          const unsigned off = offB(j);
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lds_ptr)(b + j * 1024), 16, off, soff, 0, 0);
        }
      } else {
#pragma unroll
        for (int j = 0; j < PB; ++j) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB2, (lds_ptr)(b + j * 1024), 16, voB2[j], 0, 0, 0);
      }
    };
    if constexpr (EPI == MVIT_EPI_RESID && !BAND) {
      if (p.flags & 0x4000) {
        // ---- single round (every block owns exactly one tile), residual epilogue on the producer side.
        // The residual tile (BM x BN f32 = 128 KB per block) is what the plain epilogue waits for with the matrix pipe idle: 252
        // blocks reach their epilogue together and ask for 32 MB at once (tools/ws_timing.py: 11-25 k cycles against 3 k for the
        // plain store).  Here each producer wave requests its 64 rows (32 x 16-byte loads per lane = 128 registers it otherwise
        // never uses) during the first RES_STEPS K steps, behind the step's operand DMA; the consumers park the accumulators in
        // LDS after the last K step; the producers add and store.  Loads return in order: the counted waits below allow exactly
        // the requests younger than the K tile that is about to be published.
        set_unit(0);
        int m0, n0;
        ord.get(blockIdx.x, m0, n0);
        const unsigned vm = (unsigned)min(BM, p.M - m0);
        const float* rsrc_p = p.aux ? (const float*)p.aux : (const float*)p.C;
        const int ldr = p.aux ? p.ldaux : p.ldc;
        // descriptors ranged to the tile's valid rows: rows beyond M load zeros / drop their stores (no row tests below)
        const __amdgpu_buffer_rsrc_t rsR = make_rsrc(rsrc_p + (size_t)m0 * ldr + n0, vm * (unsigned)ldr * 4u - (unsigned)0);
        const __amdgpu_buffer_rsrc_t rsC = make_rsrc((float*)p.C + (size_t)m0 * p.ldc + n0, vm * (unsigned)p.ldc * 4u);
        const int c4 = (lane & 31) * 4, rhalf = lane >> 5;
        const unsigned voR = (unsigned)rhalf * (unsigned)ldr * 4u + (unsigned)c4 * 4u;
        const unsigned voC = (unsigned)rhalf * (unsigned)p.ldc * 4u + (unsigned)c4 * 4u;
        u32x4 res[RES_PASSES];
        // (the row advance rides on the VECTOR offset: the scalar offset is outside the descriptor's range check)
        unsigned ro = voR + (unsigned)(pw * (BM / NPW)) * (unsigned)ldr * 4u;
        const unsigned rstep = 2u * (unsigned)ldr * 4u;
        auto issue_res = [&](int i0) __attribute__((always_inline)) {
#pragma unroll
          for (int j = 0; j < RES_PER_STEP; ++j) {
            res[i0 + j] = __builtin_amdgcn_raw_buffer_load_b128(rsR, ro, 0, 0);
            ro += rstep;
          }
        };
        int st = 0;
        auto next_stage = [&]() __attribute__((always_inline)) { const int c = st; st = st + 1 == NSTAGE ? 0 : st + 1; return c; };
        issue(0, next_stage(), false, 0);
        issue(1, next_stage(), false, 0);
        // column constants: two unconditional loads BEHIND the first operand requests (a null pointer reads the output's first
        // columns instead and the value is replaced below), counted in the waits like the residual requests
        const u32x4 gq = __builtin_amdgcn_raw_buffer_load_b128(make_rsrc(p.gamma ? (const void*)(p.gamma + n0) : (const void*)p.C, BN * 4u), (unsigned)c4 * 4u, 0, 0);
        const u32x4 bq = __builtin_amdgcn_raw_buffer_load_b128(make_rsrc(p.bias ? (const void*)(p.bias + n0) : (const void*)p.C, BN * 4u), (unsigned)c4 * 4u, 0, 0);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPW + 2) : "memory");
        WS_STAMP(2, WS_CYC())
        WS_STAMP(3, WS_CYC())
        __builtin_amdgcn_s_barrier();                    // B(-1) [no LDS reads pending]: producer
        // steps 0 .. RES_STEPS - 1 carry the residual requests (unrolled: `res` is a register array)
#pragma unroll
        for (int g = 0; g < RES_STEPS; ++g) {
          issue(g + 2, next_stage(), false, 0);
          issue_res(g * RES_PER_STEP);
          // in flight, oldest first: K tile g + 1 | residuals of step g - 1 | K tile g + 2 | residuals of step g
          if (g == 0)      // (K tile 1 | the two column-constant loads | K tile 2 | residuals of step 0)
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPW + RES_PER_STEP + 2) : "memory");
          else
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPW + 2 * RES_PER_STEP) : "memory");
          __builtin_amdgcn_s_barrier();                  // B(g) [no LDS reads pending]: producer
        }
        issue(RES_STEPS + 2, next_stage(), false, 0);    // (nk >= RES_STEPS + 3: launch_ws)
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPW + RES_PER_STEP) : "memory");
        __builtin_amdgcn_s_barrier();                    // B(RES_STEPS) [no LDS reads pending]: producer
        for (int g = RES_STEPS + 1; g < nk; ++g) {
          if (g + 2 < nk) {
            issue(g + 2, next_stage(), false, 0);
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPW) : "memory");
          } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          }
          __builtin_amdgcn_s_barrier();                  // B(g) [no LDS reads pending]: producer
        }
        __builtin_amdgcn_s_barrier();                    // B'(unit): the accumulator tile is parked [no LDS reads pending]: the producer's reads start below
        const f32x4 gam4 = p.gamma ? __builtin_bit_cast(f32x4, gq) : (f32x4){1.f, 1.f, 1.f, 1.f};
        const f32x4 bias4 = p.bias ? __builtin_bit_cast(f32x4, bq) : (f32x4){0.f, 0.f, 0.f, 0.f};
        const float* park = (const float*)smem + (size_t)(pw * (BM / NPW) + rhalf) * PARK_LD + c4;
        unsigned co = voC + (unsigned)(pw * (BM / NPW)) * (unsigned)p.ldc * 4u;
        const unsigned cstep = 2u * (unsigned)p.ldc * 4u;
#pragma unroll
        for (int i = 0; i < RES_PASSES; ++i) {
          const f32x4 a = *(const f32x4*)(park + (size_t)(2 * i) * PARK_LD);
          const f32x4 r = __builtin_bit_cast(f32x4, res[i]);
          f32x4 o;
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = r[e] + gam4[e] * (a[e] + bias4[e]);
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), rsC, co, 0, 0);
          co += cstep;
        }
        WS_STAMP(4, WS_CYC())
        WS_STAMP(5, WS_RT())
        return;
      }
    }
    if constexpr (NPS > 0) {
      constexpr int PRE_STEPS = NPS * PS_PPW / RES_PER_STEP;        // 32 requests of 16 bytes per lane, 4 per step: the first 8 steps of a unit
      if ((p.flags & 0x8000) && nk >= PRE_STEPS + 2) {
        // ---- d(SwiGLU), operand prefetched into the producers' registers (round 5).  The DMA form above requests a pseudo tile two
        // ring steps ahead: that hides the latency of the saved pre-activation but not its VOLUME -- 256 blocks reach their
        // epilogues together and ask for 32 MB within a few microseconds (tools/ws_timing.py: the epilogue took 13.4 k cycles per
        // tile with or without the sigmoid algebra, with the operand loaded by the consumers or DMA'd two steps ahead).  Here a
        // producer wave requests its share of the tile's operand (8 KB per slab, 32 x 16 bytes per lane = 128 registers it never
        // uses otherwise) during the first PRE_STEPS K steps of the tile, behind each step's operand DMA, and writes slab s into the
        // ring with ds_write at the step where the DMA form would have issued pseudo tile s: the 128 KB per tile now cross the
        // fabric spread over the whole K loop.  Consumers, ring protocol and LDS image are those of the DMA form.
        u32x4 pre[NPS * PS_PPW];
        int l_unit = 0, l_k = 0, l_stage = 0, l_g = 0;
        set_unit(0);
        auto advance = [&]() __attribute__((always_inline)) {
          ++l_g;
          l_stage = l_stage + 1 == NSTAGE ? 0 : l_stage + 1;
          if (++l_k == nks) {
            l_k = 0;
            ++l_unit;
            if (l_g < G) set_unit(l_unit);
          }
        };
        auto wait_allow = [&](int n) __attribute__((always_inline)) {       // at most n requests younger than what must have landed
          if (n >= PPW + 2 * RES_PER_STEP) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPW + 2 * RES_PER_STEP) : "memory");
          else if (n >= PPW + RES_PER_STEP) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPW + RES_PER_STEP) : "memory");
          else if (n >= PPW) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPW) : "memory");
          else if (n >= 2 * RES_PER_STEP) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * RES_PER_STEP) : "memory");
          else if (n >= RES_PER_STEP) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(RES_PER_STEP) : "memory");
          else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        };
        issue(l_k, l_stage, false, 0);
        advance();
        WS_STAMP(2, WS_CYC())
        issue(l_k, l_stage, false, 0);
        advance();
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPW) : "memory");
        WS_STAMP(3, WS_CYC())
        __builtin_amdgcn_s_barrier();                    // B(-1) [no LDS reads pending]: producer
        int g = 0, r_prev = 0;
        static_assert(PRE_STEPS == 8 && RES_PER_STEP * 2 == PS_PPW, "load steps are written out for 8 steps of 4 requests");
        // One ring step of the unit under consumption.  KC >= 0: load step KC (the look-ahead request is a real K tile, then four
        // operand requests of THIS unit into pre[4 KC ..]); SL >= 0: the look-ahead request is pseudo tile SL (ds_write from
        // pre[8 SL ..]); both -1: a plain step (real K tile of this or of the next unit).  Every register index is a literal.
        auto step = [&](auto kc_tag, auto sl_tag, auto tag) __attribute__((always_inline)) {
          constexpr int KC = decltype(kc_tag)::value, SL = decltype(sl_tag)::value;
          const bool req = g + 2 < G;
          int pieces = 0, r_cur = 0;
          if constexpr (SL >= 0) {
            // (the cursor stands on pseudo tile SL of the unit under consumption: always inside the walk)
            char* d = smem + l_stage * BUF_BYTES + pw * (2 * PS_WAVE_BYTES) + lane * 16;
#pragma unroll
            for (int j = 0; j < PS_PPW; ++j) *(u32x4*)(d + j * 1024) = pre[SL * PS_PPW + j];
            advance();
          } else if (req) {
            issue(l_k, l_stage, false, tag);
            advance();
            pieces = PPW;
          }
          if constexpr (KC >= 0) {
            const unsigned slab = (unsigned)(KC / 2) * 16u * (unsigned)p.ldaux * 2u;
#pragma unroll
            for (int j = 0; j < RES_PER_STEP; ++j) {
              const unsigned off = voP((KC % 2) * RES_PER_STEP + j) + slab;
              pre[KC * RES_PER_STEP + j] = __builtin_amdgcn_raw_buffer_load_b128(rsPcur, off, 0, 0);
            }
            r_cur = RES_PER_STEP;
          }
          // in flight, oldest first: request of step g + 1 | operand loads of step g - 1 | request of step g + 2 | operand loads of step g
          if (req || SL >= 0) wait_allow(r_prev + pieces + r_cur);
          else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          if constexpr (SL >= 0) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the slab is in LDS before a later barrier publishes it
          __builtin_amdgcn_s_barrier();                  // B(g) [no LDS reads pending]: producer (its ds_writes were waited for above)
          r_prev = r_cur;
          ++g;
        };
        using N1 = std::integral_constant<int, -1>;
        for (int t = 0; t < my_tiles; ++t) {
          // descriptor of the unit under CONSUMPTION (the cursor's rsP is already two steps ahead at a unit's end)
          {
            int m0, n0;
            ord.get(blockIdx.x + t * gridDim.x, m0, n0);
            const unsigned vm = (unsigned)min(BM, p.M - m0);
            rsPcur = make_rsrc((const bf16_t*)p.aux + (size_t)m0 * p.ldaux + 2 * (size_t)n0, vm * (unsigned)p.ldaux * 2u);
          }
          step(std::integral_constant<int, 0>{}, N1{}, 0);
          step(std::integral_constant<int, 1>{}, N1{}, 0);
          step(std::integral_constant<int, 2>{}, N1{}, 0);
          step(std::integral_constant<int, 3>{}, N1{}, 0);
          step(std::integral_constant<int, 4>{}, N1{}, 0);
          step(std::integral_constant<int, 5>{}, N1{}, 0);
          step(std::integral_constant<int, 6>{}, N1{}, 0);
          step(std::integral_constant<int, 7>{}, N1{}, 0);
          for (int k = PRE_STEPS; k < nk - 2; ++k) step(N1{}, N1{}, 0);           // the look-ahead request is K tile k + 2 < nk
          step(N1{}, std::integral_constant<int, 0>{}, 0);                          // steps nk - 2 .. nk + 1: pseudo tiles 0 .. 3
          step(N1{}, std::integral_constant<int, 1>{}, 0);
          step(N1{}, std::integral_constant<int, 2>{}, 0);
          step(N1{}, std::integral_constant<int, 3>{}, 0);
          step(N1{}, N1{}, 0);                                                      // steps nk + 2, nk + 3: the next unit's K tiles 0 and 1
          step(N1{}, N1{}, 0);
        }
        WS_STAMP(4, WS_CYC())
        WS_STAMP(5, WS_RT())
        return;
      }
    }
    // look-ahead cursor: the next K tile to request
    int l_unit = 0, l_k = 0, l_stage = 0, l_g = 0;
    bool l_item = my_tiles == 0;                          // the unit under the cursor is the band item (12 vs 6 pieces per request)
    if (G > 0) set_unit(0);
    bool last_item = false;                               // kind of the request made last (its pieces are what may stay in flight)
    bool last_pseudo = false;
    auto issue_next = [&](auto tag) __attribute__((always_inline)) {
      issue(l_k, l_stage, l_item, tag);
      last_item = l_item;
      last_pseudo = NPS > 0 && l_k >= nk;
      ++l_g;
      l_stage = l_stage + 1 == NSTAGE ? 0 : l_stage + 1;
      if (++l_k == nks) {
        l_k = 0;
        ++l_unit;
        l_item = l_unit >= my_tiles;
        if (l_g < G) set_unit(l_unit);
      }
    };
    // everything but the request just made has landed
    auto wait_older = [&]() __attribute__((always_inline)) {
      if (BAND && last_item)
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PAQ + PB) : "memory");
      else if (NPS > 0 && last_pseudo)
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PS_PPW) : "memory");
      else
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPW) : "memory");
    };
    if (G > 0) issue_next(0);
    WS_STAMP(2, WS_CYC())
    if (G > 1) {
      issue_next(0);
      wait_older();
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    WS_STAMP(3, WS_CYC())
    __builtin_amdgcn_s_barrier();                        // B(-1): K tile 0 has landed [no LDS reads pending]: producer waves never read LDS
    int g = 0;
    for (int t = 0; t < my_tiles + (has_item ? 1 : 0); ++t) {
      for (int k = 0; k < nks; ++k, ++g) {
        // stage (g + 2) % 3 = the one consumed in step g - 1: free since the barrier that ended it
        if (g + 2 < G) {
          issue_next(0);
          wait_older();                                  // K tile g + 1 is in LDS
        } else {
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();                    // B(g) [no LDS reads pending]: producer
      }
      if constexpr (NPS == 0)
        __builtin_amdgcn_s_barrier();                    // B'(unit): the consumers are done with the epilogue panel (= stage of step g - 1) [no LDS reads pending]: producer
    }
    WS_STAMP(4, WS_CYC())
    WS_STAMP(5, WS_RT())
    return;
  }

  // ================================================================== consumers
  const int wave_m = wave >> 1, wave_n = wave & 1;
  const int fr = lane & 15, fh = lane >> 4;            // fragment row, 16-byte K chunk inside a 32-wide sub-step
  const bool out_f32 = p.flags & MVIT_OUT_F32;
  float* Cf = (float*)p.C;
  bf16_t* Cb = (bf16_t*)p.C;
  constexpr int CPR = (EPI == MVIT_EPI_SWIGLU) ? 32 / V : WTN / V;   // lanes per output row
  constexpr int RPP = 64 / CPR;                                      // rows per pass: 8 (16 for SwiGLU)
  constexpr int NPASS = 16 / RPP;
  const int lc = (lane % CPR) * V, lr = lane / CPR;

  __builtin_amdgcn_s_barrier();                          // B(-1) [no LDS reads pending]: before the first fragment read
  WS_STAMP(2, WS_CYC())
#ifdef MVIT_WS_TIMING
  int units_done = 0;
#endif
  int stage = 0;
  // One work unit: a 256 x 128 tile (TMc = 4: 64-row wave sub-tiles) or a 64 x 128 item of the ragged band (TMc = 1: 16-row sub-tiles,
  // the A image uses rows 0..63 of the stage); m_base / n0 = its first row / column
  auto run_unit = [&](auto tm_tag, int m_base, int n0) __attribute__((always_inline)) {
    constexpr int TMc = decltype(tm_tag)::value;
    unsigned aoff[2], boff[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const unsigned sw = (unsigned)(((s * 4 + fh) ^ ((fr >> 1) & 7)) << 4);
      aoff[s] = (unsigned)(wave_m * (16 * TMc) + fr) * 128u + sw;
      boff[s] = (unsigned)A_BYTES + (unsigned)(wave_n * WTN + fr) * 128u + sw;
    }
    if (m_base + wave_m * (16 * TMc) >= p.M) {
      // this wave's rows lie entirely beyond M (ragged last tile row: rows 192.. of a tile with 144 valid rows at M = 5264): it only
      // keeps the block's barriers -- no fragment reads, no MFMAs on zero rows (the loops are power-limited: work that is not done is clock)
      for (int k = 0; k < nks; ++k) {
        __builtin_amdgcn_s_barrier();                    // B(g) [no LDS reads pending]: a wave without rows reads nothing
        stage = stage + 1 == NSTAGE ? 0 : stage + 1;
      }
      if constexpr (NPS == 0) __builtin_amdgcn_s_barrier();   // B'(unit) [no LDS reads pending]
      return;
    }
    f32x4 acc[TMc][TN];
#pragma unroll
    for (int i = 0; i < TMc; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    bf16x8 fa[2][TMc], fb[2][TN];
    {
      const char* cur = smem + stage * BUF_BYTES;
#pragma unroll
      for (int i = 0; i < TMc; ++i) fa[0][i] = *(const bf16x8*)(cur + aoff[0] + i * 2048);
#pragma unroll
      for (int j = 0; j < TN; ++j) fb[0][j] = *(const bf16x8*)(cur + boff[0] + j * 2048);
    }
    // One K tile: sub-step 0 (16 MFMAs) carries the eight fragment reads of sub-step 1, one behind every other MFMA, in the order
    // sub-step 1 consumes them (a0, b0..b3, a1..a3); sub-step 1 runs six MFMAs, hands the stage over (its reads are back: lgkmcnt(0),
    // block barrier = the next K tile has landed, see the producers) and carries the first fragments of the next stage on the rest.
    auto read_sub = [&](const char* base, int s, int r, bf16x8 (&xa)[TMc], bf16x8 (&xb)[TN]) __attribute__((always_inline)) {
      if (r == 0)
        xa[0] = *(const bf16x8*)(base + aoff[s]);
      else if (r <= TN)
        xb[r - 1] = *(const bf16x8*)(base + boff[s] + (r - 1) * 2048);
      else
        xa[r - TN] = *(const bf16x8*)(base + aoff[s] + (r - TN) * 2048);
    };
    constexpr int NM = TMc * TN, NR = TMc + TN;          // MFMAs / fragment reads per sub-step
    auto kstep = [&](auto more_tag) __attribute__((always_inline)) {
      constexpr bool more = decltype(more_tag)::value;   // another K tile of THIS output tile follows
      const char* cur = smem + stage * BUF_BYTES;
#pragma unroll
      for (int m = 0; m < TMc * TN; ++m) {
        const int i = m / TN, j = (SNAKE && ((m / TN) & 1)) ? TN - 1 - m % TN : m % TN;
        acc[i][j] = mvit_mfma16(fa[0][i], fb[0][j], acc[i][j], 0, 0, 0);
        {
#pragma unroll
          for (int r = 0; r < NR; ++r)      // reads spread over the MFMAs: ceil((m + 1) NR / NM) issued by MFMA m
            if (r >= (m * NR + NM - 1) / NM && r < ((m + 1) * NR + NM - 1) / NM) read_sub(cur, 1, r, fa[1], fb[1]);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      constexpr int HO = TMc == 4 ? 6 : 1;   // MFMAs of sub-step 1 ahead of the hand-over (measured: 2 / 6 / 10)
#pragma unroll
      for (int m = 0; m < HO; ++m) {
        const int i = m / TN, j = (SNAKE && ((m / TN) & 1)) ? TN - 1 - m % TN : m % TN;
        acc[i][j] = mvit_mfma16(fa[1][i], fb[1][j], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();                      // B(g)
      __builtin_amdgcn_sched_barrier(0);
      stage = stage + 1 == NSTAGE ? 0 : stage + 1;
      const char* nxt = smem + stage * BUF_BYTES;
#pragma unroll
      for (int m = HO; m < TMc * TN; ++m) {
        const int i = m / TN, j = (SNAKE && ((m / TN) & 1)) ? TN - 1 - m % TN : m % TN;
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
    {
    for (int k = 0; k + 1 < nk; ++k) kstep(std::true_type{});
    kstep(std::false_type{});
    }
#ifdef MVIT_WS_TIMING
    if (units_done == 0) WS_STAMP(3, WS_CYC())
#endif
    if constexpr (EPI == MVIT_EPI_RESID && !BAND && TMc == 4) {
      if (p.flags & 0x4000) {
        // single round: park the whole accumulator sub-tile (every wave is past its last fragment read -- the barrier inside the
        // last K step -- and no DMA is in flight: all three stages are free), the producer waves finish the tile (see above)
        float* park = (float*)smem + (size_t)(wave_m * WTM + 4 * fh) * PARK_LD + wave_n * WTN + fr;
#pragma unroll
        for (int i = 0; i < TMc; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) park[(i * 16 + r4) * PARK_LD + j * 16] = acc[i][j][r4];
#ifdef MVIT_WS_TIMING
        if (units_done == 0) WS_STAMP(4, WS_CYC())
        ++units_done;
#endif
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the parked values are in LDS before the hand-over
        __builtin_amdgcn_s_barrier();                        // B'(unit)
        return;
      }
    }
    if constexpr (NPS > 0 && TMc == 4) {
      // ---- d(SwiGLU) on pseudo K tiles: `stage` names pseudo tile 0 (landed: the barrier inside the last K step published it).
      // Step i: park the 16-row accumulator slab i, read it back row-wise with the slab's saved pre-activation from the stage,
      // form d(silu(a) b) * dG, store; lgkmcnt(0) + the step's barrier release the stage.  No vector-memory load anywhere.
      const int colw = n0 + wave_n * WTN, col = colw + lc;
      const int ca = ((col >> 5) << 6) + (col & 31);           // packed position of gate columns col .. col + 7 (a; b at + 32)
      const int k8 = lane & 7;
      const unsigned pre_off = (unsigned)wave * PS_WAVE_BYTES + (unsigned)lr * 256u + (k8 < 4 ? 16u * k8 : 128u + 16u * (k8 - 4));
#pragma unroll
      for (int i = 0; i < NPS; ++i) {
        const char* ps = smem + stage * BUF_BYTES;
        float* pan = (float*)(wave < 4 ? smem + stage * BUF_BYTES + PS_BYTES + wave * PANEL_BYTES
                                       : smem + NSTAGE * BUF_BYTES + (wave - 4) * PANEL_BYTES);
        // park (C/D layout of the 16x16 MFMA: col = lane & 15, row = 4 * (lane >> 4) + register); 16-byte chunk c of row r sits at
        // chunk c ^ (r & 1): the row-wise b128 reads below then touch 16 distinct chunks per lane group
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int r4 = 0; r4 < 4; ++r4) {
            const int row = 4 * fh + r4;
            pan[row * 64 + ((((4 * j + (fr >> 2)) ^ (r4 & 1))) << 2) + (fr & 3)] = acc[i][j][r4];     // (row & 1 == r4 & 1)
          }
#pragma unroll
        for (int it = 0; it < NPASS; ++it) {
          const int rl_ = it * RPP + lr;                       // RPP = 8: it * 8 + lane / 8
          const int par = rl_ & 1;
          const f32x4 t0 = *(const f32x4*)(pan + rl_ * 64 + (((2 * k8) ^ par) << 2));
          const f32x4 t1 = *(const f32x4*)(pan + rl_ * 64 + (((2 * k8 + 1) ^ par) << 2));
          const uint4 qa = *(const uint4*)(ps + pre_off + it * (RPP * 256));
          const uint4 qb = *(const uint4*)(ps + pre_off + it * (RPP * 256) + 64);
          const float v[V] = {t0[0], t0[1], t0[2], t0[3], t1[0], t1[1], t1[2], t1[3]};
          const uint32_t ua[4] = {qa.x, qa.y, qa.z, qa.w}, ub[4] = {qb.x, qb.y, qb.z, qb.w};
          float da[V], db[V];
#pragma unroll
          for (int e = 0; e < V; ++e) {
            const float a_ = (e & 1) ? hi16f(ua[e >> 1]) : lo16f(ua[e >> 1]);
            const float b_ = (e & 1) ? hi16f(ub[e >> 1]) : lo16f(ub[e >> 1]);
            const float sg = sigmoidf_(a_);
            const float vs = v[e] * sg;
            da[e] = vs * b_ * (1.f + a_ * (1.f - sg));
            db[e] = vs * a_;
          }
          const int row = m_base + wave_m * WTM + i * 16 + rl_;
          if (row < p.M) {
            bf16_t* dst = Cb + (size_t)row * p.ldc + ca;
            *(uint4*)dst = make_uint4(pack2bf(da[0], da[1]), pack2bf(da[2], da[3]), pack2bf(da[4], da[5]), pack2bf(da[6], da[7]));
            *(uint4*)(dst + 32) = make_uint4(pack2bf(db[0], db[1]), pack2bf(db[2], db[3]), pack2bf(db[4], db[5]), pack2bf(db[6], db[7]));
          }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // this wave's reads of the pseudo tile and of its panel are back
        __builtin_amdgcn_s_barrier();                          // B(g): the stage may be refilled, the next step's tile has landed
        __builtin_amdgcn_sched_barrier(0);
        stage = stage + 1 == NSTAGE ? 0 : stage + 1;
      }
#ifdef MVIT_WS_TIMING
      if (units_done == 0) WS_STAMP(4, WS_CYC())
      ++units_done;
#endif
      return;
    }
    // `stage` now names the first K tile of the NEXT output tile; the stage consumed last (every consumer is past its reads: the
    // barrier inside the last step) holds the epilogue panels until B'
    const int last = stage == 0 ? NSTAGE - 1 : stage - 1;
    float* stg = (float*)(smem + last * BUF_BYTES) + (size_t)wave * SLAB;

    // ---------------------------------------------------------------- epilogue (vector paths only: N % 128 == 0, aligned operands)
    if constexpr (EPI == MVIT_EPI_STORE) {
      if (!out_f32 && !(p.flags & MVIT_ACCUM_BF16) && !(p.flags & 0x10000)) {
        // Plain bf16 store through a HALF-SIZE panel (round 5).  The f32 panel epilogue is LDS-write bound: 8 waves park 128 KB per
        // tile at 64 B/clk (tools/ws_timing.py: 3.1 k cycles per tile).  A lane's four accumulator registers of a 16x16 block are four
        // consecutive ROWS of one column, so rows (2 P, 2 P + 1) round to bf16 and share one dword: 32 ds_write_b32 per lane
        // instead of 64; a lane reads 8 such dwords back (8 columns x 2 rows), splits low / high halves with v_perm and stores one
        // 16-byte group per row.  Bias is added before the rounding, as in the f32 form (same bits).
        constexpr int SLDW = WTN + 4;                              // panel row (one row PAIR) in dwords
        uint32_t* stgw = (uint32_t*)(smem + last * BUF_BYTES) + (size_t)wave * (8 * SLDW);
        const int colw = n0 + wave_n * WTN;
        float bc[TN];
#pragma unroll
        for (int j = 0; j < TN; ++j) bc[j] = p.bias ? p.bias[colw + j * 16 + fr] : 0.f;
        const int P = lane >> 3, k8 = lane & 7;
#pragma unroll
        for (int i = 0; i < TMc; ++i) {
#pragma unroll
          for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int h = 0; h < 2; ++h)
              stgw[(2 * fh + h) * SLDW + j * 16 + fr] = pack2bf(acc[i][j][2 * h] + bc[j], acc[i][j][2 * h + 1] + bc[j]);
          if (i == 0) __builtin_amdgcn_s_waitcnt(0x0F70);          // vmcnt(0): the bias loads (from here on only stores are pending)
          const uint4 t0 = *(const uint4*)(stgw + P * SLDW + k8 * 8), t1 = *(const uint4*)(stgw + P * SLDW + k8 * 8 + 4);
          // low halves = row 2 P, high halves = row 2 P + 1 (v_perm_b32 selectors: bytes 0,1 of both / bytes 2,3 of both)
          const uint4 lo = make_uint4(__builtin_amdgcn_perm(t0.y, t0.x, 0x05040100u), __builtin_amdgcn_perm(t0.w, t0.z, 0x05040100u),
                                      __builtin_amdgcn_perm(t1.y, t1.x, 0x05040100u), __builtin_amdgcn_perm(t1.w, t1.z, 0x05040100u));
          const uint4 hi = make_uint4(__builtin_amdgcn_perm(t0.y, t0.x, 0x07060302u), __builtin_amdgcn_perm(t0.w, t0.z, 0x07060302u),
                                      __builtin_amdgcn_perm(t1.y, t1.x, 0x07060302u), __builtin_amdgcn_perm(t1.w, t1.z, 0x07060302u));
          const int row = m_base + wave_m * (16 * TMc) + i * 16 + 2 * P;
          bf16_t* dst = Cb + (size_t)row * p.ldc + colw + k8 * 8;
          if (row < p.M) *(uint4*)dst = lo;
          if (row + 1 < p.M) *(uint4*)(dst + p.ldc) = hi;
        }
#ifdef MVIT_WS_TIMING
        if (units_done == 0) WS_STAMP(4, WS_CYC())
        ++units_done;
#endif
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");         // (panel reads finished: the producers refill this stage behind B')
        __builtin_amdgcn_s_barrier();                              // B'(unit)
        return;
      }
    }
    const int colw = n0 + wave_n * WTN;
    const int col = colw + lc;
    float bias[V], bias2[V], gam[V];
#pragma unroll
    for (int e = 0; e < V; ++e) bias[e] = bias2[e] = 0.f, gam[e] = 1.f;
    if (p.bias) {
      const float4 t0 = *(const float4*)(p.bias + col), t1 = *(const float4*)(p.bias + col + 4);
      bias[0] = t0.x, bias[1] = t0.y, bias[2] = t0.z, bias[3] = t0.w, bias[4] = t1.x, bias[5] = t1.y, bias[6] = t1.z, bias[7] = t1.w;
      if constexpr (EPI == MVIT_EPI_SWIGLU) {
        const float4 u0 = *(const float4*)(p.bias + col + 32), u1 = *(const float4*)(p.bias + col + 36);
        bias2[0] = u0.x, bias2[1] = u0.y, bias2[2] = u0.z, bias2[3] = u0.w, bias2[4] = u1.x, bias2[5] = u1.y, bias2[6] = u1.z, bias2[7] = u1.w;
      }
    }
    if (EPI == MVIT_EPI_RESID && p.gamma) {
      const float4 t0 = *(const float4*)(p.gamma + col), t1 = *(const float4*)(p.gamma + col + 4);
      gam[0] = t0.x, gam[1] = t0.y, gam[2] = t0.z, gam[3] = t0.w, gam[4] = t1.x, gam[5] = t1.y, gam[6] = t1.z, gam[7] = t1.w;
    }
    constexpr bool AUX = (EPI == MVIT_EPI_RESID || EPI == MVIT_EPI_DSWIGLU);
    // per-element operands of the whole sub-tile (residual stream / saved pre-activation) are requested up front: the fragment
    // registers are dead here, and a load requested behind a store would wait for that store too (one counter, see the header)
    uint4 pre[TMc][NPASS][2];
    if constexpr (AUX) {
#pragma unroll
      for (int i = 0; i < TMc; ++i)
#pragma unroll
        for (int it = 0; it < NPASS; ++it) {
          const int row = m_base + wave_m * (16 * TMc) + i * 16 + it * RPP + lr;
          const int rr = row < p.M ? row : p.M - 1;      // (clamped: an unconditional load, results of rows >= M are not stored)
          if constexpr (EPI == MVIT_EPI_RESID) {
            const float* rp = p.aux ? (const float*)p.aux + (size_t)rr * p.ldaux + col : Cf + (size_t)rr * p.ldc + col;
            pre[i][it][0] = ((const uint4*)rp)[0];
            pre[i][it][1] = ((const uint4*)rp)[1];
          } else {
            const bf16_t* u = (const bf16_t*)p.aux + (size_t)rr * p.ldaux + (((col >> 5) << 6) + (col & 31));
            pre[i][it][0] = *(const uint4*)u;
            pre[i][it][1] = *(const uint4*)(u + 32);
          }
        }
    }
    auto panel8 = [&](int rl_, int c0, float (&o)[V]) __attribute__((always_inline)) {
      const float4 t0 = *(const float4*)(stg + rl_ * SLD + c0), t1 = *(const float4*)(stg + rl_ * SLD + c0 + 4);
      o[0] = t0.x, o[1] = t0.y, o[2] = t0.z, o[3] = t0.w, o[4] = t1.x, o[5] = t1.y, o[6] = t1.z, o[7] = t1.w;
    };
    auto un8bf = [&](const uint4& tq, float (&o)[V]) __attribute__((always_inline)) {
      const uint32_t u[4] = {tq.x, tq.y, tq.z, tq.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) o[2 * e] = lo16f(u[e]), o[2 * e + 1] = hi16f(u[e]);
    };
    auto pk8 = [&](const float (&o)[V]) __attribute__((always_inline)) {
      return make_uint4(pack2bf(o[0], o[1]), pack2bf(o[2], o[3]), pack2bf(o[4], o[5]), pack2bf(o[6], o[7]));
    };
#pragma unroll
    for (int i = 0; i < TMc; ++i) {
      // park rows 16 i .. 16 i + 15 of the sub-tile (C/D layout of the 16x16 MFMA: col = lane & 15, row = 4 * (lane >> 4) + register)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) stg[(r4 + 4 * fh) * SLD + j * 16 + fr] = acc[i][j][r4];
      // ONE wait for every load of this epilogue (column constants, per-element operands), in front of the first store: from here
      // on only stores are pending, and nothing below waits for them (s_waitcnt through the builtin, so that hipcc's own counter
      // bookkeeping sees it: with the loads under `if (p.bias)` it otherwise re-waits -- for zero, stores included -- at every use)
      if (i == 0) __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0)
#pragma unroll
      for (int it = 0; it < NPASS; ++it) {
        const int rl_ = it * RPP + lr;
        const int row = m_base + wave_m * (16 * TMc) + i * 16 + rl_;
        const bool rok = row < p.M;
        if constexpr (EPI == MVIT_EPI_SWIGLU) {
          float a_[V], b_[V], g_[V];
          panel8(rl_, lc, a_);
          panel8(rl_, lc + 32, b_);
#pragma unroll
          for (int e = 0; e < V; ++e) a_[e] += bias[e], b_[e] += bias2[e];
#pragma unroll
          for (int e = 0; e < V; ++e) g_[e] = a_[e] * sigmoidf_(a_[e]) * b_[e];
          if (rok) {
            if (p.aux) {
              bf16_t* aux = (bf16_t*)p.aux + (size_t)row * p.ldaux + col;
              *(uint4*)aux = pk8(a_);
              *(uint4*)(aux + 32) = pk8(b_);
            }
            *(uint4*)(Cb + (size_t)row * p.ldc + ((colw >> 1) + lc)) = pk8(g_);
          }
        } else {
          float v[V];
          panel8(rl_, lc, v);
#pragma unroll
          for (int e = 0; e < V; ++e) v[e] += bias[e];
          if constexpr (EPI == MVIT_EPI_STORE) {
            const size_t o = (size_t)row * p.ldc + col;
            if (rok) {
              if (out_f32) {
                ((float4*)(Cf + o))[0] = make_float4(v[0], v[1], v[2], v[3]);
                ((float4*)(Cf + o))[1] = make_float4(v[4], v[5], v[6], v[7]);
              } else {
                if (p.flags & MVIT_ACCUM_BF16) {
                  float old[V];
                  un8bf(*(const uint4*)(Cb + o), old);
#pragma unroll
                  for (int e = 0; e < V; ++e) v[e] += old[e];
                }
                *(uint4*)(Cb + o) = pk8(v);
              }
            }
          } else if constexpr (EPI == MVIT_EPI_RESID) {
            const uint4 t0 = pre[i][it][0], t1 = pre[i][it][1];
            float r_[V] = {__uint_as_float(t0.x), __uint_as_float(t0.y), __uint_as_float(t0.z), __uint_as_float(t0.w),
                           __uint_as_float(t1.x), __uint_as_float(t1.y), __uint_as_float(t1.z), __uint_as_float(t1.w)};
#pragma unroll
            for (int e = 0; e < V; ++e) r_[e] += gam[e] * v[e];     // (DropPath row factors: gemm_kernel.hpp, see ws_supported)
            if (rok) {
              float* dst = Cf + (size_t)row * p.ldc + col;
              ((float4*)dst)[0] = make_float4(r_[0], r_[1], r_[2], r_[3]);
              ((float4*)dst)[1] = make_float4(r_[4], r_[5], r_[6], r_[7]);
            }
          } else if constexpr (EPI == MVIT_EPI_DSWIGLU) {
            // gate columns col..col+7 live at packed positions ca.. (a) and ca+32.. (b) of the saved pre-activation
            const int ca = ((col >> 5) << 6) + (col & 31);
            float a_[V], b_[V], da[V], db[V];
            un8bf(pre[i][it][0], a_);
            un8bf(pre[i][it][1], b_);
#pragma unroll
            for (int e = 0; e < V; ++e) {
              const float sg = sigmoidf_(a_[e]);
              const float vs = v[e] * sg;
              da[e] = vs * b_[e] * (1.f + a_[e] * (1.f - sg));
              db[e] = vs * a_[e];
            }
            if (rok) {
              bf16_t* dst = Cb + (size_t)row * p.ldc + ca;
              *(uint4*)dst = pk8(da);
              *(uint4*)(dst + 32) = pk8(db);
            }
          }
        }
      }
    }
#ifdef MVIT_WS_TIMING
    if (units_done == 0) WS_STAMP(4, WS_CYC())           // (stores issued, not drained)
    ++units_done;
#endif
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (panel reads finished, not just issued: the producers refill this stage behind B')
    __builtin_amdgcn_s_barrier();                        // B'(unit): the panel stage may be refilled
  };
  int vt = blockIdx.x;
  for (int t = 0; t < my_tiles; ++t, vt += gridDim.x) {
    int m0, n0;
    ord.get(vt, m0, n0);
    run_unit(std::integral_constant<int, 4>{}, m0, n0);
  }
  if constexpr (BAND) {
    if (has_item) run_unit(std::integral_constant<int, 1>{}, item_m, item_n);
  }
#ifdef MVIT_WS_TIMING
  WS_STAMP(5, WS_CYC())
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // stores drained
  WS_STAMP(6, WS_CYC())
  WS_STAMP(7, WS_RT())
#endif
}

}  // namespace ws

// problems the wave-specialised kernel takes (everything else stays on gemm_kernel.hpp)
bool ws_supported(const mvit_gemm_args& a) {
  if (a.amode != MVIT_A_DENSE || a.ksplit > 1 || (a.flags & (MVIT_ATOMIC | 0x400 | 0x800 | 0x2000 | 0x4000 | 0x8000 | 0x10000 | 0x20000))) return false;
  if (a.epi != MVIT_EPI_STORE && a.epi != MVIT_EPI_SWIGLU && a.epi != MVIT_EPI_RESID && a.epi != MVIT_EPI_DSWIGLU) return false;
  if (a.M < 1024 || (a.N % 128) || (a.K % 64) || a.K < 64) return false;
  if (a.A2 && (a.K2 > 64 || a.K2 <= 0)) return false;
  if (a.A2 && a.epi == MVIT_EPI_DSWIGLU) return false;
  if (a.epi == MVIT_EPI_RESID && (!(a.flags & MVIT_OUT_F32) || a.rowscale)) return false;   // (DropPath row factors: 8 more registers than the 168 this kernel has)
  if (a.epi != MVIT_EPI_STORE && a.epi != MVIT_EPI_RESID && (a.flags & MVIT_OUT_F32)) return false;
  if (a.epi != MVIT_EPI_STORE && (a.flags & MVIT_ACCUM_BF16)) return false;
  // mvit_fast_div: tile index x divisor below 2^32 (tiles < 2^20 at these sizes; checked, not assumed)
  {
    const long long tiles = (long long)((a.M + 255) / 256) * (a.N / 128), pg = 4ll * (a.N / 128);
    if (tiles * (pg > 256 ? pg : 256) >= 0xffffffffll) return false;
  }
  // 32-bit byte offsets inside a tile's descriptor range
  const long long maxld = a.lda > a.ldb ? a.lda : a.ldb;
  if (256ll * maxld * 2 >= 0x7fffffffll) return false;
  // ... and inside the output / epilogue-operand ranges: the single-round residual path and the d(SwiGLU) operand build descriptor
  // offsets from ldc (f32: x 4) and ldaux (f32 residual x 4, packed bf16 pre-activation x 2)
  if (256ll * a.ldc * ((a.flags & MVIT_OUT_F32) ? 4 : 2) >= 0x7fffffffll) return false;
  if (a.aux && 256ll * a.ldaux * (a.epi == MVIT_EPI_RESID ? 4 : 2) >= 0x7fffffffll) return false;
  return true;
}

template <int EPI, bool BAND>
static int launch_ws_one(const mvit_gemm_args& a, hipStream_t s) {
  const int tiles = ((a.M + ws::BM - 1) / ws::BM) * (a.N / ws::BN);
  const size_t lds = (size_t)ws::NSTAGE * ws::BUF_BYTES + ((EPI == MVIT_EPI_DSWIGLU && !BAND) ? 4 * ws::PANEL_BYTES : 0);   // (+ panels of waves 4-7)
  int gx = gemm_num_cus();
  if (gx > tiles) gx = tiles;
  auto kern = ws::gemm_ws_kernel<EPI, BAND>;
  static mvit_per_device_size raised;
  if (mvit_ensure_dynamic_lds((const void*)kern, lds, raised) != MVIT_OK) return MVIT_EINVAL;
  ws::WsExtra xp;
  xp.grid_magic = mvit_div_magic((unsigned)gx);
  xp.pg_magic = mvit_div_magic((unsigned)(MVIT_WS_GROUP_M * (a.N / ws::BN)));
  hipLaunchKernelGGL(kern, dim3(gx), dim3(64 * (ws::NCW + ws::NPW)), lds, s, a, xp);
  return MVIT_LAUNCH_CHECK();
}

// Band mode pays when the partly empty last tile row is what pushes the launch into another round of tiles: whole tile rows = whole
// rounds, band items <= one per block.  (fc1 of the batch-16 training step: M = 5264, N = 8192 -> 1280 tiles = 5 rounds + 192 items.)
bool ws_band_mode(const mvit_gemm_args& a) {
  if (a.epi != MVIT_EPI_STORE && a.epi != MVIT_EPI_SWIGLU) return false;
  const long long cus = gemm_num_cus(), nt = a.N / ws::BN, rf = a.M / ws::BM;
  const int tail = a.M - (int)rf * ws::BM;
  if (tail <= 0 || rf <= 0) return false;
  const long long rounds_all = ((rf + 1) * nt + cus - 1) / cus, rounds_full = (rf * nt + cus - 1) / cus;
  const long long items = (long long)((tail + 63) / 64) * nt;
  return rounds_full < rounds_all && (rf * nt) % cus == 0 && items <= cus;
}

int launch_ws(const mvit_gemm_args& a0, hipStream_t s, int band_knob, int resid_single_knob, int dsw_reg_knob, int pack_store_knob) {
  mvit_gemm_args a = a0;
  a.flags &= ~(0x2000 | 0x4000 | 0x8000 | 0x10000 | 0x20000);
  if (a.epi == MVIT_EPI_STORE && !pack_store_knob) a.flags |= 0x10000;   // (measurement: the f32-panel store epilogue)
  if (band_knob && ws_band_mode(a)) {
    a.flags |= 0x2000;
    if (band_knob == 2) a.flags |= 0x20000;                              // (measurement: band items spread evenly over the XCDs)
    return a.epi == MVIT_EPI_STORE ? launch_ws_one<MVIT_EPI_STORE, true>(a, s) : launch_ws_one<MVIT_EPI_SWIGLU, true>(a, s);
  }
  // single-round residual epilogue on the producer waves (every block owns one tile, enough K steps to carry the residual requests)
  if (a.epi == MVIT_EPI_RESID && resid_single_knob && (long long)((a.M + ws::BM - 1) / ws::BM) * (a.N / ws::BN) <= gemm_num_cus() &&
      a.K / ws::BK >= ws::RES_STEPS + 3 && !a.A2)
    a.flags |= 0x4000;
  if (a.epi == MVIT_EPI_DSWIGLU && dsw_reg_knob) a.flags |= 0x8000;     // operand of the d(SwiGLU) epilogue through the producers' registers
  switch (a.epi) {
    case MVIT_EPI_STORE: return launch_ws_one<MVIT_EPI_STORE, false>(a, s);
    case MVIT_EPI_SWIGLU: return launch_ws_one<MVIT_EPI_SWIGLU, false>(a, s);
    case MVIT_EPI_RESID: return launch_ws_one<MVIT_EPI_RESID, false>(a, s);
    case MVIT_EPI_DSWIGLU: return launch_ws_one<MVIT_EPI_DSWIGLU, false>(a, s);
    default: return MVIT_EINVAL;
  }
}

}  // namespace mvit_gemm
