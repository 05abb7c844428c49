// Fused multi-head self-attention (flash style) for gfx950, head_dim <= 64, non-causal, no mask.
//
// Reads the packed qkv projection [B, N, 3, H, Dh] (bf16) in place, as timm's Attention does after
// qkv(x).reshape(B,N,3,H,Dh) (reached through src/generators/foundation_models.py:53-57), and writes
// O [B, N, H*Dh].  Everything is computed in the transposed form so that the softmax axis is lane-local:
//     S^T[key][q] = K Q^T           (A = K rows from LDS, B = Q fragment held in registers)
//     O^T[d][q]  += V^T P^T         (A = V^T via ds_read_b64_tr_b16 on the row-major V tile, B = P from registers)
// With v_mfma_f32_32x32x16_bf16 a lane owns one query column (lane&31) and 16 keys of each 32-key tile,
// so the running max / sum are per-lane scalars plus one exchange with lane^32.  The register layout of
// P (keys (e&3)+8*(e>>2)+4*half per 16-key step) is used directly as the MFMA k-slot order; the V^T
// fragments are gathered in that same order, so P never moves between lanes.
// Work split: 4 wavefronts x 32 queries per block, K/V tiles of 64 keys double-buffered in LDS.
#include <type_traits>
#include "common.hpp"
#include "../../include/miphei_hip.h"

namespace {

typedef short v4s __attribute__((ext_vector_type(4)));
#ifndef MVIT_ATTN_PK
#define MVIT_ATTN_PK 1
#endif
#if MVIT_ATTN_PK
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 fma2(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
#else
// (measurement build: the same arithmetic on scalar VALU instructions -- compile the unit with -fno-slp-vectorize)
struct f32x2 {
  float x, y;
  __device__ __forceinline__ f32x2& operator+=(const f32x2& o) { x += o.x, y += o.y; return *this; }
};
__device__ __forceinline__ f32x2 operator*(const f32x2& a, const f32x2& b) { return {a.x * b.x, a.y * b.y}; }
__device__ __forceinline__ f32x2 fma2(f32x2 a, f32x2 b, f32x2 c) { return {__builtin_fmaf(a.x, b.x, c.x), __builtin_fmaf(a.y, b.y, c.y)}; }
#endif
#ifndef MVIT_ATTN_DKV_REV
#define MVIT_ATTN_DKV_REV 1   // 0: the dK/dV kernel walks its XCD's pairs in the dQ kernel's order (measurement)
#endif
constexpr float LOG2E = 1.4426950408889634f;
constexpr float LN2 = 0.6931471805599453f;
constexpr int KVB = 64;          // keys per LDS tile
constexpr int TILE_BYTES = KVB * 128;  // 64 rows x 64 bf16

// XOR swizzle of a tile row's eight 16-byte chunks.  The key is a bit permutation of (row >> 1) & 7 -- row bit 1 on chunk bit 2, row
// bits 2, 3 on chunk bits 0, 1 -- so that BOTH read patterns are conflict-free: the b128 fragment reads (32 consecutive rows, one
// chunk: any bijection of the three bits does) and the transposing b64 reads, whose 32-lane group covers 4 consecutive rows x 64
// bytes: rows r and r + 2 lie 256 bytes = one bank sweep apart and must take different 64-byte halves, i.e. row bit 1 has to
// reach chunk bit 2.  (Through round 4 the key was (row >> 1) & 7 itself: every transposing read 2-way conflicted, 25-32 % of the
// LDS cycles of the three kernels -- SQ_LDS_BANK_CONFLICT in profiles/r02_attn_counters.txt; model: tools/debug/attn_lds_banks.py.)
__device__ __forceinline__ int swz_key(int row) { return (((row >> 1) & 1) << 2) | ((row >> 2) & 3); }
__device__ __forceinline__ int swz(int row, int chunk) { return row * 128 + ((chunk ^ swz_key(row)) << 4); }

// 4 consecutive rows x this lane's column, from a row-major [rows][64] bf16 tile (hardware transpose read):
// a 16-lane group addresses a 4x16 block (lane i: row i>>2, cols 4*(i&3)..+3) and lane i receives column i.
__device__ __forceinline__ v4s tr_read4(const char* tile, int row_base, int col_base16, int lane) {
  const int i = lane & 15;
  const int row = row_base + (i >> 2);
  const int col = col_base16 + 4 * (i & 3);  // element column
  const int chunk = col >> 3;
  const char* p = tile + swz(row, chunk) + ((col & 7) << 1);
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s*)p);
}

__device__ __forceinline__ bf16x8 join(v4s a, v4s b) {
  union { struct { v4s lo, hi; } s; bf16x8 v; } u;
  u.s.lo = a;
  u.s.hi = b;
  return u.v;
}

struct AttnDims {
  int B, N, H, Dh;
  float scale;
  // block map without integer divisions (round 5): nx = row blocks per (batch, head) pair, and the multiply-high constants of the
  // divisions by nx and H, built by the host (make_dims).  A division by a run-time value costs a wave ~25 instructions and two
  // trips through the vector unit (v_rcp_iflag + readfirstlane); twelve waves per CU ran two of them at once on the CU's one scalar
  // unit in front of everything else (profiles/r04_attn_timing.txt: 17 % of a wave's cycles before its first tile step).
  unsigned nx, nx_magic, h_magic;
};
__device__ __forceinline__ unsigned fast_div(unsigned x, unsigned d, unsigned magic) { return mvit_fast_div(x, d, magic); }   // (common.hpp)

// DMA one 64-row x 64-col bf16 tile (rows row0.. of a matrix with row stride `rs` elements) straight into LDS
// (buffer_load ... lds, 16 B per lane): lane l of a wave fills row l>>3, 16-byte slot l&7 of 8 consecutive rows; the
// XOR swizzle is applied to the SOURCE chunk (slot s of row r holds chunk s ^ swz_key(r)); rows >= nvalid and
// columns >= Dh come back as zeros through an out-of-range offset.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const bf16_t* ptr) {
  const unsigned long long v = (unsigned long long)ptr;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
  return __builtin_amdgcn_make_buffer_rsrc((void*)(((unsigned long long)hi << 32) | lo), 0, 0x7fffffff, 0x00020000);
}
// This lane's two 16-byte pieces of a 64-row tile as byte offsets from the tile's first row: tile constants, the tile advance
// rides on the scalar offset of the DMA, so issuing a tile costs no address arithmetic (only the ragged last tile re-checks rows).
struct TileOff { unsigned full[2]; };
__device__ __forceinline__ TileOff tile_offsets(int Dh, unsigned rs, int tid) {
  const int wave_u = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  TileOff o;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int row = 8 * wave_u + 32 * j + (lane >> 3);
    const int c = (lane & 7) ^ swz_key(row);
    o.full[j] = c * 8 < Dh ? ((unsigned)row * rs + (unsigned)c * 8u) * 2u : 0x80000000u;
  }
  return o;
}
__device__ __forceinline__ void dma_tile(__amdgpu_buffer_rsrc_t rsrc, char* tile, const TileOff& o, int row0, int nvalid, unsigned rs,
                                         int tid) {
  typedef __attribute__((address_space(3))) void* lds_ptr;
  const int wave_u = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int soff = row0 * (int)rs * 2;
  unsigned off[2] = {o.full[0], o.full[1]};
  if (row0 + KVB > nvalid) {  // uniform: the ragged last tile
#pragma unroll
    for (int j = 0; j < 2; ++j)
      if (row0 + 8 * wave_u + 32 * j + (lane >> 3) >= nvalid) off[j] = 0x80000000u;
  }
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    asm volatile("" : "+v"(off[j]));
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr)(tile + (8 * wave_u + 32 * j) * 128), 16, off[j], soff, 0, 0);
  }
}

// Two-slot ring of [K|V] (or [Q|dO]) tile pairs: the DMA of tile t+1 is issued behind the barrier of step t and waited for at the
// top of step t+1.  The tile loop is unrolled by two so that the slot of a step is a compile-time constant (every LDS address of
// the step is then lane constant + immediate).  (Deeper rings: 48 KB per block costs the third resident block per CU.)
constexpr int NRING = 2;
constexpr int NRING_Q = 2;
constexpr float RESCALE_THR = 6.f;  // log2 units
__device__ __forceinline__ bf16x8 pack8(const float* v) {
  union { uint32_t u[4]; bf16x8 v; } r;
#pragma unroll
  for (int i = 0; i < 4; ++i) r.u[i] = pack2bf(v[2 * i], v[2 * i + 1]);
  return r.v;
}

// Epilogue store of one row per lane pair: lanes l and l + 32 hold the 4-column groups 8g + 4*half of the same row.  Packed groups
// (g, g + 1) are exchanged with v_permlane32_swap so that each lane owns 8 consecutive columns and issues one 16-byte store per
// pair (half the store instructions of the 8-byte form; cdna_hip_programming.md T21).  Dh % 16 != 0 keeps the 8-byte form.
__device__ __forceinline__ void store_row_groups(bf16_t* row, int Dh, int half, const uint2 (&w)[2][4], bool ok) {
  if (Dh & 15) {
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int d = 32 * dt + 8 * g + 4 * half;
        if (ok && d < Dh) *(uint2*)(row + d) = w[dt][g];
      }
    return;
  }
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int pr = 0; pr < 2; ++pr) {
      uint2 a = w[dt][2 * pr], b = w[dt][2 * pr + 1];
      const auto r0 = __builtin_amdgcn_permlane32_swap(a.x, b.x, false, false);
      const auto r1 = __builtin_amdgcn_permlane32_swap(a.y, b.y, false, false);
      const int d = 32 * dt + 16 * pr + 8 * half;   // lanes 0-31: columns 16pr .. +7, lanes 32-63: 16pr + 8 .. +15
      if (ok && d < Dh) *(uint4*)(row + d) = make_uint4(r0[0], r1[0], r0[1], r1[1]);
    }
}

// Block -> (row block, (batch, head)) coordinates.  The grid is 1-D; hardware places workgroup L on XCD L % 8, and the row
// blocks of one (batch, head) pair re-read the same K/V (or Q/dO) tiles, so each XCD is given a contiguous run of the
// pair-major order: the re-reads then hit that XCD's L2 instead of crossing the fabric once per row block
// (forward: 130 MB -> one pass over q, k, v, o per launch).  Bijective for any block count.
struct BlockXY { int x, y; };
// REV: the XCD's run is walked backwards.  The dK/dV kernel runs right after the dQ kernel on the same q / k / v / dO tiles: started
// from the end of the run, its first round of blocks meets the pairs the dQ kernel touched LAST, still in that XCD's L2, instead of
// opening with a cold burst (its prologue was 26 % of a wave's cycles, profiles/r05_attn_timing.txt).
template <bool REV = false>
__device__ __forceinline__ BlockXY block_xy(const AttnDims& dm) {
  const int total = gridDim.x, L = blockIdx.x;
  const int q = total >> 3, r = total & 7, xcd = L & 7;
  const int loc = REV ? (q + (xcd < r ? 1 : 0)) - 1 - (L >> 3) : (L >> 3);
  const int wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
  const int y = (int)fast_div((unsigned)wg, dm.nx, dm.nx_magic);
  return {wg - y * (int)dm.nx, y};
}

// ------------------------------------------------------------------ forward
// -DMVIT_ATTN_TIMING (measurement build, tools/debug/attn_timing.py): s_memtime stamps summed per wave -- [0] launch -> first tile step
// (descriptor set-up, Q fragments, first DMA), [1] waits at the top of the steps (own DMA + block barrier), [2] the steps' work,
// [3] epilogue, [4] steps -- written to the `lse` buffer's tail (the caller over-allocates it): 8 longs per wave
#ifdef MVIT_ATTN_TIMING
#define ATT_STAMP(k) { const long long now_ = (long long)__builtin_readcyclecounter(); tsum[k] += now_ - tlast; tlast = now_; }
#else
#define ATT_STAMP(k)
#endif
__global__ __launch_bounds__(256, 3) void attn_fwd_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ out,
                                                       bf16_t* __restrict__ out_res, float* __restrict__ lse, AttnDims dm) {
#ifdef MVIT_ATTN_TIMING
  long long tsum[5] = {0, 0, 0, 0, 0}, tlast = (long long)__builtin_readcyclecounter();
#endif
  extern __shared__ __attribute__((aligned(16))) char smem[];  // [NRING_Q][K|V]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int half = lane >> 5, l31 = lane & 31;
  const BlockXY bxy = block_xy(dm);
  const int bh = bxy.y, b = (int)fast_div((unsigned)bh, (unsigned)dm.H, dm.h_magic), h = bh - b * dm.H;
  const int N = dm.N, Dh = dm.Dh;
  const size_t rs = (size_t)3 * dm.H * Dh;  // row stride of packed qkv
  const bf16_t* qb = qkv + (size_t)b * N * rs + (size_t)h * Dh;
  const bf16_t* kb = qb + (size_t)dm.H * Dh;
  const bf16_t* vb = kb + (size_t)dm.H * Dh;
  const int q0 = bxy.x * 128 + wave * 32;
  const int q = q0 + l31;

  // Q fragment (B operand: k = d, col = q)
  bf16x8 qf[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const int d = 16 * s + 8 * half;
    uint4 t = make_uint4(0, 0, 0, 0);
    if (q < N && d < Dh) t = *(const uint4*)(qb + (size_t)q * rs + d);
    qf[s] = *(bf16x8*)&t;
  }

  f32x16 oacc[2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) oacc[i][r] = 0.f;
  float m_run = -1e30f, l_run = 0.f;
  const float sc = dm.scale * LOG2E;

  const int ntiles = (N + KVB - 1) / KVB;
  const __amdgpu_buffer_rsrc_t rK = make_rsrc(kb), rV = make_rsrc(vb);
  const TileOff to = tile_offsets(Dh, (unsigned)rs, tid);
  auto issue = [&](int t, char* dst) {
    dma_tile(rK, dst, to, t * KVB, N, (unsigned)rs, tid);
    dma_tile(rV, dst + TILE_BYTES, to, t * KVB, N, (unsigned)rs, tid);
  };
  issue(0, smem);
  // `live`: this wave owns at least one real query row (padding-only waves still take part in the barriers and the DMA)
  auto step = [&](int t, auto slot_tag, auto ragged_tag, auto live_tag) __attribute__((always_inline)) {
    constexpr int SLOT = decltype(slot_tag)::value;
    constexpr bool RAGGED = decltype(ragged_tag)::value;
    ATT_STAMP(t == 0 ? 0 : 2)
    // lgkmcnt(0) too: s_barrier does not wait for LDS reads in flight, and hipcc sinks the MFMAs that consume the previous tile's last
    // fragment reads below this barrier (they are not memory operations) -- the reads then cross it unfinished while the other waves
    // issue the DMA that refills their slot (round 4: one 32-query slab in ~1e4 launches came back with a few stale K / V^T rows)
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();  // tile t visible to all waves; every wave is done with tile t-1 (the other slot)
    ATT_STAMP(1)
#ifdef MVIT_ATTN_TIMING
    tsum[4] += 1;
#endif
    if (t + 1 < ntiles) issue(t + 1, smem + (SLOT ^ 1) * 2 * TILE_BYTES);
    const char* Ks = smem + SLOT * 2 * TILE_BYTES;
    const char* Vs = Ks + TILE_BYTES;
    const int kv0 = t * KVB;

    if constexpr (!decltype(live_tag)::value) return;
    {
      const bool kt1_live = !RAGGED || kv0 + 32 < N;  // second 32-key half of the tile holds at least one real key
    f32x16 st[2];
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
#pragma unroll
      for (int r = 0; r < 16; ++r) st[kt][r] = kt == 1 && !kt1_live ? -1e30f : 0.f;
      if (kt == 1 && !kt1_live) continue;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const bf16x8 a = *(const bf16x8*)(Ks + swz(32 * kt + l31, 2 * s + half));
        st[kt] = mvit_mfma32(a, qf[s], st[kt], 0, 0, 0);
      }
    }
    // online softmax (log2 domain): p = 2^(s*sc - m); the key mask only exists in the ragged-tile instantiation
    if constexpr (RAGGED) {
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = kv0 + 32 * kt + (r & 3) + 8 * (r >> 2) + 4 * half;
          if (key >= N) st[kt][r] = -1e30f;
        }
    }
    float mloc = st[0][0];
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int r = 0; r < 16; ++r) mloc = fmaxf(mloc, st[kt][r]);
    mloc = fmaxf(mloc * sc, -1e30f);
    mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64));
    // lazy rescale: O and l are only rescaled when the running max grows by more than 2^RESCALE_THR (rare after the
    // first tiles); until then P is formed against the stale max and stays <= 2^RESCALE_THR
    const float m_new = fmaxf(m_run, mloc);
    if (!__all(m_new - m_run <= RESCALE_THR)) {
      const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
      l_run *= alpha;
      m_run = m_new;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[i][r] *= alpha;
    }
    // (packed f32 FMA / add: two scores per VALU instruction; the exponentials stay scalar)
    const f32x2 sc2 = {sc, sc}, nm2 = {-m_run, -m_run};
    f32x2 ls2 = {0.f, 0.f};
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        const f32x2 x = fma2((f32x2){st[kt][r], st[kt][r + 1]}, sc2, nm2);
        const f32x2 p2 = {__builtin_amdgcn_exp2f(x.x), __builtin_amdgcn_exp2f(x.y)};
        st[kt][r] = p2.x;
        st[kt][r + 1] = p2.y;
        ls2 += p2;
      }
    l_run += ls2.x + ls2.y;
    // O^T += V^T P^T
    // (Round 4: this loop's V^T fragment reads are the LAST LDS reads of a step.  They used to cross the next step's barrier
    //  unfinished -- see the wait in front of it -- and about one launch in 40 returned a 32-query slab with wrong columns 32..63.
    //  Rearranging this loop hid that for a while (the reads happened to complete earlier); the fix is the lgkmcnt(0) at the barrier.)
#pragma unroll
    for (int s2 = 0; s2 < 4; ++s2) {
      const int kt = s2 >> 1, h2 = s2 & 1;
      if (kt == 1 && !kt1_live) continue;  // P is exactly 0 there
      float pv[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) pv[e] = st[kt][8 * h2 + e];
      const bf16x8 pb = pack8(pv);
      const int kbase = 32 * kt + 16 * h2 + 4 * half;
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) {
        const int cb = 32 * dt + 16 * ((lane >> 4) & 1);
        const bf16x8 a = join(tr_read4(Vs, kbase, cb, lane), tr_read4(Vs, kbase + 8, cb, lane));
        oacc[dt] = mvit_mfma32(a, pb, oacc[dt], 0, 0, 0);
      }
    }
    }
  };
  // full tiles two at a time (compile-time ring slot), then the odd full tile, then the ragged last tile: one straight-line body
  // per loop, so the accumulators never move between registers
  auto run = [&](auto live_tag) __attribute__((always_inline)) {
    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, 1>;
    const int nfull = N / KVB;
    int t = 0;
    for (; t + 2 <= nfull; t += 2) {
      step(t, S0{}, std::false_type{}, live_tag);
      step(t + 1, S1{}, std::false_type{}, live_tag);
    }
    if (t < nfull) {
      step(t, S0{}, std::false_type{}, live_tag);
      if (t + 1 < ntiles) step(t + 1, S1{}, std::true_type{}, live_tag);
    } else if (t < ntiles) {
      step(t, S0{}, std::true_type{}, live_tag);
    }
  };
  if (q0 >= N)  // wave-uniform
    run(std::false_type{});
  else
    run(std::true_type{});
  ATT_STAMP(2)
  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  const float inv = 1.f / l_tot;
  {
    // rounding residual of O (bf16 again): the backward pass forms D = sum_d dO * (O + residual), i.e. from O at ~16 mantissa
    // bits.  With D from the bf16 O alone its error (2^-9 |O||dO|) does not cancel against dP in dS = P (dP - D) the way it does
    // in the unfused softmax backward, and where attention is near-uniform (dQ is a small residual of large terms) dQ came out
    // 20-30 % wrong (tools/debug/attn_insitu.py).
    uint2 wo[2][4], wr[2][4];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float v0 = oacc[dt][4 * g] * inv, v1 = oacc[dt][4 * g + 1] * inv, v2 = oacc[dt][4 * g + 2] * inv, v3 = oacc[dt][4 * g + 3] * inv;
        wo[dt][g].x = pack2bf(v0, v1);
        wo[dt][g].y = pack2bf(v2, v3);
        wr[dt][g].x = pack2bf(v0 - lo16f(wo[dt][g].x), v1 - hi16f(wo[dt][g].x));
        wr[dt][g].y = pack2bf(v2 - lo16f(wo[dt][g].y), v3 - hi16f(wo[dt][g].y));
      }
    const size_t ro = ((size_t)b * N + (q < N ? q : 0)) * ((size_t)dm.H * Dh) + (size_t)h * Dh;
    store_row_groups(out + ro, Dh, half, wo, q < N);
    if (out_res) store_row_groups(out_res + ro, Dh, half, wr, q < N);
    if (q < N && lse && half == 0) lse[(size_t)bh * N + q] = (m_run + log2f(l_tot)) * LN2;
  }
#ifdef MVIT_ATTN_TIMING
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  ATT_STAMP(3)
  if (lane == 0) {
    long long* prof = (long long*)(lse + (size_t)dm.B * dm.H * N) + ((size_t)blockIdx.x * 4 + wave) * 8;
    for (int k = 0; k < 5; ++k) prof[k] = tsum[k];
    prof[5] = (long long)__builtin_readcyclecounter();
    prof[6] = __builtin_amdgcn_s_getreg((8 << 11) | (0 << 6) | 20) /* XCC_ID */;
  }
#endif
}

// ------------------------------------------------------------------ backward, dQ (query-stationary, S^T form)
//   S^T = K Q^T, P = exp(S*scale - L), dP^T = V dO^T, dS^T = P*(dP^T - D)*scale, dQ^T += K^T dS^T
// D[b,h,q] = sum_d dO*O is formed here from the rows this lane already holds (O is one more 16-byte load per chunk)
// and stored for the dK/dV kernel, which runs after this one: no separate preparation pass.
__global__ __launch_bounds__(256, 3) void attn_bwd_dq_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ o,
                                                          const bf16_t* __restrict__ ores,
                                                          const bf16_t* __restrict__ dO, const float* __restrict__ lse,
                                                          float* __restrict__ Dv, bf16_t* __restrict__ dqkv, AttnDims dm) {
#ifdef MVIT_ATTN_TIMING
  long long tsum[5] = {0, 0, 0, 0, 0}, tlast = (long long)__builtin_readcyclecounter();
  const long long tstart = tlast;
#endif
  extern __shared__ __attribute__((aligned(16))) char smem[];  // [NRING_Q][K|V]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int half = lane >> 5, l31 = lane & 31;
  const BlockXY bxy = block_xy(dm);
  const int bh = bxy.y, b = (int)fast_div((unsigned)bh, (unsigned)dm.H, dm.h_magic), h = bh - b * dm.H;
  const int N = dm.N, Dh = dm.Dh;
  const size_t rs = (size_t)3 * dm.H * Dh, ors = (size_t)dm.H * Dh;
  const bf16_t* qb = qkv + (size_t)b * N * rs + (size_t)h * Dh;
  const bf16_t* kb = qb + (size_t)dm.H * Dh;
  const bf16_t* vb = kb + (size_t)dm.H * Dh;
  const bf16_t* dob = dO + (size_t)b * N * ors + (size_t)h * Dh;
  const int q = bxy.x * 128 + wave * 32 + l31;

  // first K / V tile on its way before the fragment loads below (their use in the D sum would otherwise put one whole memory
  // round trip in front of the DMA's)
  const int ntiles = (N + KVB - 1) / KVB;
  const __amdgpu_buffer_rsrc_t rK = make_rsrc(kb), rV = make_rsrc(vb);
  const TileOff to = tile_offsets(Dh, (unsigned)rs, tid);
  auto issue = [&](int t, char* dst) {
    dma_tile(rK, dst, to, t * KVB, N, (unsigned)rs, tid);
    dma_tile(rV, dst + TILE_BYTES, to, t * KVB, N, (unsigned)rs, tid);
  };
  issue(0, smem);

  bf16x8 qf[4], dof[4];
  float dsum = 0.f;
  const bf16_t* rsrc = ores ? ores : o;   // residual of O (absent: O is read a second time and the value dropped)
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const int d = 16 * s + 8 * half;
    uint4 t = make_uint4(0, 0, 0, 0), u = make_uint4(0, 0, 0, 0), ov = make_uint4(0, 0, 0, 0), rv = make_uint4(0, 0, 0, 0);
    if (q < N && d < Dh) {
      t = *(const uint4*)(qb + (size_t)q * rs + d);
      u = *(const uint4*)(dob + (size_t)q * ors + d);
      ov = *(const uint4*)(o + (size_t)b * N * ors + (size_t)h * Dh + (size_t)q * ors + d);
      rv = *(const uint4*)(rsrc + (size_t)b * N * ors + (size_t)h * Dh + (size_t)q * ors + d);   // (unconditional: no branch per load)
    }
    if (!ores) rv = make_uint4(0, 0, 0, 0);
    qf[s] = *(bf16x8*)&t;
    dof[s] = *(bf16x8*)&u;
    const uint32_t ua[4] = {ov.x, ov.y, ov.z, ov.w}, uc[4] = {u.x, u.y, u.z, u.w}, ur[4] = {rv.x, rv.y, rv.z, rv.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      dsum += (lo16f(ua[j]) + lo16f(ur[j])) * lo16f(uc[j]);
      dsum += (hi16f(ua[j]) + hi16f(ur[j])) * hi16f(uc[j]);
    }
  }
  dsum += __shfl_xor(dsum, 32, 64);  // the other half of the head dimension
  const float Lq = q < N ? lse[(size_t)bh * N + q] * LOG2E : 0.f;
  const float Dq = dsum;
  if (half == 0 && q < N) Dv[(size_t)bh * N + q] = dsum;
  const float sc = dm.scale * LOG2E;

  f32x16 dqacc[2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) dqacc[i][r] = 0.f;

  auto step = [&](int t, auto slot_tag, auto ragged_tag, auto live_tag) __attribute__((always_inline)) {
    constexpr int SLOT = decltype(slot_tag)::value;
    constexpr bool RAGGED = decltype(ragged_tag)::value;
    ATT_STAMP(t == 0 ? 0 : 2)
    // lgkmcnt(0) too: s_barrier does not wait for LDS reads in flight, and hipcc sinks the MFMAs that consume the previous tile's last
    // fragment reads below this barrier (they are not memory operations) -- the reads then cross it unfinished while the other waves
    // issue the DMA that refills their slot (round 4: one 32-query slab in ~1e4 launches came back with a few stale K / V^T rows)
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();  // tile t visible to all waves; every wave is done with tile t-1 (the other slot)
    ATT_STAMP(1)
#ifdef MVIT_ATTN_TIMING
    tsum[4] += 1;
#endif
    if (t + 1 < ntiles) issue(t + 1, smem + (SLOT ^ 1) * 2 * TILE_BYTES);
    const char* Ks = smem + SLOT * 2 * TILE_BYTES;
    const char* Vs = Ks + TILE_BYTES;
    const int kv0 = t * KVB;
    if constexpr (!decltype(live_tag)::value) return;
    {
      const bool kt1_live = !RAGGED || kv0 + 32 < N;
      // one 32-key sub-tile at a time: S^T, dP^T -> dS^T -> dQ^T
#pragma unroll
      for (int kt = 0; kt < 2; ++kt) {
        if (kt == 1 && !kt1_live) continue;  // dS is exactly 0 there
        f32x16 st, dp;
#pragma unroll
        for (int r = 0; r < 16; ++r) st[r] = 0.f, dp[r] = 0.f;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          const bf16x8 a = *(const bf16x8*)(Ks + swz(32 * kt + l31, 2 * s + half));
          st = mvit_mfma32(a, qf[s], st, 0, 0, 0);
          const bf16x8 v = *(const bf16x8*)(Vs + swz(32 * kt + l31, 2 * s + half));
          dp = mvit_mfma32(v, dof[s], dp, 0, 0, 0);
        }
        // dS^T = P (dP - D) scale, two scores per packed VALU instruction
        const f32x2 sc2 = {sc, sc}, nL2 = {-Lq, -Lq}, s2 = {dm.scale, dm.scale}, nD2 = {-Dq * dm.scale, -Dq * dm.scale};
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
          const int key = kv0 + 32 * kt + (r & 3) + 8 * (r >> 2) + 4 * half;
          const f32x2 x = fma2((f32x2){st[r], st[r + 1]}, sc2, nL2);
          f32x2 p2 = {__builtin_amdgcn_exp2f(x.x), __builtin_amdgcn_exp2f(x.y)};
          if (RAGGED) p2 = {key < N ? p2.x : 0.f, key + 1 < N ? p2.y : 0.f};
          const f32x2 ds2 = p2 * fma2((f32x2){dp[r], dp[r + 1]}, s2, nD2);
          st[r] = ds2.x;
          st[r + 1] = ds2.y;
        }
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2) {
          float pv[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) pv[e] = st[8 * h2 + e];
          const bf16x8 dsb = pack8(pv);
          const int kbase = 32 * kt + 16 * h2 + 4 * half;
#pragma unroll
          for (int dt = 0; dt < 2; ++dt) {
            const int cb = 32 * dt + 16 * ((lane >> 4) & 1);
            const bf16x8 a = join(tr_read4(Ks, kbase, cb, lane), tr_read4(Ks, kbase + 8, cb, lane));
            dqacc[dt] = mvit_mfma32(a, dsb, dqacc[dt], 0, 0, 0);
          }
        }
      }
    }
  };
  auto run = [&](auto live_tag) __attribute__((always_inline)) {   // (see the forward kernel)
    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, 1>;
    const int nfull = N / KVB;
    int t = 0;
    for (; t + 2 <= nfull; t += 2) {
      step(t, S0{}, std::false_type{}, live_tag);
      step(t + 1, S1{}, std::false_type{}, live_tag);
    }
    if (t < nfull) {
      step(t, S0{}, std::false_type{}, live_tag);
      if (t + 1 < ntiles) step(t + 1, S1{}, std::true_type{}, live_tag);
    } else if (t < ntiles) {
      step(t, S0{}, std::true_type{}, live_tag);
    }
  };
  if (bxy.x * 128 + wave * 32 >= N)  // wave-uniform: padding rows only
    run(std::false_type{});
  else
    run(std::true_type{});
  ATT_STAMP(2)
  {
    uint2 wq[2][4];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        wq[dt][g].x = pack2bf(dqacc[dt][4 * g], dqacc[dt][4 * g + 1]);
        wq[dt][g].y = pack2bf(dqacc[dt][4 * g + 2], dqacc[dt][4 * g + 3]);
      }
    store_row_groups(dqkv + ((size_t)b * N + (q < N ? q : 0)) * rs + (size_t)h * Dh, Dh, half, wq, q < N);
  }
#ifdef MVIT_ATTN_TIMING
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  ATT_STAMP(3)
  if (lane == 0) {
    long long* prof = (long long*)(Dv + (size_t)dm.B * dm.H * N) + ((size_t)blockIdx.x * 4 + wave) * 8;
    for (int k = 0; k < 5; ++k) prof[k] = tsum[k];
    prof[5] = (long long)__builtin_readcyclecounter();
    prof[6] = tstart;
    prof[7] = __builtin_amdgcn_s_getreg((8 << 11) | (0 << 6) | 20) /* XCC_ID */;
  }
#endif
}

// ------------------------------------------------------------------ backward, dK / dV (key-stationary, S form)
//   S = Q K^T, P = exp(S*scale - L_q), dP = dO V^T, dS = P*(dP - D_q)*scale, dV^T += dO^T P, dK^T += Q^T dS
__global__ __launch_bounds__(256, 2) void attn_bwd_dkv_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ dO,
                                                           const float* __restrict__ lse, const float* __restrict__ Dv_c,
                                                           bf16_t* __restrict__ dqkv, AttnDims dm) {
  float* Dv = const_cast<float*>(Dv_c);   // (the timing build writes its stamps behind the D values)
#ifdef MVIT_ATTN_TIMING
  long long tsum[5] = {0, 0, 0, 0, 0}, tlast = (long long)__builtin_readcyclecounter();
  const long long tstart = tlast;
#endif
  extern __shared__ __attribute__((aligned(16))) char smem[];  // [NRING][Q|dO] + L[Npad] + D[Npad] (f32)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int half = lane >> 5, l31 = lane & 31;
  const BlockXY bxy = block_xy<MVIT_ATTN_DKV_REV>(dm);
  const int bh = bxy.y, b = (int)fast_div((unsigned)bh, (unsigned)dm.H, dm.h_magic), h = bh - b * dm.H;
  const int N = dm.N, Dh = dm.Dh;
  const size_t rs = (size_t)3 * dm.H * Dh, ors = (size_t)dm.H * Dh;
  const bf16_t* qb = qkv + (size_t)b * N * rs + (size_t)h * Dh;
  const bf16_t* kb = qb + (size_t)dm.H * Dh;
  const bf16_t* vb = kb + (size_t)dm.H * Dh;
  const bf16_t* dob = dO + (size_t)b * N * ors + (size_t)h * Dh;
  const int key = bxy.x * 128 + wave * 32 + l31;
  const int Npad = ((N + KVB - 1) / KVB) * KVB;
  float* LD = (float*)(smem + NRING * 2 * TILE_BYTES);  // L[Npad] (log2 units) then D[Npad]

  bf16x8 kf[4], vf[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const int d = 16 * s + 8 * half;
    uint4 t = make_uint4(0, 0, 0, 0), u = make_uint4(0, 0, 0, 0);
    if (key < N && d < Dh) {
      t = *(const uint4*)(kb + (size_t)key * rs + d);
      u = *(const uint4*)(vb + (size_t)key * rs + d);
    }
    kf[s] = *(bf16x8*)&t;
    vf[s] = *(bf16x8*)&u;
  }
  const float sc = dm.scale * LOG2E;
  f32x16 dkacc[2], dvacc[2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) dkacc[i][r] = 0.f, dvacc[i][r] = 0.f;

  const __amdgpu_buffer_rsrc_t rQ = make_rsrc(qb), rD = make_rsrc(dob);
  const int ntiles = (N + KVB - 1) / KVB;
  const TileOff toq = tile_offsets(Dh, (unsigned)rs, tid), tod = tile_offsets(Dh, (unsigned)ors, tid);
  auto issue = [&](int t, char* dst) {
    dma_tile(rQ, dst, toq, t * KVB, N, (unsigned)rs, tid);
    dma_tile(rD, dst + TILE_BYTES, tod, t * KVB, N, (unsigned)ors, tid);
  };
  issue(0, smem);
  for (int i = tid; i < Npad; i += 256) {  // per-row log-sum-exp and dO.O of the whole head (published by the first barrier)
    LD[i] = i < N ? lse[(size_t)bh * N + i] * LOG2E : 0.f;
    LD[Npad + i] = i < N ? Dv[(size_t)bh * N + i] * dm.scale : 0.f;   // D pre-multiplied by the softmax scale
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  auto step = [&](int t, auto slot_tag, auto ragged_tag, auto live_tag) __attribute__((always_inline)) {
    constexpr int SLOT = decltype(slot_tag)::value;
    constexpr bool RAGGED = decltype(ragged_tag)::value;
    ATT_STAMP(t == 0 ? 0 : 2)
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // (lgkmcnt: see the forward kernel's step)
    __builtin_amdgcn_s_barrier();
    ATT_STAMP(1)
#ifdef MVIT_ATTN_TIMING
    tsum[4] += 1;
#endif
    if (t + 1 < ntiles) issue(t + 1, smem + (SLOT ^ 1) * 2 * TILE_BYTES);
    const char* Qs = smem + SLOT * 2 * TILE_BYTES;
    const char* Ds = Qs + TILE_BYTES;
    const int qt0 = t * KVB;
    const float* Ls = LD + qt0;
    if constexpr (!decltype(live_tag)::value) return;
    {
      const bool qt1_live = !RAGGED || qt0 + 32 < N;
      // one 32-row query sub-tile at a time: S, dP -> P, dS -> dV^T, dK^T (keeps the live accumulator set small)
#pragma unroll
      for (int qt = 0; qt < 2; ++qt) {
        if (qt == 1 && !qt1_live) continue;  // P = dS = 0 for padding query rows
        f32x16 st, dp;
#pragma unroll
        for (int r = 0; r < 16; ++r) st[r] = 0.f, dp[r] = 0.f;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          const bf16x8 a = *(const bf16x8*)(Qs + swz(32 * qt + l31, 2 * s + half));
          st = mvit_mfma32(a, kf[s], st, 0, 0, 0);
          const bf16x8 g = *(const bf16x8*)(Ds + swz(32 * qt + l31, 2 * s + half));
          dp = mvit_mfma32(g, vf[s], dp, 0, 0, 0);
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int qb4 = 32 * qt + 8 * g + 4 * half;  // 4 consecutive query rows per register group
          const float4 L4 = *(const float4*)(Ls + qb4), D4 = *(const float4*)(Ls + Npad + qb4);
          const float Lv[4] = {L4.x, L4.y, L4.z, L4.w}, Dv_[4] = {D4.x, D4.y, D4.z, D4.w};
#pragma unroll
          for (int e = 0; e < 4; e += 2) {   // two query rows per packed VALU instruction
            const int r = 4 * g + e;
            const f32x2 x = fma2((f32x2){st[r], st[r + 1]}, (f32x2){sc, sc}, (f32x2){-Lv[e], -Lv[e + 1]});
            f32x2 p2 = {__builtin_amdgcn_exp2f(x.x), __builtin_amdgcn_exp2f(x.y)};
            if (RAGGED) p2 = {qt0 + qb4 + e < N ? p2.x : 0.f, qt0 + qb4 + e + 1 < N ? p2.y : 0.f};
            const f32x2 ds2 = p2 * fma2((f32x2){dp[r], dp[r + 1]}, (f32x2){dm.scale, dm.scale},
                                                              (f32x2){-Dv_[e], -Dv_[e + 1]});   // (D is stored scaled)
            st[r] = p2.x, st[r + 1] = p2.y;        // P
            dp[r] = ds2.x, dp[r + 1] = ds2.y;      // dS
          }
        }
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2) {
          float pv[8], dv[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) pv[e] = st[8 * h2 + e], dv[e] = dp[8 * h2 + e];
          const bf16x8 pb = pack8(pv), dsb = pack8(dv);
          const int qbase = 32 * qt + 16 * h2 + 4 * half;
#pragma unroll
          for (int dt = 0; dt < 2; ++dt) {
            const int cb = 32 * dt + 16 * ((lane >> 4) & 1);
            const bf16x8 a = join(tr_read4(Ds, qbase, cb, lane), tr_read4(Ds, qbase + 8, cb, lane));
            dvacc[dt] = mvit_mfma32(a, pb, dvacc[dt], 0, 0, 0);
            const bf16x8 a2 = join(tr_read4(Qs, qbase, cb, lane), tr_read4(Qs, qbase + 8, cb, lane));
            dkacc[dt] = mvit_mfma32(a2, dsb, dkacc[dt], 0, 0, 0);
          }
        }
      }
    }
  };
  auto run = [&](auto live_tag) __attribute__((always_inline)) {   // (see the forward kernel)
    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, 1>;
    const int nfull = N / KVB;
    int t = 0;
    for (; t + 2 <= nfull; t += 2) {
      step(t, S0{}, std::false_type{}, live_tag);
      step(t + 1, S1{}, std::false_type{}, live_tag);
    }
    if (t < nfull) {
      step(t, S0{}, std::false_type{}, live_tag);
      if (t + 1 < ntiles) step(t + 1, S1{}, std::true_type{}, live_tag);
    } else if (t < ntiles) {
      step(t, S0{}, std::true_type{}, live_tag);
    }
  };
  if (bxy.x * 128 + wave * 32 >= N)  // wave-uniform: padding rows only
    run(std::false_type{});
  else
    run(std::true_type{});
  ATT_STAMP(2)
  {
    uint2 wk[2][4], wv[2][4];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        wk[dt][g].x = pack2bf(dkacc[dt][4 * g], dkacc[dt][4 * g + 1]);
        wk[dt][g].y = pack2bf(dkacc[dt][4 * g + 2], dkacc[dt][4 * g + 3]);
        wv[dt][g].x = pack2bf(dvacc[dt][4 * g], dvacc[dt][4 * g + 1]);
        wv[dt][g].y = pack2bf(dvacc[dt][4 * g + 2], dvacc[dt][4 * g + 3]);
      }
    bf16_t* krow = dqkv + ((size_t)b * N + (key < N ? key : 0)) * rs + (size_t)(dm.H + h) * Dh;
    store_row_groups(krow, Dh, half, wk, key < N);
    store_row_groups(krow + (size_t)dm.H * Dh, Dh, half, wv, key < N);
  }
#ifdef MVIT_ATTN_TIMING
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  ATT_STAMP(3)
  if (lane == 0) {
    long long* prof = (long long*)(Dv + (size_t)dm.B * dm.H * N) + ((size_t)(gridDim.x + blockIdx.x) * 4 + wave) * 8;
    for (int k = 0; k < 5; ++k) prof[k] = tsum[k];
    prof[5] = (long long)__builtin_readcyclecounter();
    prof[6] = tstart;
    prof[7] = __builtin_amdgcn_s_getreg((8 << 11) | (0 << 6) | 20) /* XCC_ID */;
  }
#endif
}


// ------------------------------------------------------------------ backward in ONE pass per (batch, head) pair (round 6)
// N <= 7 * 48 = 336 keys and Dh = 64: a whole pair fits one workgroup, so S and dP are formed ONCE (5 matmuls and one exp pass instead of
// the 7 and 2 of the two kernels above), q / k / v / dO are read once, and there is one prologue instead of two.
//   key wave w (one per 48 keys; K and V fragments, dK^T and dV^T accumulators in registers for the whole kernel), per 32-query step:
//     X  S = Q K^T, dP = dO V^T          (A = Q / dO rows of the step's LDS tile, B = the resident fragments; v_mfma_f32_16x16x32_bf16)
//     Y  P = exp2(S sc - L), dS = P (dP scale - D scale); dS -> LDS as bf16 [key][32 q]     (lane = key column, 4 query rows per tile)
//     Z  dV^T += dO^T P, dK^T += Q^T dS  (A = transposing reads of the same tiles; P / dS stay in their lanes as B operands)
//   dQ needs dS with the key axis as the MFMA k dimension, i.e. transposed across lanes AND summed over all key waves.  Instead of a
//   cross-wave f32 reduction (1 MB of LDS partials per pair) the bf16 dS^T tile of a step (22 KB) is read back one step later by
//   transposing reads and multiplied with K^T over ALL keys: a fixed summation order -- bit-identical from run to run, no atomics, no
//   partials in HBM.  That product (W) belongs to the HELPER wave, which has the registers the key waves lack: it keeps the K^T operand
//   fragments of all four 16-row d tiles (4 x 11 x 4 = 176 registers, read once from the K rows in LDS), so a step's eight dQ tiles cost
//   44 transposing reads instead of the 352 of one tile per wave (the first version: LDS-bound, 1570 cycles per tile).  The helper also
//   issues every DMA ([Q | dO | O | O residual] step tiles two steps ahead) and forms D = sum_d dO (O + residual) from the landed tile.
//   One s_barrier per step.  (Tried and dropped, docs/rounds/round6.md: three barrier-separated slots per step with the two waves of a SIMD
//   one slot apart -- slower, the slots' maxima add up; the second half of the waves one phase behind (Z(j - 1) X Y) -- no difference.)
// LDS: K rows 44 KB (after step 0: staging of the dK / dV rows) + two dS^T buffers 44 KB + four step-tile slots 64 KB + L, D 2.8 KB =
// 155 KB: one workgroup per CU (512 threads, 2 waves per SIMD, <= 256 registers).  Swizzles: tools/debug/attn_fused_lds_banks.py.
namespace fused {
constexpr int KW = 48;                 // keys per key wave
constexpr int MAXKW = 7;               // key waves (+ 1 helper = 512 threads)
constexpr int QB = 32;                 // query rows per step
constexpr int NSLOT = 4;               // ring of step tiles
constexpr int TILE_B = QB * 128;       // one 32-row tile of Q, dO, O or O's residual
constexpr int SLOT_B = 4 * TILE_B;
constexpr int MAXK32 = 11;             // 32-key steps of the dQ product at 7 key waves

typedef __attribute__((address_space(3))) void* lds_ptr;
__device__ __forceinline__ v4s trd(const char* p) { return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s*)p); }
__device__ __forceinline__ f32x4 mfma16(bf16x8 a, bf16x8 b, f32x4 c) { return mvit_mfma16(a, b, c, 0, 0, 0); }

// -DMVIT_ATTN_TIMING (measurement build, tools/debug/attn_fused_timing.py): s_memtime stamps summed per wave and phase, written behind
// the D values (the tool over-allocates `dsum`): 16 longs per wave
#ifdef MVIT_ATTN_TIMING
#define FST_DECL long long tsum[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = (long long)__builtin_readcyclecounter(); const long long tstart = tlast;
#define FST(k) { const long long now_ = (long long)__builtin_readcyclecounter(); tsum[k] += now_ - tlast; tlast = now_; }
#define FST_WRITE(P, dm, wave, lane)                                                                                              \
  if (lane == 0) {                                                                                                               \
    long long* prof = (long long*)((P).Dv + (size_t)(dm).B * (dm).H * (dm).N) + ((size_t)blockIdx.x * 8 + wave) * 16;             \
    for (int k = 0; k < 8; ++k) prof[k] = tsum[k];                                                                               \
    prof[8] = tstart, prof[9] = (long long)__builtin_readcyclecounter();                                                         \
    prof[10] = __builtin_amdgcn_s_getreg((8 << 11) | (0 << 6) | 20) /* XCC_ID */;                                                \
  }
#else
#define FST_DECL
#define FST(k)
#define FST_WRITE(P, dm, wave, lane)
#endif

struct Ptrs {
  const bf16_t *qb, *kb, *vb, *dob, *ob, *orb;
  bf16_t* dq;       // dqkv row 0 of this batch element, q section of this head (k: + H*64, v: + 2*H*64)
  const float* lse;
  float* Dv;
};

// Step-tile DMA, shared by all waves: a tile is 16 pieces of 1 KB (4 matrices x 4 row groups of 8) and issuing one costs a wave 100-200
// cycles, so wave w issues pieces w and w + 8 (row group w & 3 of matrices w >> 2 and (w >> 2) + 2: waves 0-3 Q and O, waves 4-7 dO and
// O's residual) at the top of a step and waits for them (vmcnt) in front of the step's barrier.  Lane l of a piece fills (row 8 m + (l >> 3),
// physical chunk l & 7) from source chunk (l & 7) ^ (((row >> 1) & 3) << 1); rows >= N come back as zeros through an out-of-range offset.
struct TilePieces {
  __amdgpu_buffer_rsrc_t r0, r1;
  unsigned v0, v1;     // lane offsets inside a tile (bytes)
  unsigned s0, s1;     // row strides (bytes)
  int m, dst0, dst1;   // row group, LDS offsets inside a slot
  bool on1;
};
__device__ __forceinline__ TilePieces tile_pieces(int w, int lane, const Ptrs& P, size_t rs, size_t ors) {
  TilePieces t;
  const int hi = (w >> 2) & 1;
  t.m = w & 3;
  const int row = 8 * t.m + (lane >> 3), c = (lane & 7) ^ (((row >> 1) & 3) << 1);
  t.s0 = (unsigned)(hi ? ors : rs) * 2u, t.s1 = (unsigned)ors * 2u;
  t.v0 = (unsigned)row * t.s0 + (unsigned)c * 16u;
  t.v1 = (unsigned)row * t.s1 + (unsigned)c * 16u;
  t.r0 = make_rsrc(hi ? P.dob : P.qb);
  t.r1 = make_rsrc(hi ? (P.orb ? P.orb : P.ob) : P.ob);
  t.on1 = !hi || P.orb != nullptr;
  t.dst0 = hi * TILE_B + t.m * 1024;
  t.dst1 = (2 + hi) * TILE_B + t.m * 1024;
  return t;
}
template <bool FAST>
__device__ __forceinline__ void issue_one(const TilePieces& t, char* RING, int j, int N, int lane) {
  char* dst = RING + (j & (NSLOT - 1)) * SLOT_B;
  const int q0 = QB * j;
  const bool ok = q0 + 8 * t.m + (lane >> 3) < N;
  unsigned o0 = ok ? t.v0 : 0x80000000u, o1 = ok ? t.v1 : 0x80000000u;
  asm volatile("" : "+v"(o0), "+v"(o1));
  __builtin_amdgcn_raw_ptr_buffer_load_lds(t.r0, (lds_ptr)(dst + t.dst0), 16, o0, q0 * (int)t.s0, 0, 0);
  if (FAST || t.on1) __builtin_amdgcn_raw_ptr_buffer_load_lds(t.r1, (lds_ptr)(dst + t.dst1), 16, o1, q0 * (int)t.s1, 0, 0);
}

// this wave's pieces of tile j: pair `wave` (precomputed), and with fewer than 8 waves the pairs wave + NW, wave + 2 NW, ... as well
struct TileIssue {
  TilePieces tp;
  const Ptrs* P;
  size_t rs, ors;
  int wave, NW;
};
// FAST = 8 waves and a residual of O (the training shape): one pair per wave, no run-time conditions in the steady state
template <bool FAST>
__device__ __forceinline__ void issue_pieces(const TileIssue& ti, char* RING, int j, int N, int lane) {
  issue_one<FAST>(ti.tp, RING, j, N, lane);
  if constexpr (!FAST)
    for (int pw = ti.wave + ti.NW; pw < 8; pw += ti.NW) issue_one<false>(tile_pieces(pw, lane, *ti.P, ti.rs, ti.ors), RING, j, N, lane);
}

template <bool FAST, bool KMASK>
__device__ __forceinline__ void key_wave(char* smem, int NKR, int NQ, int wave, int lane, const Ptrs& P, const AttnDims& dm, size_t rs,
                                         const TileIssue& tp, int bh, const uint4 (&kraw)[3][2], const uint4 (&vraw)[3][2]) {
  FST_DECL
  const int l15 = lane & 15, g = lane >> 4, N = dm.N;
  char* DS = smem + NKR * 128;
  char* RING = smem + NKR * 256;
  float* LD = (float*)(RING + NSLOT * SLOT_B);
  const int NQP = NQ * QB;
  const float sc = dm.scale * LOG2E, scale = dm.scale;
  // FAST: D = sum_d dO (O + residual) is formed by key waves 0-3, 8 rows of the NEXT step's query block each (lane = row 8 w + (l >> 3),
  // 16-byte chunk l & 7 of the landed dO / O / residual tiles: 3 reads, 8 v_dot2c_f32_bf16, 3 lane exchanges) at the top of a step, where
  // the score registers are dead.  On the helper (the step's critical path: it also owns all of W) the same sum was 1400 cycles per step.
  auto d_rows = [&](int j) __attribute__((always_inline)) {
    const char* Gs = RING + (j & (NSLOT - 1)) * SLOT_B + TILE_B;
    const int row = 8 * wave + (lane >> 3);
    const int off = row * 128 + (((lane & 7) ^ (((row >> 1) & 3) << 1)) << 4);
    const uint4 vg4 = *(const uint4*)(Gs + off), vo4 = *(const uint4*)(Gs + TILE_B + off), vr4 = *(const uint4*)(Gs + 2 * TILE_B + off);
    const uint32_t ug[4] = {vg4.x, vg4.y, vg4.z, vg4.w}, uo[4] = {vo4.x, vo4.y, vo4.z, vo4.w}, ur[4] = {vr4.x, vr4.y, vr4.z, vr4.w};
    float d0 = 0.f, d1 = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      d0 = mvit_dot2(uo[k], ug[k], d0);
      d1 = mvit_dot2(ur[k], ug[k], d1);
    }
    d0 += d1;
    d0 += __shfl_xor(d0, 1, 64);
    d0 += __shfl_xor(d0, 2, 64);
    d0 += __shfl_xor(d0, 4, 64);
    const int q = QB * j + row;
    if ((lane & 7) == 0) {
      LD[NQP + q] = d0 * scale;             // (padding rows: dO = 0 -> D = 0)
      if (q < N) P.Dv[(size_t)bh * N + q] = d0;
    }
  };

  // resident K / V fragments: B operand (column = key 48 w + 16 kt + l15, k = d = 32 ks + 8 g ..), requested by the kernel's first
  // instructions (frag_loads), ahead of the K-row and tile DMA
  bf16x8 kf[3][2], vf[3][2];
  float cinit[3];
#pragma unroll
  for (int kt = 0; kt < 3; ++kt) {
    const int key = KW * wave + 16 * kt + l15;
    cinit[kt] = key < N ? 0.f : -1e30f;      // S of a padding key starts at -1e30: P = exp2(-1e30 sc - L) = 0 with no masking instruction
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) kf[kt][ks] = *(const bf16x8*)&kraw[kt][ks], vf[kt][ks] = *(const bf16x8*)&vraw[kt][ks];
  }
  f32x4 dk[4][3], dv[4][3];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt)
#pragma unroll
    for (int kt = 0; kt < 3; ++kt) dk[dt][kt] = (f32x4){0.f, 0.f, 0.f, 0.f}, dv[dt][kt] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // lane constants of the step tile (32 rows x 128 B, 16-byte chunk c of row r at c ^ (((r >> 1) & 3) << 1))
  const int swA = ((l15 >> 1) & 3) << 1;                                   // fragment reads: row 16 qt + l15
  int offA[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) offA[ks] = l15 * 128 + (((4 * ks + g) ^ swA) << 4);
  const int rowT = 4 * g + (l15 >> 2), swT = ((rowT >> 1) & 3) << 1;       // transposing reads: rows 4 g + (l15 >> 2) (+ 16)
  int offT[4];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) offT[dt] = rowT * 128 + (((2 * dt + ((l15 & 3) >> 1)) ^ swT) << 4) + ((l15 & 1) << 3);
  // dS^T rows of this wave's keys: [key][32 q] bf16, 8-byte slot s (4 queries) at s ^ ((key >> 1) & 7) = s ^ (l15 >> 1)
  const int fds = l15 >> 1;
  const int offW0 = (KW * wave + l15) * 64 + ((g ^ fds) << 3), offW1 = (KW * wave + l15) * 64 + (((4 + g) ^ fds) << 3);

  union Frag { bf16x8 v; uint2 h[2]; } pb[3], dsb[3];
  // X + Y, one 16-query half at a time: S, dP (A = Q / dO rows from LDS) -> P, dS -> packed bf16 halves of the B operands of dV / dK
  // (k slots 0-3: q = 4 g + i, 4-7: q = 16 + 4 g + i) and of the dS^T rows; the f32 scores of only one half are ever live
  auto XY = [&](int j) __attribute__((always_inline)) {
    const char* Qs = RING + (j & (NSLOT - 1)) * SLOT_B;
    const char* Gs = Qs + TILE_B;
    char* DSj = DS + (j & 1) * NKR * 64;
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
      f32x4 st[3], dp[3];
#pragma unroll
      for (int kt = 0; kt < 3; ++kt) {
        const float ci = KMASK ? cinit[kt] : 0.f;
        st[kt] = (f32x4){ci, ci, ci, ci};
        dp[kt] = (f32x4){0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const bf16x8 aq = *(const bf16x8*)(Qs + qt * 2048 + offA[ks]);
        const bf16x8 ag = *(const bf16x8*)(Gs + qt * 2048 + offA[ks]);
#pragma unroll
        for (int kt = 0; kt < 3; ++kt) {
          st[kt] = mfma16(aq, kf[kt][ks], st[kt]);
          dp[kt] = mfma16(ag, vf[kt][ks], dp[kt]);
        }
      }
      // rows q = 32 j + 16 qt + 4 g + r: L and D as one float4 each; padding rows carry L = 1e30, D = 0
      const float4 L4 = *(const float4*)(LD + QB * j + 16 * qt + 4 * g), D4 = *(const float4*)(LD + NQP + QB * j + 16 * qt + 4 * g);
      const f32x2 nL[2] = {{-L4.x, -L4.y}, {-L4.z, -L4.w}}, nD[2] = {{-D4.x, -D4.y}, {-D4.z, -D4.w}};
#pragma unroll
      for (int kt = 0; kt < 3; ++kt) {
        const f32x2 x0 = fma2((f32x2){st[kt][0], st[kt][1]}, (f32x2){sc, sc}, nL[0]), x1 = fma2((f32x2){st[kt][2], st[kt][3]}, (f32x2){sc, sc}, nL[1]);
        const f32x2 p0 = {__builtin_amdgcn_exp2f(x0.x), __builtin_amdgcn_exp2f(x0.y)}, p1 = {__builtin_amdgcn_exp2f(x1.x), __builtin_amdgcn_exp2f(x1.y)};
        const f32x2 d0 = p0 * fma2((f32x2){dp[kt][0], dp[kt][1]}, (f32x2){scale, scale}, nD[0]);
        const f32x2 d1 = p1 * fma2((f32x2){dp[kt][2], dp[kt][3]}, (f32x2){scale, scale}, nD[1]);
        uint2 pw, dw;
        pw.x = pack2bf(p0.x, p0.y), pw.y = pack2bf(p1.x, p1.y);
        dw.x = pack2bf(d0.x, d0.y), dw.y = pack2bf(d1.x, d1.y);
        pb[kt].h[qt] = pw;
        dsb[kt].h[qt] = dw;
        *(uint2*)(DSj + kt * 1024 + (qt ? offW1 : offW0)) = dw;
      }
    }
  };
  // Z: dV^T += dO^T P, dK^T += Q^T dS (A = transposing reads of the step tile)
  auto Z = [&](int j) __attribute__((always_inline)) {
    const char* Qs = RING + (j & (NSLOT - 1)) * SLOT_B;
    const char* Gs = Qs + TILE_B;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      const bf16x8 ag = join(trd(Gs + offT[dt]), trd(Gs + offT[dt] + 2048));
      const bf16x8 aq = join(trd(Qs + offT[dt]), trd(Qs + offT[dt] + 2048));
#pragma unroll
      for (int kt = 0; kt < 3; ++kt) {
        dv[dt][kt] = mfma16(ag, pb[kt].v, dv[dt][kt]);
        dk[dt][kt] = mfma16(aq, dsb[kt].v, dk[dt][kt]);
      }
    }
  };
  // vmcnt(0): this wave's pieces of the tile two steps ahead have landed; lgkmcnt(0): s_barrier does not wait for LDS reads in flight
  // (see the forward kernel's step)
  auto bar = [&]() __attribute__((always_inline)) {
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    FST(3)
  };

  if (FAST) {
    if (wave < 4)
      asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");
    else
      asm volatile("s_waitcnt vmcnt(5) lgkmcnt(0)" ::: "memory");
  } else {
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  }
  __builtin_amdgcn_s_barrier();                                            // step tiles 0 / 1 and L are in LDS (not FAST: the K rows too)
  if (FAST && wave < 4) {
    d_rows(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  __builtin_amdgcn_s_barrier();                                            // ... and D of block 0 (from the landed tile 0)
  FST(0)
  for (int j = 0; j < NQ; ++j) {
    if (j + 2 < NQ) issue_pieces<FAST>(tp, RING, j + 2, N, lane);          // ring slot (j + 2) & 3 held tile j - 2, last read in step j - 2
    if (FAST && wave < 4 && j + 1 < NQ) d_rows(j + 1);                     // tile j + 1 landed before the previous step's barrier
    XY(j);
    FST(1)
    Z(j);
    FST(2)
    bar();
  }
  // dK / dV leave through LDS as full 128-byte rows: lane = key column with 4 consecutive d per accumulator -> [key][64 d] staging in
  // this wave's 6 KB of the K-row region (dead since step 0: the helper holds K^T in registers) -> 16 bytes per lane, 8 lanes per row
  {
    char* stg = smem + wave * (KW * 128);
    const size_t HD = (size_t)dm.H * 64;
#pragma unroll
    for (int m = 0; m < 2; ++m) {
#pragma unroll
      for (int kt = 0; kt < 3; ++kt)
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
          const f32x4 a = m ? dv[dt][kt] : dk[dt][kt];
          uint2 w;
          w.x = pack2bf(a[0], a[1]), w.y = pack2bf(a[2], a[3]);
          const int row = 16 * kt + l15, slot = (4 * dt + g) ^ (l15 & 15);           // 8-byte slot XOR row: conflict-free b64 writes
          *(uint2*)(stg + row * 128 + (slot << 3)) = w;
        }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        const int row = 8 * i + (lane >> 3), c = lane & 7;                          // 16-byte chunk c of the row = slots 2 c, 2 c + 1
        const uint2 lo = *(const uint2*)(stg + row * 128 + (((2 * c) ^ (row & 15)) << 3));
        const uint2 hi = *(const uint2*)(stg + row * 128 + (((2 * c + 1) ^ (row & 15)) << 3));
        const int key = KW * wave + row;
        if (key < N) *(uint4*)(P.dq + (size_t)key * rs + (1 + m) * HD + 8 * c) = make_uint4(lo.x, lo.y, hi.x, hi.y);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
  }
#ifdef MVIT_ATTN_TIMING
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
  FST(4)
  FST_WRITE(P, dm, wave, lane)
}

// The helper: D and the dQ product with K^T held in registers.  FULL: the K dimension of that product is the compile-time 11 steps of 7
// key waves (a run-time bound puts a branch around every step: each then waits for its own two reads -- 4400 cycles per query block
// instead of 1500).
template <bool FAST>
__device__ __forceinline__ void helper_wave(char* smem, int NKR, int NQ, int wave, int lane, const Ptrs& P, const AttnDims& dm, size_t rs,
                                            size_t ors, int bh, const TileIssue& tp) {
  FST_DECL
  const int N = dm.N;
  const int l15 = lane & 15, g = lane >> 4, r4 = l15 >> 2, c4 = l15 & 3;
  char* KS = smem;
  char* DS = smem + NKR * 128;
  char* RING = smem + NKR * 256;
  float* LD = (float*)(RING + NSLOT * SLOT_B);
  const int NQP = NQ * QB, nk32 = NKR >> 5;
  constexpr bool FULL = FAST;
  const bool has_res = FAST || P.orb != nullptr;
  // D = sum_d dO (O + residual) of query block j from its landed tile: lane = (row l >> 1, column half l & 1), 3 x 4 reads of 16 bytes
  auto d_block = [&](int j) __attribute__((always_inline)) {
    const char* Gs = RING + (j & (NSLOT - 1)) * SLOT_B + TILE_B;
    const int row = lane >> 1, sw = ((row >> 1) & 3) << 1;
    float dsum = 0.f, dsum2 = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int off = row * 128 + (((4 * (lane & 1) + i) ^ sw) << 4);
      const uint4 vg4 = *(const uint4*)(Gs + off), vo4 = *(const uint4*)(Gs + TILE_B + off);
      uint4 vr4 = make_uint4(0, 0, 0, 0);
      if (has_res) vr4 = *(const uint4*)(Gs + 2 * TILE_B + off);
      const uint32_t ug[4] = {vg4.x, vg4.y, vg4.z, vg4.w}, uo[4] = {vo4.x, vo4.y, vo4.z, vo4.w}, ur[4] = {vr4.x, vr4.y, vr4.z, vr4.w};
      // dO . O + dO . residual on v_dot2c_f32_bf16 (two bf16 products per instruction, f32 accumulate: products of bf16 are exact in f32)
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        dsum = mvit_dot2(uo[k], ug[k], dsum);
        dsum2 = mvit_dot2(ur[k], ug[k], dsum2);   // (no residual: zeros)
      }
    }
    dsum += dsum2;
    dsum += __shfl_xor(dsum, 1, 64);
    const int q = QB * j + row;
    if ((lane & 1) == 0) {
      LD[NQP + q] = dsum * dm.scale;        // (padding rows: dO = 0 -> D = 0)
      if (q < N) P.Dv[(size_t)bh * N + q] = dsum;
    }
  };
  // K^T operand fragments of the dQ product (A: row = d 16 dt + l15, k = keys 32 ks2 + 8 g ..) from the K rows in LDS:
  // 8-byte slot s of key k at s ^ (h(k) << 2), h = key bit 1 | key bit 3 << 1
  bf16x8 afr[4][MAXK32];
  auto load_afr = [&]() __attribute__((always_inline)) {
    const int krow = 8 * g + r4, hk = ((r4 >> 1) & 1) | ((g & 1) << 1);
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      const char* pa = KS + krow * 128 + (((4 * dt + c4) ^ (hk << 2)) << 3);
#pragma unroll
      for (int ks2 = 0; ks2 < MAXK32; ++ks2) {
        v4s lo = {0, 0, 0, 0}, hi = {0, 0, 0, 0};
        if (FULL || ks2 < nk32) lo = trd(pa + ks2 * 4096), hi = trd(pa + ks2 * 4096 + 512);
        afr[dt][ks2] = join(lo, hi);
      }
    }
  };
  // W: dQ^T (64 d x 32 q of query block jq) = K^T dS^T over all keys; dS^T rows [key][32 q] bf16, 8-byte slot s at s ^ ((key >> 1) & 7)
  auto W = [&](int jq) __attribute__((always_inline)) {
    const char* DSq = DS + (jq & 1) * NKR * 64;
    const int krow = 8 * g + r4, f0 = 4 * (g & 1) + (r4 >> 1);               // (key >> 1) & 7 of the first read's row, + 2 for the second's
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
      const char* pb0 = DSq + krow * 64 + (((4 * qt + c4) ^ f0) << 3);
      const char* pb1 = DSq + (krow + 4) * 64 + (((4 * qt + c4) ^ (f0 + 2)) << 3);
      f32x4 acc[4];
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) acc[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks2 = 0; ks2 < MAXK32; ++ks2) {
        if (FULL || ks2 < nk32) {
          const bf16x8 b = join(trd(pb0 + ks2 * 2048), trd(pb1 + ks2 * 2048));
#pragma unroll
          for (int dt = 0; dt < 4; ++dt) acc[dt] = mfma16(afr[dt][ks2], b, acc[dt]);
        }
      }
      const int q = QB * jq + 16 * qt + l15;
      if (q < N) {
#pragma unroll
        for (int dt = 0; dt < 4; dt += 2) {     // lanes hold d rows 16 dt + 4 g ..: two d tiles = two 8-byte stores
          uint2 w0, w1;
          w0.x = pack2bf(acc[dt][0], acc[dt][1]), w0.y = pack2bf(acc[dt][2], acc[dt][3]);
          w1.x = pack2bf(acc[dt + 1][0], acc[dt + 1][1]), w1.y = pack2bf(acc[dt + 1][2], acc[dt + 1][3]);
          *(uint2*)(P.dq + (size_t)q * rs + 16 * dt + 4 * g) = w0;
          *(uint2*)(P.dq + (size_t)q * rs + 16 * (dt + 1) + 4 * g) = w1;
        }
      }
    }
  };
  auto bar = [&]() __attribute__((always_inline)) {
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    FST(3)
  };
  for (int i = lane; i < NQP; i += 64) LD[i] = i < N ? P.lse[(size_t)bh * N + i] * LOG2E : 1e30f;   // padding rows: P = exp2(.. - 1e30) = 0
  if (FAST) {
    if (wave < 4)
      asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");
    else
      asm volatile("s_waitcnt vmcnt(5) lgkmcnt(0)" ::: "memory");
  } else {
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  }
  __builtin_amdgcn_s_barrier();                                  // every wave's share of tiles 0 / 1 is in LDS (not FAST: of the K rows too)
  if (!FAST) d_block(0);                                         // (FAST: key waves 0-3 form D)
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  FST(0)
  // step 0 has no dQ product: the K^T fragments are read instead -- FAST: after step 0's barrier, behind which every wave's K pieces
  // have landed (NQ == 1: right behind the loop)
  if (2 < NQ) issue_pieces<FAST>(tp, RING, 2, N, lane);
  if (!FAST && NQ > 1) d_block(1);
  if (!FAST) load_afr();
  FST(1)
  bar();
  if (FAST) load_afr();
  int j = 1;
  for (; j + 2 < NQ; ++j) {                                      // steady state: one basic block, so that the scheduler can put D's vector
    issue_pieces<FAST>(tp, RING, j + 2, N, lane);                      // work between the MFMAs of W
    if (!FAST) d_block(j + 1);                                   // tile j + 1: issued in step j - 1, landed before that step's barrier
    FST(1)
    W(j - 1);
    FST(2)
    bar();
  }
  for (; j < NQ; ++j) {
    if (!FAST && j + 1 < NQ) d_block(j + 1);
    FST(1)
    W(j - 1);
    FST(2)
    bar();
  }
  W(NQ - 1);
  FST(2)
#ifdef MVIT_ATTN_TIMING
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
  FST(4)
  FST_WRITE(P, dm, wave, lane)
}

template <bool FAST>
__global__ __launch_bounds__(512, 2) void attn_bwd_fused_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ o,
                                                                const bf16_t* __restrict__ ores, const bf16_t* __restrict__ dO,
                                                                const float* __restrict__ lse, float* __restrict__ Dv,
                                                                bf16_t* __restrict__ dqkv, AttnDims dm, int NKW) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int bh = blockIdx.x, b = (int)fast_div((unsigned)bh, (unsigned)dm.H, dm.h_magic), h = bh - b * dm.H;
  const int N = dm.N, NW = NKW + 1;
  const int NKR = (NKW * KW + 31) & ~31, NQ = (N + QB - 1) / QB;
  const size_t rs = (size_t)3 * dm.H * 64, ors = (size_t)dm.H * 64;
  Ptrs P;
  P.qb = qkv + (size_t)b * N * rs + (size_t)h * 64;
  P.kb = P.qb + ors;
  P.vb = P.kb + ors;
  P.dob = dO + (size_t)b * N * ors + (size_t)h * 64;
  P.ob = o + (size_t)b * N * ors + (size_t)h * 64;
  P.orb = ores ? ores + (size_t)b * N * ors + (size_t)h * 64 : nullptr;
  P.dq = dqkv + (size_t)b * N * rs + (size_t)h * 64;
  P.lse = lse;
  P.Dv = Dv;
  // key waves: the K / V fragment loads are the kernel's first memory instructions (the prologue is an HBM burst: 256 CUs x 145 KB;
  // behind the 5-6 K-row DMA pieces and two tile pieces per wave they left ~1000 cycles later)
  uint4 kraw[3][2], vraw[3][2];
  if (wave != NKW) {
    const int l15 = lane & 15, g = lane >> 4;
#pragma unroll
    for (int kt = 0; kt < 3; ++kt) {
      const int key = KW * wave + 16 * kt + l15;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        kraw[kt][ks] = make_uint4(0, 0, 0, 0), vraw[kt][ks] = make_uint4(0, 0, 0, 0);
        if (key < N) {
          kraw[kt][ks] = *(const uint4*)(P.kb + (size_t)key * rs + 32 * ks + 8 * g);
          vraw[kt][ks] = *(const uint4*)(P.vb + (size_t)key * rs + 32 * ks + 8 * g);
        }
      }
    }
  }
  // step tiles 0 and 1
  char* RING = smem + NKR * 256;
  TileIssue tp;
  tp.tp = tile_pieces(wave, lane, P, rs, ors);
  tp.P = &P, tp.rs = rs, tp.ors = ors, tp.wave = wave, tp.NW = NW;
  issue_pieces<FAST>(tp, RING, 0, N, lane);
  if (NQ > 1) issue_pieces<FAST>(tp, RING, 1, N, lane);
  {
    // rows of the dS^T buffers that no key wave writes (keys NKW * 48 .. NKR - 1): zero once, they are multiplied with zero K rows
    char* DS = smem + NKR * 128;
    const int pad0 = NKW * KW * 64, padn = (NKR - NKW * KW) * 64;
    for (int i = tid * 16; i < padn; i += 512 * 16) {
      *(uint4*)(DS + pad0 + i) = make_uint4(0, 0, 0, 0);
      *(uint4*)(DS + NKR * 64 + pad0 + i) = make_uint4(0, 0, 0, 0);
    }
    // K rows -> LDS for the helper's K^T fragments (all waves share the 1 KB pieces): 8-byte slot s of key k at s ^ (h(k) << 2).  LAST:
    // nothing needs them before the helper's fragment loads in step 1, so (FAST) the first barrier does not wait for them -- each
    // wave's pieces are covered by its vmcnt(0) in front of step 0's barrier
    const __amdgpu_buffer_rsrc_t rK = make_rsrc(P.kb);
    for (int m = wave; m < NKR / 8; m += NW) {
      const int row = 8 * m + (lane >> 3);
      const int hk = ((row >> 1) & 1) | (((row >> 3) & 1) << 1);
      const int c = (lane & 7) ^ (hk << 1);
      unsigned off = row < N ? ((unsigned)row * (unsigned)rs + (unsigned)c * 8u) * 2u : 0x80000000u;
      asm volatile("" : "+v"(off));
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rK, (lds_ptr)(smem + m * 1024), 16, off, 0, 0, 0);
    }
  }
  if (wave == NKW) {
    // the helper is the youngest wave of its SIMD: with priority it does not lose every issue arbitration to the key wave beside it
    // (priority outranks age; 67 -> 59 k cycles per pair while it also formed D; 65.1 / 64.1 / 63.2 us at priority 0 / 1 / 3 since)
    __builtin_amdgcn_s_setprio(3);
    helper_wave<FAST>(smem, NKR, NQ, wave, lane, P, dm, rs, ors, bh, tp);
  } else if (KW * (wave + 1) > N) {
    key_wave<FAST, true>(smem, NKR, NQ, wave, lane, P, dm, rs, tp, bh, kraw, vraw);
  } else {
    key_wave<FAST, false>(smem, NKR, NQ, wave, lane, P, dm, rs, tp, bh, kraw, vraw);
  }
}
}  // namespace fused

}  // namespace

static bool make_dims(AttnDims& dm, int B, int N, int H, int Dh, float scale) {
  const unsigned nx = (unsigned)((N + 127) / 128);
  const unsigned long long total = (unsigned long long)nx * (unsigned)B * (unsigned)H;
  const unsigned dmax = nx > (unsigned)H ? nx : (unsigned)H;
  if (total * dmax >= 0xffffffffull) return false;           // (fast_div's range; also keeps the 1-D grid far below its limit)
  dm = AttnDims{B, N, H, Dh, scale, nx, mvit_div_magic(nx), mvit_div_magic((unsigned)H)};
  return true;
}

// the one-pass backward takes every shape whose keys fit seven key waves; the two-kernel form keeps the rest (N = 1301 at 512 x 512 tiles)
static bool fused_bwd_ok(int N, int Dh) {
#ifdef MVIT_DEBUG_KNOBS
  static const int knob = getenv("MVIT_ATTN_FUSED") ? atoi(getenv("MVIT_ATTN_FUSED")) : 1;   // measurement library: 0 = the two-kernel form
  if (!knob) return false;
#endif
  return Dh == 64 && N <= fused::KW * fused::MAXKW;
}

extern "C" {

MVIT_API int mvit_attention_fwd(const void* qkv, void* out, void* out_res, float* lse, int B, int N, int H, int Dh, float scale,
                                mvit_stream_t stream) {
  MVIT_CLEAR_ERROR();
  if (B <= 0 || N <= 0 || H <= 0 || Dh <= 0 || Dh > 64 || (Dh & 7)) return MVIT_EINVAL;
  AttnDims dm;
  if (!make_dims(dm, B, N, H, Dh, scale)) return MVIT_EINVAL;
  hipLaunchKernelGGL(attn_fwd_kernel, dim3(((N + 127) / 128) * B * H), dim3(256), NRING_Q * 2 * TILE_BYTES, (hipStream_t)stream,
                     (const bf16_t*)qkv, (bf16_t*)out, (bf16_t*)out_res, lse, dm);
  return MVIT_LAUNCH_CHECK();
}

MVIT_API int mvit_attention_bwd(const void* qkv, const void* out, const void* out_res, const void* d_out, const float* lse,
                                float* dsum, void* dqkv, int B, int N, int H, int Dh, float scale, mvit_stream_t stream) {
  MVIT_CLEAR_ERROR();
  if (B <= 0 || N <= 0 || H <= 0 || Dh <= 0 || Dh > 64 || (Dh & 7)) return MVIT_EINVAL;
  AttnDims dm;
  if (!make_dims(dm, B, N, H, Dh, scale)) return MVIT_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  if (fused_bwd_ok(N, Dh)) {
    // one workgroup per (batch, head) pair: S and dP once, q / k / v / dO read once, dQ reduced inside the block in a fixed order
    const int NKW = (N + fused::KW - 1) / fused::KW, NKR = (NKW * fused::KW + 31) & ~31, NQP = ((N + fused::QB - 1) / fused::QB) * fused::QB;
    const size_t lds = (size_t)NKR * 256 + (size_t)fused::NSLOT * fused::SLOT_B + 2 * (size_t)NQP * 4;
    static mvit_per_device_size lds_raised_f[2];
    const bool fast = NKW == fused::MAXKW && out_res != nullptr;      // 8 waves and a residual of O: the instantiation without run-time conditions
    const void* fn = fast ? (const void*)fused::attn_bwd_fused_kernel<true> : (const void*)fused::attn_bwd_fused_kernel<false>;
    if (mvit_ensure_dynamic_lds(fn, lds, lds_raised_f[fast]) != MVIT_OK) return MVIT_EINVAL;
    if (fast)
      hipLaunchKernelGGL(fused::attn_bwd_fused_kernel<true>, dim3(B * H), dim3(64 * (NKW + 1)), lds, s, (const bf16_t*)qkv, (const bf16_t*)out,
                         (const bf16_t*)out_res, (const bf16_t*)d_out, lse, dsum, (bf16_t*)dqkv, dm, NKW);
    else
      hipLaunchKernelGGL(fused::attn_bwd_fused_kernel<false>, dim3(B * H), dim3(64 * (NKW + 1)), lds, s, (const bf16_t*)qkv, (const bf16_t*)out,
                         (const bf16_t*)out_res, (const bf16_t*)d_out, lse, dsum, (bf16_t*)dqkv, dm, NKW);
    return MVIT_LAUNCH_CHECK();
  }
  hipLaunchKernelGGL(attn_bwd_dq_kernel, dim3(((N + 127) / 128) * B * H), dim3(256), NRING_Q * 2 * TILE_BYTES, s, (const bf16_t*)qkv,
                     (const bf16_t*)out, (const bf16_t*)out_res, (const bf16_t*)d_out, lse, dsum, (bf16_t*)dqkv, dm);
  const size_t lds_kv = (size_t)NRING * 2 * TILE_BYTES + 2 * (size_t)(((N + KVB - 1) / KVB) * KVB) * 4;
  static mvit_per_device_size lds_raised;  // grow-only per device: the attribute is a per-function, per-device maximum
  if (mvit_ensure_dynamic_lds((const void*)attn_bwd_dkv_kernel, lds_kv, lds_raised) != MVIT_OK) return MVIT_EINVAL;
  hipLaunchKernelGGL(attn_bwd_dkv_kernel, dim3(((N + 127) / 128) * B * H), dim3(256), lds_kv, s, (const bf16_t*)qkv,
                     (const bf16_t*)d_out, lse, dsum, (bf16_t*)dqkv, dm);
  return MVIT_LAUNCH_CHECK();
}

#ifdef MVIT_DEBUG_KNOBS
// measurement library only (make dbg): the two backward kernels as separate launches, so that tools/attn_overlap.py can put them on
// two streams (the dK/dV kernel reads the D vector the dQ kernel writes: the tool keeps a D from an earlier serial run)
MVIT_API int mvit_attention_bwd_part(int which, const void* qkv, const void* out, const void* out_res, const void* d_out, const float* lse,
                                     float* dsum, void* dqkv, int B, int N, int H, int Dh, float scale, mvit_stream_t stream) {
  MVIT_CLEAR_ERROR();
  AttnDims dm;
  if (!make_dims(dm, B, N, H, Dh, scale)) return MVIT_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  if (which == 0) {
    hipLaunchKernelGGL(attn_bwd_dq_kernel, dim3(((N + 127) / 128) * B * H), dim3(256), NRING_Q * 2 * TILE_BYTES, s, (const bf16_t*)qkv,
                       (const bf16_t*)out, (const bf16_t*)out_res, (const bf16_t*)d_out, lse, dsum, (bf16_t*)dqkv, dm);
  } else {
    const size_t lds_kv = (size_t)NRING * 2 * TILE_BYTES + 2 * (size_t)(((N + KVB - 1) / KVB) * KVB) * 4;
    static mvit_per_device_size lds_raised;
    if (mvit_ensure_dynamic_lds((const void*)attn_bwd_dkv_kernel, lds_kv, lds_raised) != MVIT_OK) return MVIT_EINVAL;
    hipLaunchKernelGGL(attn_bwd_dkv_kernel, dim3(((N + 127) / 128) * B * H), dim3(256), lds_kv, s, (const bf16_t*)qkv,
                       (const bf16_t*)d_out, lse, dsum, (bf16_t*)dqkv, dm);
  }
  return MVIT_LAUNCH_CHECK();
}
#endif

}  // extern "C"
