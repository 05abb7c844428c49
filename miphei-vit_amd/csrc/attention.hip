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
        st[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, qf[s], st[kt], 0, 0, 0);
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
        oacc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, pb, oacc[dt], 0, 0, 0);
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
        wr[dt][g].x = pack2bf(v0 - __uint_as_float(wo[dt][g].x << 16), v1 - __uint_as_float(wo[dt][g].x & 0xffff0000u));
        wr[dt][g].y = pack2bf(v2 - __uint_as_float(wo[dt][g].y << 16), v3 - __uint_as_float(wo[dt][g].y & 0xffff0000u));
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
      dsum += (__uint_as_float(ua[j] << 16) + __uint_as_float(ur[j] << 16)) * __uint_as_float(uc[j] << 16);
      dsum += (__uint_as_float(ua[j] & 0xffff0000u) + __uint_as_float(ur[j] & 0xffff0000u)) * __uint_as_float(uc[j] & 0xffff0000u);
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
          st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, qf[s], st, 0, 0, 0);
          const bf16x8 v = *(const bf16x8*)(Vs + swz(32 * kt + l31, 2 * s + half));
          dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v, dof[s], dp, 0, 0, 0);
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
            dqacc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, dsb, dqacc[dt], 0, 0, 0);
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
          st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, kf[s], st, 0, 0, 0);
          const bf16x8 g = *(const bf16x8*)(Ds + swz(32 * qt + l31, 2 * s + half));
          dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(g, vf[s], dp, 0, 0, 0);
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
            dvacc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, pb, dvacc[dt], 0, 0, 0);
            const bf16x8 a2 = join(tr_read4(Qs, qbase, cb, lane), tr_read4(Qs, qbase + 8, cb, lane));
            dkacc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, dsb, dkacc[dt], 0, 0, 0);
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

}  // namespace

static bool make_dims(AttnDims& dm, int B, int N, int H, int Dh, float scale) {
  const unsigned nx = (unsigned)((N + 127) / 128);
  const unsigned long long total = (unsigned long long)nx * (unsigned)B * (unsigned)H;
  const unsigned dmax = nx > (unsigned)H ? nx : (unsigned)H;
  if (total * dmax >= 0xffffffffull) return false;           // (fast_div's range; also keeps the 1-D grid far below its limit)
  dm = AttnDims{B, N, H, Dh, scale, nx, mvit_div_magic(nx), mvit_div_magic((unsigned)H)};
  return true;
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
