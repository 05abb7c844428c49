// Direct 3x3 convolution (stride 1, pad 1) with LDS-staged input tiles for the full-resolution decoder layers with few channels.
//
// Reference ops: the conv of Fusion_Block / Basic_Conv3x3 (src/generators/mipheivit.py:20-41,76-93: nn.Conv2d(k=3, s=1, p=1,
// bias=False) followed by BatchNorm2d) for the last fusion stage (67 -> 32 channels at the tile's full 256 x 256 resolution) and
// its input gradient (32 -> 64: the adjoint convolution on the flipped, transposed weights).  The implicit-GEMM path re-gathers
// the 3x3 window of every K tile through L2 (9x the input bytes as L2 -> LDS traffic, 0.2 PFLOP/s on this layer); here a block
// stages an (8+2) x (32+2) pixel halo tile ONCE in LDS (buffer_load ... lds, zero fill outside the image through out-of-range
// offsets, double-buffered across the tiles a persistent block walks) and all nine taps read it from there.
//
// MFMA mapping (v_mfma_f32_32x32x16_bf16): the product is computed transposed, D[n][pixel] = sum_k W[n][k] X[pixel][k] with
// A = weight fragment (row = output channel), B = pixel fragment (column = pixel of one 32-pixel row segment), so a lane owns ONE
// pixel and 4-channel groups of it: the epilogue stores 8-byte pieces straight from the accumulators (no LDS transposition) and
// BatchNorm statistics are per-lane register sums over all tiles of the block, reduced once at the end.
// A wave owns two output rows of the tile; per 16-channel K step and horizontal tap kx it reads the four input rows it needs
// once and uses them for the three vertical taps (12 pixel fragments + 9 weight fragments per 18 MFMAs).
#include "common.hpp"
#include "../../include/miphei_hip.h"

namespace {

constexpr int CD_TH = 8, CD_TW = 32;   // output tile: 8 rows x 32 pixels, 4 waves x 2 rows

template <int CIN>
struct CdGeom {
  static constexpr int CG = CIN / 8;                                  // 16-byte channel groups per pixel
  static constexpr int CINK = (CIN + 15) / 16 * 16;                   // K extent per tap (zero padded)
  static constexpr int WROW = CINK + 8;                               // weight row stride in LDS / global pack (bank spread)
  static constexpr int UNITS = (CD_TH + 2) * (CD_TW + 2) * CG;        // 16-byte units of one halo tile
  static constexpr int PIECES = (UNITS + 255) / 256;                  // DMA instructions per wave per tile
  static constexpr int TILE_BYTES = PIECES * 256 * 16;                // (slack: the last piece may run past the tile)
  // Bank spread of the pixel fragments: 16 consecutive pixels of a ds_read_b128 lane group must hit 16 distinct 16-byte columns
  // of the 256-byte bank row.  Column = (CG * pix + slot) mod 16: a permutation for odd CG (72 channels: CG = 9); for CG = 1, 2, 4,
  // 8 the slot of channel group g is rotated by pix >> SH (the rotation is applied to the SOURCE group of the DMA, the LDS image
  // itself stays lane-linear).
  static constexpr bool POW2 = (CG & (CG - 1)) == 0;
  static constexpr int SH = CG >= 16 ? 0 : (CG == 8 ? 1 : CG == 4 ? 2 : CG == 2 ? 3 : 4);
  static __device__ __forceinline__ int slot_of(int pix, int g) { return POW2 && CG > 1 ? (g + (pix >> SH)) & (CG - 1) : g; }
  static __device__ __forceinline__ int group_of(int pix, int slot) { return POW2 && CG > 1 ? (slot - (pix >> SH)) & (CG - 1) : slot; }
};

template <int CIN, int COUT>
constexpr size_t cd_lds_bytes() {
  return (size_t)9 * COUT * CdGeom<CIN>::WROW * 2 + 2 * (size_t)CdGeom<CIN>::TILE_BYTES + 64;
}

struct CdArgs {
  const bf16_t* X;       // [B, H, W, ldx] bf16, CIN channels used per pixel
  const bf16_t* Wp;      // [9][COUT][WROW] packed weights (mvit_pack_conv3x3_direct)
  bf16_t* Y;             // [B, H, W, ldy] bf16, COUT channels written per pixel
  double* stats;         // nullable: [nslots][2][COUT] per-channel sum / sum of squares of the f32 results
  int B, H, W, ldx, ldy, nslots;
};

template <int CIN, int COUT>
__global__ __launch_bounds__(256) void conv3x3_direct_kernel(const CdArgs p) {
  using G = CdGeom<CIN>;
  constexpr int NT = COUT / 32, NKK = G::CINK / 16;
  typedef __attribute__((address_space(3))) void* lds_ptr;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* Ws = smem;                                              // [9][COUT][WROW] bf16
  char* Xs = smem + (size_t)9 * COUT * G::WROW * 2;             // 2 x halo tile
  char* zero16 = Xs + 2 * (size_t)G::TILE_BYTES;                // 16 zero bytes (+ pad): source of the padded half K step
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const int l31 = lane & 31, half = lane >> 5;

  // weights: one pass, 16-byte pieces (the pack is already in the LDS image layout)
  {
    const int n16 = 9 * COUT * G::WROW / 8;
    for (int i = tid; i < n16; i += 256) ((uint4*)Ws)[i] = ((const uint4*)p.Wp)[i];
    if (tid < 4) ((uint4*)zero16)[tid] = make_uint4(0, 0, 0, 0);
    __syncthreads();      // (no DMA in flight yet: the plain barrier with its waits is what is wanted here)
  }

  const int tiles_x = (p.W + CD_TW - 1) / CD_TW, tiles_y = (p.H + CD_TH - 1) / CD_TH;
  const int ntiles = p.B * tiles_y * tiles_x;
  // whole input as one raw buffer (offsets are bytes; out-of-image pixels use an out-of-range offset -> zeros)
  const unsigned long long xv = (unsigned long long)p.X;
  const unsigned xlo = __builtin_amdgcn_readfirstlane((unsigned)xv), xhi = __builtin_amdgcn_readfirstlane((unsigned)(xv >> 32));
  const __amdgpu_buffer_rsrc_t rsX =
      __builtin_amdgcn_make_buffer_rsrc((void*)(((unsigned long long)xhi << 32) | xlo), 0, 0x7fffffff, 0x00020000);

  auto issue_tile = [&](int t, int buf) __attribute__((always_inline)) {
    const int tx = t % tiles_x, ty = (t / tiles_x) % tiles_y, b = t / (tiles_x * tiles_y);
    const int y0 = ty * CD_TH - 1, x0 = tx * CD_TW - 1;
#pragma unroll
    for (int i = 0; i < G::PIECES; ++i) {
      const int u = (i * 4 + wave_u) * 64 + lane;              // 16-byte unit of the halo tile image
      const int pix = u / G::CG, cg = G::group_of(pix, u - pix * G::CG);
      const int r = pix / (CD_TW + 2), cc = pix - r * (CD_TW + 2);
      const int iy = y0 + r, ix = x0 + cc;
      const bool ok = u < G::UNITS && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
      unsigned off = ok ? (unsigned)((((size_t)b * p.H + iy) * p.W + ix) * p.ldx + cg * 8) * 2u : 0x80000000u;
      asm volatile("" : "+v"(off));   // one unconditional DMA per piece (the vmcnt accounting counts them)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsX, (lds_ptr)(Xs + (size_t)buf * G::TILE_BYTES + (size_t)(i * 4 + wave_u) * 1024), 16,
                                               off, 0, 0, 0);
    }
  };

  // per-lane BatchNorm partial sums over every tile of this block: channel n = nt*32 + (r&3) + 8*(r>>2) + 4*half
  float st_s[NT][16], st_q[NT][16];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int r = 0; r < 16; ++r) st_s[nt][r] = st_q[nt][r] = 0.f;

  int t = blockIdx.x;
  if (t < ntiles) issue_tile(t, 0);
  int buf = 0;
  for (; t < ntiles; t += gridDim.x) {
    const int tn = t + gridDim.x;
    // (lgkmcnt(0): s_barrier does not wait for LDS reads in flight and the MFMAs consuming the last reads may be scheduled below it;
    //  the buffer is refilled by DMA right behind this barrier -- see attention.hip's step)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                 // every wave has finished reading the buffer the next tile goes into
    if (tn < ntiles) {
      issue_tile(tn, buf ^ 1);
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(G::PIECES) : "memory");   // this tile's pieces have landed, the next tile's fly
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();                 // ... and so have the other waves' pieces [no LDS reads pending]: none issued since the barrier above
    const char* xs = Xs + (size_t)buf * G::TILE_BYTES;

    f32x16 acc[2][NT];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[m][nt][r] = 0.f;

    // pixel (row ry of the halo tile, column l31 + kx), channels kk*16 + half*8 ..; rows ry = 2*wave .. 2*wave + 3
    const int rbase = 2 * wave;
#pragma unroll
    for (int kk = 0; kk < NKK; ++kk) {
      const bool kzero = (kk * 16 + half * 8) >= CIN;     // padded half of the last K step: zero pixel fragment (zero weights too)
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        bf16x8 xb[4];
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
          const int pix = (rbase + rr) * (CD_TW + 2) + l31 + kx;
          const char* q = kzero ? zero16 : xs + ((size_t)pix * G::CG + G::slot_of(pix, kk * 2 + half)) * 16;
          xb[rr] = *(const bf16x8*)q;
        }
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) {
            const bf16x8 wa = *(const bf16x8*)(Ws + ((size_t)((ky * 3 + kx) * COUT + nt * 32 + l31) * G::WROW + kk * 16 + half * 8) * 2);
            acc[0][nt] = mvit_mfma32(wa, xb[ky], acc[0][nt], 0, 0, 0);
            acc[1][nt] = mvit_mfma32(wa, xb[ky + 1], acc[1][nt], 0, 0, 0);
          }
        }
      }
    }

    // epilogue: lane = pixel (l31), registers = channel groups of four: 8-byte stores, statistics in registers
    {
      const int tx = t % tiles_x, ty = (t / tiles_x) % tiles_y, b = t / (tiles_x * tiles_y);
      const int ox = tx * CD_TW + l31;
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        const int oy = ty * CD_TH + rbase + m;
        const bool ok = oy < p.H && ox < p.W;
        bf16_t* yp = p.Y + (((size_t)b * p.H + oy) * p.W + ox) * p.ldy;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
          for (int q4 = 0; q4 < 4; ++q4) {
            const float v0 = acc[m][nt][4 * q4], v1 = acc[m][nt][4 * q4 + 1], v2 = acc[m][nt][4 * q4 + 2], v3 = acc[m][nt][4 * q4 + 3];
            if (ok) {
              uint2 o;
              o.x = pack2bf(v0, v1), o.y = pack2bf(v2, v3);
              *(uint2*)(yp + nt * 32 + 8 * q4 + 4 * half) = o;
              st_s[nt][4 * q4] += v0, st_s[nt][4 * q4 + 1] += v1, st_s[nt][4 * q4 + 2] += v2, st_s[nt][4 * q4 + 3] += v3;
              st_q[nt][4 * q4] += v0 * v0, st_q[nt][4 * q4 + 1] += v1 * v1, st_q[nt][4 * q4 + 2] += v2 * v2, st_q[nt][4 * q4 + 3] += v3 * v3;
            }
          }
      }
    }
    buf ^= 1;
  }

  if (p.stats) {
    // sum over the 32 lanes (pixels) of each half, then over the 4 waves through LDS, then one f64 atomic per channel
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    float* red = (float*)Xs;     // [4 waves][2][COUT]
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float s = st_s[nt][r], q = st_q[nt][r];
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) {
          s += __shfl_xor(s, o, 64);
          q += __shfl_xor(q, o, 64);
        }
        if (l31 == 0) {
          const int n = nt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
          red[(wave * 2 + 0) * COUT + n] = s;
          red[(wave * 2 + 1) * COUT + n] = q;
        }
      }
    __syncthreads();
    if (tid < COUT) {
      double s = 0., q = 0.;
#pragma unroll
      for (int w = 0; w < 4; ++w) s += red[(w * 2 + 0) * COUT + tid], q += red[(w * 2 + 1) * COUT + tid];
      double* st = p.stats + (size_t)(blockIdx.x % p.nslots) * 2 * COUT;
      atomicAdd(st + tid, s);
      atomicAdd(st + COUT + tid, q);
    }
  }
}

// ------------------------------------------------------------------ weight gradient
// dWn[n][(ky,kx,c)] += sum_pixels X[pixel + (ky-1, kx-1)][c] * dY[pixel][n]   (f32 atomics; output-channel major, so that the
// lanes of a wave -- consecutive c -- add to consecutive addresses).
// Same staging as the forward kernel (halo tile of X plus the 8 x 32 tile of dY, both DMA'd and double-buffered); the contraction
// runs over PIXELS, so both MFMA operands are gathered with the transposing LDS read (ds_read_b64_tr_b16: 4 consecutive pixels of
// one channel per lane).  D[n][c] tiles: 9 taps x ceil(CIN / 32) channel tiles, dealt round-robin to the 4 waves, which keep them
// in registers over every tile of the persistent block (the dY fragments of a tile, 16 K steps, are read once per wave and held
// in registers); one pass of atomics per block at the end.
typedef short v4s_cd __attribute__((ext_vector_type(4)));
__device__ __forceinline__ bf16x8 cd_join(v4s_cd a, v4s_cd b) {
  union { struct { v4s_cd lo, hi; } s; bf16x8 v; } u;
  u.s.lo = a;
  u.s.hi = b;
  return u.v;
}
// 4 consecutive rows (stride `rs` bytes) x this lane's column: lane i of a 16-lane group addresses row i>>2, columns 4*(i&3)..+3
// and receives column i of the 4 x 16 block
__device__ __forceinline__ v4s_cd cd_tr4(const char* base, int rs, int col16, int lane) {
  const int i = lane & 15;
  const char* q = base + (i >> 2) * rs + (col16 + 4 * (i & 3)) * 2;
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s_cd*)q);
}

template <int CIN, int COUT>
struct CwGeom {
  static constexpr int CT = (CIN + 31) / 32;                 // 32-channel tiles of the input channels
  static constexpr int NPAIR = 9 * CT;                       // (tap, channel tile) accumulators in total
  static constexpr int PER_WAVE = (NPAIR + 3) / 4;
  static constexpr int DY_BYTES = CD_TH * CD_TW * COUT * 2;  // dY tile
  static constexpr int DY_PIECES = DY_BYTES / (256 * 16);
  static_assert(DY_BYTES % (256 * 16) == 0, "dY tile must be whole DMA rounds");
};

template <int CIN, int COUT>
constexpr size_t cw_lds_bytes() {
  return 2 * (size_t)CdGeom<CIN>::TILE_BYTES + 2 * (size_t)CwGeom<CIN, COUT>::DY_BYTES + 512;
}

struct CwArgs {
  const bf16_t* X;     // [B, H, W, ldx]
  const bf16_t* dY;    // [B, H, W, ldy], COUT channels
  float* dWt;          // [COUT][9 * CIN] f32 (output-channel major, k = tap * CIN + c), accumulated into
  int B, H, W, ldx, ldy;
};

template <int CIN, int COUT>
__global__ __launch_bounds__(256) void conv3x3_direct_wgrad_kernel(const CwArgs p) {
  static_assert(COUT == 32, "one 32-row MFMA tile of output channels");
  using G = CdGeom<CIN>;
  using Wg = CwGeom<CIN, COUT>;
  typedef __attribute__((address_space(3))) void* lds_ptr;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* Xs = smem;                                     // 2 x halo tile of X (pixel-major, CIN channels, no channel rotation here)
  char* Ys = smem + 2 * (size_t)G::TILE_BYTES;         // 2 x dY tile [256 pixels][COUT]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const int l31 = lane & 31, half = lane >> 5, sub = (lane >> 4) & 1;
  const int tiles_x = (p.W + CD_TW - 1) / CD_TW, tiles_y = (p.H + CD_TH - 1) / CD_TH;
  const int ntiles = p.B * tiles_y * tiles_x;
  auto rsrc = [](const void* ptr) __attribute__((always_inline)) {
    const unsigned long long v = (unsigned long long)ptr;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return __builtin_amdgcn_make_buffer_rsrc((void*)(((unsigned long long)hi << 32) | lo), 0, 0x7fffffff, 0x00020000);
  };
  const __amdgpu_buffer_rsrc_t rsX = rsrc(p.X), rsY = rsrc(p.dY);

  auto issue_tile = [&](int t, int buf) __attribute__((always_inline)) {
    const int tx = t % tiles_x, ty = (t / tiles_x) % tiles_y, b = t / (tiles_x * tiles_y);
    const int y0 = ty * CD_TH - 1, x0 = tx * CD_TW - 1;
#pragma unroll
    for (int i = 0; i < G::PIECES; ++i) {
      const int u = (i * 4 + wave_u) * 64 + lane;
      const int pix = u / G::CG, cg = u - pix * G::CG;
      const int r = pix / (CD_TW + 2), cc = pix - r * (CD_TW + 2);
      const int iy = y0 + r, ix = x0 + cc;
      const bool ok = u < G::UNITS && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
      unsigned off = ok ? (unsigned)((((size_t)b * p.H + iy) * p.W + ix) * p.ldx + cg * 8) * 2u : 0x80000000u;
      asm volatile("" : "+v"(off));
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsX, (lds_ptr)(Xs + (size_t)buf * G::TILE_BYTES + (size_t)(i * 4 + wave_u) * 1024), 16,
                                               off, 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < Wg::DY_PIECES; ++i) {
      const int u = (i * 4 + wave_u) * 64 + lane;     // 16-byte unit of the dY tile: pixel u / (COUT/8), group u % (COUT/8)
      const int pix = u / (COUT / 8), cg = u - pix * (COUT / 8);
      const int oy = ty * CD_TH + pix / CD_TW, ox = tx * CD_TW + pix % CD_TW;
      const bool ok = oy < p.H && ox < p.W;             // pixels outside the image contribute zero gradient
      unsigned off = ok ? (unsigned)((((size_t)b * p.H + oy) * p.W + ox) * p.ldy + cg * 8) * 2u : 0x80000000u;
      asm volatile("" : "+v"(off));
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsY, (lds_ptr)(Ys + (size_t)buf * Wg::DY_BYTES + (size_t)(i * 4 + wave_u) * 1024), 16,
                                               off, 0, 0, 0);
    }
  };
  constexpr int NDMA = G::PIECES + Wg::DY_PIECES;

  f32x16 acc[Wg::PER_WAVE];
#pragma unroll
  for (int a = 0; a < Wg::PER_WAVE; ++a)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;

  int t = blockIdx.x;
  if (t < ntiles) issue_tile(t, 0);
  int buf = 0;
  for (; t < ntiles; t += gridDim.x) {
    const int tn = t + gridDim.x;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // (reads of the previous tile finished, not just issued)
    __builtin_amdgcn_s_barrier();
    if (tn < ntiles) {
      issue_tile(tn, buf ^ 1);
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NDMA) : "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();   // [no LDS reads pending]: none issued since the barrier above
    const char* xs = Xs + (size_t)buf * G::TILE_BYTES;
    const char* ys = Ys + (size_t)buf * Wg::DY_BYTES;
    // A operand (rows = output channels n, K = pixels): the 16 K steps of the tile, held in registers for all pairs of this wave.
    // K step s = pixels [16 s, 16 s + 16) of the tile (row s >> 1, columns 16 (s & 1) ..); K-slot order of the fragments:
    // half 0 -> pixels {0-3, 8-11}, half 1 -> {4-7, 12-15} of the step (the same order is used for the B operand).
    bf16x8 fa[16];
#pragma unroll
    for (int s_ = 0; s_ < 16; ++s_) {
      const char* base = ys + (size_t)(16 * s_ + 4 * half) * (COUT * 2);
      fa[s_] = cd_join(cd_tr4(base, COUT * 2, 16 * sub, lane), cd_tr4(base + 8 * COUT * 2, COUT * 2, 16 * sub, lane));
    }
#pragma unroll
    for (int a = 0; a < Wg::PER_WAVE; ++a) {
      const int pair = a * 4 + wave;                   // (tap, channel tile) of this accumulator
      if (pair < Wg::NPAIR) {
        const int tap = pair / Wg::CT, ct = pair - tap * Wg::CT;
        const int ky = tap / 3, kx = tap - ky * 3;
        const int cb = ct * 32 + 16 * sub;             // this lane group's 16 channels
        const bool cok = cb < CIN;                     // (CIN % 16 == 8: the last 16-channel group is half padding, handled below)
#pragma unroll
        for (int s_ = 0; s_ < 16; ++s_) {
          const int prow = (s_ >> 1) + ky, pcol = 16 * (s_ & 1) + kx + 4 * half;
          const char* base = xs + ((size_t)(prow * (CD_TW + 2) + pcol) * CIN) * 2;
          bf16x8 fb;
          if (cok) {
            fb = cd_join(cd_tr4(base, CIN * 2, cb, lane), cd_tr4(base + 8 * CIN * 2, CIN * 2, cb, lane));
          } else {
            const uint4 z = make_uint4(0, 0, 0, 0);
            fb = *(const bf16x8*)&z;
          }
          acc[a] = mvit_mfma32(fa[s_], fb, acc[a], 0, 0, 0);
        }
      }
    }
    buf ^= 1;
  }
  // D[n][c]: column c = lane & 31 of the channel tile, row n = (r&3) + 8*(r>>2) + 4*half
#pragma unroll
  for (int a = 0; a < Wg::PER_WAVE; ++a) {
    const int pair = a * 4 + wave;
    if (pair >= Wg::NPAIR) continue;
    const int tap = pair / Wg::CT, ct = pair - tap * Wg::CT;
    const int c = ct * 32 + l31;
    if (c >= CIN) continue;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int n = (r & 3) + 8 * (r >> 2) + 4 * half;
      atomicAdd(p.dWt + (size_t)n * (9 * CIN) + tap * CIN + c, acc[a][r]);   // lanes -> consecutive c: one line per half wave
    }
  }
}

// nn.Conv2d weight [Cout, Cin, 3, 3] f32 -> [9][NOUT][WROW] bf16 for the direct kernel.
//   mode 0 (forward):  out[tap][n][c]  = W[n][(c + rot) % Cin][ky][kx]                      n < Cout, c < Cin
//   mode 1 (dgrad):    out[tap][ci][co] = W[co][(ci + rot) % Cin][2 - ky][2 - kx]            ci < NOUT (<= Cin), co < Cout
// everything else (channel padding, the +8 row pad) is zero.
__global__ __launch_bounds__(256) void pack_conv_direct_kernel(const float* __restrict__ W, bf16_t* __restrict__ out, int Cout,
                                                               int Cin, int NOUT, int KIN, int WROW, int rot, int mode) {
  const int total = 9 * NOUT * WROW;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
    const int k = i % WROW, n = (i / WROW) % NOUT, tap = i / (WROW * NOUT);
    const int ky = tap / 3, kx = tap - ky * 3;
    float v = 0.f;
    if (k < KIN) {
      if (mode == 0) {
        if (n < Cout) v = W[((size_t)n * Cin + (k + rot) % Cin) * 9 + ky * 3 + kx];
      } else {
        v = W[((size_t)k * Cin + (n + rot) % Cin) * 9 + (2 - ky) * 3 + (2 - kx)];
      }
    }
    out[i] = f2bf(v);
  }
}

template <int CIN, int COUT>
int launch_direct(const CdArgs& a, hipStream_t s) {
  const size_t lds = cd_lds_bytes<CIN, COUT>();
  static mvit_per_device_size raised;
  auto kern = conv3x3_direct_kernel<CIN, COUT>;
  if (mvit_ensure_dynamic_lds((const void*)kern, lds, raised) != MVIT_OK) return MVIT_EINVAL;
  const int tiles = a.B * ((a.H + CD_TH - 1) / CD_TH) * ((a.W + CD_TW - 1) / CD_TW);
  const int blocks = tiles < mvit_num_cus() ? tiles : mvit_num_cus();      // one persistent block per CU (LDS)
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), lds, s, a);
  return MVIT_LAUNCH_CHECK();
}

template <int CIN, int COUT>
int launch_direct_wgrad(const CwArgs& a, hipStream_t s) {
  const size_t lds = cw_lds_bytes<CIN, COUT>();
  static mvit_per_device_size raised;
  auto kern = conv3x3_direct_wgrad_kernel<CIN, COUT>;
  if (mvit_ensure_dynamic_lds((const void*)kern, lds, raised) != MVIT_OK) return MVIT_EINVAL;
  const int tiles = a.B * ((a.H + CD_TH - 1) / CD_TH) * ((a.W + CD_TW - 1) / CD_TW);
  const int blocks = tiles < mvit_num_cus() ? tiles : mvit_num_cus();
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), lds, s, a);
  return MVIT_LAUNCH_CHECK();
}

}  // namespace

extern "C" {

MVIT_API int mvit_conv3x3_direct_wgrad(const void* X, const void* dY, float* dWt, int B, int H, int W, int Cin_pad, int ldx, int Cout,
                                       int ldy, mvit_stream_t stream) {
  MVIT_CLEAR_ERROR();
  if (!X || !dY || !dWt || B <= 0 || H <= 0 || W <= 0 || (ldx & 7) || ldx < Cin_pad || (ldy & 7) || ldy < Cout) return MVIT_EINVAL;
  if ((size_t)B * H * W * ldx * 2 >= 0x7fffffffull || (size_t)B * H * W * ldy * 2 >= 0x7fffffffull) return MVIT_EINVAL;
  CwArgs a{(const bf16_t*)X, (const bf16_t*)dY, dWt, B, H, W, ldx, ldy};
  if (Cout == 32 && Cin_pad == 72) return launch_direct_wgrad<72, 32>(a, (hipStream_t)stream);
  if (Cout == 32 && Cin_pad == 32) return launch_direct_wgrad<32, 32>(a, (hipStream_t)stream);
  if (Cout == 32 && Cin_pad == 8) return launch_direct_wgrad<8, 32>(a, (hipStream_t)stream);
  return MVIT_EINVAL;
}

MVIT_API int mvit_conv3x3_direct_supported(int Cin_pad, int Cout) {
  return (Cout == 32 && (Cin_pad == 72 || Cin_pad == 64 || Cin_pad == 32 || Cin_pad == 8)) || (Cout == 64 && Cin_pad == 32);
}

MVIT_API int mvit_conv3x3_direct(const void* X, const void* Wp, void* Y, double* stats, int nslots, int B, int H, int W, int Cin_pad,
                                 int ldx, int Cout, int ldy, mvit_stream_t stream) {
  MVIT_CLEAR_ERROR();
  if (!X || !Wp || !Y || B <= 0 || H <= 0 || W <= 0 || (ldx & 7) || ldx < Cin_pad || (ldy & 3) || ldy < Cout || (stats && nslots <= 0))
    return MVIT_EINVAL;
  if ((size_t)B * H * W * ldx * 2 >= 0x7fffffffull) return MVIT_EINVAL;   // 32-bit byte offsets of the raw buffer
  CdArgs a{(const bf16_t*)X, (const bf16_t*)Wp, (bf16_t*)Y, stats, B, H, W, ldx, ldy, nslots};
  hipStream_t s = (hipStream_t)stream;
  if (Cout == 32) {
    if (Cin_pad == 72) return launch_direct<72, 32>(a, s);
    if (Cin_pad == 64) return launch_direct<64, 32>(a, s);
    if (Cin_pad == 32) return launch_direct<32, 32>(a, s);
    if (Cin_pad == 8) return launch_direct<8, 32>(a, s);
  } else if (Cout == 64) {
    if (Cin_pad == 32) return launch_direct<32, 64>(a, s);
  }
  return MVIT_EINVAL;
}

MVIT_API int mvit_pack_conv3x3_direct(const float* W, void* out, int Cout, int Cin, int n_out, int k_in, int k_pad, int rot,
                                      int mode, mvit_stream_t stream) {
  MVIT_CLEAR_ERROR();
  if (!W || !out || Cout <= 0 || Cin <= 0 || n_out <= 0 || k_in <= 0 || k_pad < k_in || (k_pad & 7) || rot < 0 ||
      (mode != 0 && mode != 1))
    return MVIT_EINVAL;
  if (mode == 0 && (k_in > Cin || n_out < Cout)) return MVIT_EINVAL;
  if (mode == 1 && (k_in > Cout || n_out > Cin)) return MVIT_EINVAL;
  const int wrow = (k_pad + 15) / 16 * 16 + 8;
  const int total = 9 * n_out * wrow;
  hipLaunchKernelGGL(pack_conv_direct_kernel, dim3((total + 255) / 256 > 1024 ? 1024 : (total + 255) / 256), dim3(256), 0,
                     (hipStream_t)stream, W, (bf16_t*)out, Cout, Cin, n_out, k_in, wrow, rot, mode);
  return MVIT_LAUNCH_CHECK();
}

}  // extern "C"
