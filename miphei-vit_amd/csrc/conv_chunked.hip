// Direct 3x3 convolution (stride 1, pad 1) with LDS-staged input tiles for the WIDE decoder layers: the input channels are walked in
// chunks of 32, the output channels in slices of 64.
//
// Reference ops: the conv of Fusion_Block (src/generators/mipheivit.py:76-93: nn.Conv2d(k=3, s=1, p=1, bias=False) + BatchNorm2d)
// for the stages 1728 -> 256 @ 32^2, 352 -> 128 @ 64^2 and 176 -> 64 @ 128^2 (tile 256^2), and their input gradients (the adjoint
// convolution on the flipped, transposed weights: 256 -> 1728, 128 -> 352, 64 -> 176).  conv_direct.hip keeps the whole weight
// tensor in LDS, which only the narrow full-resolution layer allows; here a work item is (8 x 32-pixel tile, 64-channel output
// slice) and a step is one 32-channel chunk of the input: the block DMAs the chunk's (8+2) x (32+2) halo tile (21.8 KB) and the
// chunk's nine [64 x 32] weight taps (36.9 KB, packed by mvit_conv3x3_chunked_pack in the LDS image layout) into a two-slot ring
// (buffer_load ... lds, counted vmcnt + raw s_barrier: the next step's operands fly while this step's 72 MFMAs per wave run), and
// all nine taps of the chunk read the halo tile from LDS -- the implicit-GEMM path re-gathered it through L2 for every tap.
// The MFMA mapping is conv_direct.hip's (transposed product, a lane owns one pixel, 8-byte epilogue stores straight from the
// accumulators); BatchNorm statistics are flushed per item into the block's own statistic slot.
// Items are ordered slice-major: the blocks that run at the same time share one slice's weights (L2-resident, <= 2 MB).
#include "common.hpp"
#include "../../include/miphei_hip.h"

namespace {

constexpr int CC_TH = 8, CC_TW = 32;          // output tile: 8 rows x 32 pixels, 4 waves x 2 rows
constexpr int CC_CK = 32, CC_NS = 64;         // input channels per step, output channels per item
constexpr int CC_CG = CC_CK / 8;              // 16-byte channel groups per pixel and chunk
constexpr int CC_WROW = CC_CK + 8;            // weight row stride (elements): bank spread of the ds_read_b128 weight fragments
constexpr int CC_WBYTES = 9 * CC_NS * CC_WROW * 2;                      // 46080: one (slice, chunk) weight block
constexpr int CC_WPIECES = (CC_WBYTES + 4095) / 4096;                   // DMA instructions per wave (4 waves x 1 KB each)
constexpr int CC_WBUF = CC_WPIECES * 4096;
constexpr int CC_XUNITS = (CC_TH + 2) * (CC_TW + 2) * CC_CG;            // 16-byte units of one halo tile chunk
constexpr int CC_XPIECES = (CC_XUNITS + 255) / 256;
constexpr int CC_XBUF = CC_XPIECES * 4096;
constexpr int CC_LDS = 2 * (CC_WBUF + CC_XBUF);

// bank spread of the pixel fragments (as conv_direct.hip, CG = 4): slot of channel group g of pixel pix is rotated by pix >> 2
__device__ __forceinline__ int cc_slot_of(int pix, int g) { return (g + (pix >> 2)) & (CC_CG - 1); }
__device__ __forceinline__ int cc_group_of(int pix, int slot) { return (slot - (pix >> 2)) & (CC_CG - 1); }

struct CcArgs {
  const bf16_t* X;       // [B, H, W, ldx] bf16, channels [0, Cin) used
  const bf16_t* Wp;      // [slices][chunks][9][64][40] bf16 (mvit_conv3x3_chunked_pack)
  bf16_t* Y;             // [B, H, W, ldy] bf16, channels [0, Cout) written
  double* stats;         // nullable: [nslots][2][Cout]
  int B, H, W, ldx, ldy, Cin, Cout, nslots;
};

__global__ __launch_bounds__(256) void conv3x3_chunked_kernel(const CcArgs p) {
  typedef __attribute__((address_space(3))) void* lds_ptr;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* Wb = smem;                          // 2 x weight block
  char* Xb = smem + 2 * CC_WBUF;            // 2 x halo tile chunk
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const int l31 = lane & 31, half = lane >> 5;

  const int tiles_x = (p.W + CC_TW - 1) / CC_TW, tiles_y = (p.H + CC_TH - 1) / CC_TH;
  const int ntiles = p.B * tiles_y * tiles_x;
  const int nchunk = (p.Cin + CC_CK - 1) / CC_CK, nslice = (p.Cout + CC_NS - 1) / CC_NS;
  const int nitems = ntiles * nslice;

  auto make_rsrc = [](const void* ptr) __attribute__((always_inline)) {
    const unsigned long long v = (unsigned long long)ptr;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return __builtin_amdgcn_make_buffer_rsrc((void*)(((unsigned long long)hi << 32) | lo), 0, 0x7fffffff, 0x00020000);
  };
  const __amdgpu_buffer_rsrc_t rsX = make_rsrc(p.X);

  // one step = (item, chunk): the chunk's halo tile and weight block into ring slot `buf`
  auto issue_step = [&](int item, int chunk, int buf) __attribute__((always_inline)) {
    const int slice = item / ntiles, t = item - slice * ntiles;
    const int tx = t % tiles_x, ty = (t / tiles_x) % tiles_y, b = t / (tiles_x * tiles_y);
    const int y0 = ty * CC_TH - 1, x0 = tx * CC_TW - 1;
#pragma unroll
    for (int i = 0; i < CC_XPIECES; ++i) {
      const int u = (i * 4 + wave_u) * 64 + lane;
      const int pix = u / CC_CG, cg = cc_group_of(pix, u - pix * CC_CG);
      const int r = pix / (CC_TW + 2), cc = pix - r * (CC_TW + 2);
      const int iy = y0 + r, ix = x0 + cc, ch = chunk * CC_CK + cg * 8;
      const bool ok = u < CC_XUNITS && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W && ch < p.Cin;
      unsigned off = ok ? (unsigned)((((size_t)b * p.H + iy) * p.W + ix) * p.ldx + ch) * 2u : 0x80000000u;
      asm volatile("" : "+v"(off));   // one unconditional DMA per piece (the vmcnt accounting counts them)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsX, (lds_ptr)(Xb + (size_t)buf * CC_XBUF + (size_t)(i * 4 + wave_u) * 1024), 16, off, 0, 0, 0);
    }
    const __amdgpu_buffer_rsrc_t rsW = make_rsrc((const char*)p.Wp + ((size_t)slice * nchunk + chunk) * CC_WBYTES);
#pragma unroll
    for (int i = 0; i < CC_WPIECES; ++i) {
      const unsigned byte = (unsigned)((i * 4 + wave_u) * 64 + lane) * 16u;
      unsigned off = byte < (unsigned)CC_WBYTES ? byte : 0x80000000u;
      asm volatile("" : "+v"(off));
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (lds_ptr)(Wb + (size_t)buf * CC_WBUF + (size_t)(i * 4 + wave_u) * 1024), 16, off, 0, 0, 0);
    }
  };
  constexpr int STEP_DMAS = CC_XPIECES + CC_WPIECES;

  int item = blockIdx.x;
  if (item >= nitems) return;
  issue_step(item, 0, 0);
  int buf = 0;
  const int rbase = 2 * wave;
  for (; item < nitems; item += gridDim.x) {
    f32x16 acc[2][2];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[m][nt][r] = 0.f;

    for (int chunk = 0; chunk < nchunk; ++chunk) {
      // the next step: next chunk of this item, or chunk 0 of the block's next item
      const bool last = chunk + 1 == nchunk;
      const int nitem = last ? item + gridDim.x : item, nchk = last ? 0 : chunk + 1;
      // (lgkmcnt(0): s_barrier does not wait for LDS reads in flight, and the MFMAs consuming the last reads may be scheduled below it;
      //  the slot is refilled by DMA right behind this barrier -- see attention.hip's step)
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();                 // every wave has finished reading the slot the next step goes into
      if (nitem < nitems) {
        issue_step(nitem, nchk, buf ^ 1);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(STEP_DMAS) : "memory");   // this step's operands have landed, the next step's fly
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      __builtin_amdgcn_s_barrier();                 // ... and so have the other waves' pieces [no LDS reads pending]: none issued since the barrier above
      const char* xs = Xb + (size_t)buf * CC_XBUF;
      const char* ws = Wb + (size_t)buf * CC_WBUF;
#pragma unroll
      for (int kk = 0; kk < CC_CK / 16; ++kk) {
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          bf16x8 xb[4];
#pragma unroll
          for (int rr = 0; rr < 4; ++rr) {
            const int pix = (rbase + rr) * (CC_TW + 2) + l31 + kx;
            xb[rr] = *(const bf16x8*)(xs + ((size_t)pix * CC_CG + cc_slot_of(pix, kk * 2 + half)) * 16);
          }
#pragma unroll
          for (int ky = 0; ky < 3; ++ky) {
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
              const bf16x8 wa = *(const bf16x8*)(ws + ((size_t)((ky * 3 + kx) * CC_NS + nt * 32 + l31) * CC_WROW + kk * 16 + half * 8) * 2);
              acc[0][nt] = mvit_mfma32(wa, xb[ky], acc[0][nt], 0, 0, 0);
              acc[1][nt] = mvit_mfma32(wa, xb[ky + 1], acc[1][nt], 0, 0, 0);
            }
          }
        }
      }
      buf ^= 1;
    }

    // epilogue: lane = pixel (l31), registers = channel groups of four: 8-byte stores; statistics of this item's 64 channels
    {
      const int slice = item / ntiles, t = item - slice * ntiles;
      const int tx = t % tiles_x, ty = (t / tiles_x) % tiles_y, b = t / (tiles_x * tiles_y);
      const int ox = tx * CC_TW + l31;
      float st_s[2][16], st_q[2][16];
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) st_s[nt][r] = st_q[nt][r] = 0.f;
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        const int oy = ty * CC_TH + rbase + m;
        const bool ok = oy < p.H && ox < p.W;
        bf16_t* yp = p.Y + (((size_t)b * p.H + oy) * p.W + ox) * p.ldy + slice * CC_NS;
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
          for (int q4 = 0; q4 < 4; ++q4) {
            const int n = slice * CC_NS + nt * 32 + 8 * q4 + 4 * half;
            const float v0 = acc[m][nt][4 * q4], v1 = acc[m][nt][4 * q4 + 1], v2 = acc[m][nt][4 * q4 + 2], v3 = acc[m][nt][4 * q4 + 3];
            if (ok && n < p.Cout) {
              uint2 o;
              o.x = pack2bf(v0, v1), o.y = pack2bf(v2, v3);
              *(uint2*)(yp + nt * 32 + 8 * q4 + 4 * half) = o;
              st_s[nt][4 * q4] += v0, st_s[nt][4 * q4 + 1] += v1, st_s[nt][4 * q4 + 2] += v2, st_s[nt][4 * q4 + 3] += v3;
              st_q[nt][4 * q4] += v0 * v0, st_q[nt][4 * q4 + 1] += v1 * v1, st_q[nt][4 * q4 + 2] += v2 * v2, st_q[nt][4 * q4 + 3] += v3 * v3;
            }
          }
      }
      if (p.stats) {
        // sum over the 32 pixels of each half wave, then over the 4 waves through LDS (fixed order), one f64 add per channel
        // into this block's slot.  Scratch: the ring slot just consumed (buf ^ 1 after the toggle; `buf` already holds the
        // prefetched first chunk of the next item); the step that refills it is issued behind the next loop barrier.
        float* red = (float*)(Xb + (size_t)(buf ^ 1) * CC_XBUF);    // [4 waves][2][64]
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                          // (all waves are past their last reads of that slot)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
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
              red[(wave * 2 + 0) * CC_NS + n] = s;
              red[(wave * 2 + 1) * CC_NS + n] = q;
            }
          }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (tid < CC_NS && slice * CC_NS + tid < p.Cout) {
          double s = 0., q = 0.;
#pragma unroll
          for (int w = 0; w < 4; ++w) s += red[(w * 2 + 0) * CC_NS + tid], q += red[(w * 2 + 1) * CC_NS + tid];
          double* st = p.stats + (size_t)(blockIdx.x % p.nslots) * 2 * p.Cout;
          atomicAdd(st + slice * CC_NS + tid, s);
          atomicAdd(st + p.Cout + slice * CC_NS + tid, q);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (the reads of `red` are back before the slot is refilled)
      }
    }
  }
}

// ------------------------------------------------------------------ weight gradient of the wide layers
// dWn[n][(ky,kx,c)] += sum_pixels dY[pixel][n] * X[pixel + (ky-1, kx-1)][c]   (f32; output-channel major, Cin_pad per tap)
// A block owns ONE (64-channel output slice, 32-channel input chunk) pair and walks a group of 8 x 32-pixel tiles: per tile it
// stages the chunk's halo tile of X (21.8 KB) and the slice's dY tile (32 KB) in a two-slot LDS ring, and its 8 waves -- wave =
// (output-channel tile of 32, quarter of the tile's pixels) -- keep the nine [32 x 32] tap blocks of their quarter in registers
// (144 accumulators) over ALL tiles of the block.  The contraction runs over pixels, so both MFMA operands are gathered with the
// transposing LDS read (ds_read_b64_tr_b16), as in conv_direct.hip's weight-gradient kernel.  At the end the four pixel quarters
// are added in quarter order through LDS and the block adds its [9][64][32] result to the output: with one tile group per pair
// (the 1728 -> 256 layer: 216 pairs) every element has a single writer; otherwise groups meet in f32 atomics.
typedef short v4s_cc __attribute__((ext_vector_type(4)));
__device__ __forceinline__ bf16x8 cc_join(v4s_cc a, v4s_cc b) {
  union { struct { v4s_cc lo, hi; } s; bf16x8 v; } u;
  u.s.lo = a;
  u.s.hi = b;
  return u.v;
}
// 4 consecutive rows (stride `rs` bytes) x this lane's column: lane i of a 16-lane group addresses row i>>2, columns 4*(i&3)..+3
// and receives column i of the 4 x 16 block
__device__ __forceinline__ v4s_cc cc_tr4(const char* base, int rs, int col16, int lane) {
  const int i = lane & 15;
  const char* q = base + (i >> 2) * rs + (col16 + 4 * (i & 3)) * 2;
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s_cc*)q);
}

constexpr int CW_THREADS = 512;
constexpr int CW_XPIECES = (CC_XUNITS + CW_THREADS - 1) / CW_THREADS;        // 3
constexpr int CW_XBUF = CW_XPIECES * CW_THREADS * 16;                        // 24576
constexpr int CW_YBYTES = CC_TH * CC_TW * CC_NS * 2;                         // 32768: dY tile [256 pixels][64 channels]
constexpr int CW_YPIECES = CW_YBYTES / (CW_THREADS * 16);                    // 4
constexpr int CW_LDS = 2 * (CW_XBUF + CW_YBYTES);
static_assert(2 * 9 * 1024 * 4 <= CW_LDS, "the quarter reduction reuses the ring");

struct CwcArgs {
  const bf16_t* X;     // [B, H, W, ldx], channels [0, Cin)
  const bf16_t* dY;    // [B, H, W, ldy], channels [0, Cout)
  float* dWn;          // [Cout][9 * Cin_pad] f32, accumulated into
  int B, H, W, ldx, ldy, Cin, Cout, Cin_pad, groups;
};

__global__ __launch_bounds__(CW_THREADS) void conv3x3_chunked_wgrad_kernel(const CwcArgs p) {
  typedef __attribute__((address_space(3))) void* lds_ptr;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* Xs = smem;                       // 2 x halo tile chunk [340 pixels][32 channels] (plain pixel-major image: tr reads)
  char* Ys = smem + 2 * CW_XBUF;         // 2 x dY tile [256 pixels][64 channels]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const int l31 = lane & 31, half = lane >> 5, sub = (lane >> 4) & 1;
  const int nt = wave & 1, quarter = wave >> 1;       // output-channel tile of the slice, pixel quarter (two tile rows)
  const int tiles_x = (p.W + CC_TW - 1) / CC_TW, tiles_y = (p.H + CC_TH - 1) / CC_TH;
  const int ntiles = p.B * tiles_y * tiles_x;
  const int nchunk = (p.Cin + CC_CK - 1) / CC_CK, nslice = (p.Cout + CC_NS - 1) / CC_NS;
  const int npair = nchunk * nslice;
  // Block -> (tile group, pair).  Round 5: the hardware places workgroup L on XCD L % 8, and the pairs that share operands should share
  // an L2: each XCD gets a CONTIGUOUS run of the (group, chunk, slice) order with the slice running fastest, so the nslice blocks that read
  // the same X chunk sit on one XCD, and that XCD streams each dY slice once for its ~npair / 8 / nslice chunks (before: pair = L % npair,
  // i.e. the blocks of one chunk on different XCDs -- 512 MB of fabric traffic per launch of the 1728 -> 256 layer against ~65 MB of
  // operands, profiles/r04_pmc_traffic.json).  Bijective for any block count (the XCD-contiguous renumbering of gemm_ws.hip's TileOrder).
  const int nb = gridDim.x, L = blockIdx.x, xq = nb >> 3, xr = nb & 7, xcd = L & 7;
  const int wg = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (L >> 3);
  const int group = wg / npair, pair = wg - group * npair;
  const int chunk = pair / nslice, slice = pair - chunk * nslice;
  auto rsrc = [](const void* ptr) __attribute__((always_inline)) {
    const unsigned long long v = (unsigned long long)ptr;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return __builtin_amdgcn_make_buffer_rsrc((void*)(((unsigned long long)hi << 32) | lo), 0, 0x7fffffff, 0x00020000);
  };
  const __amdgpu_buffer_rsrc_t rsX = rsrc(p.X), rsY = rsrc(p.dY);

  auto issue_tile = [&](int t, int buf) __attribute__((always_inline)) {
    const int tx = t % tiles_x, ty = (t / tiles_x) % tiles_y, b = t / (tiles_x * tiles_y);
    const int y0 = ty * CC_TH - 1, x0 = tx * CC_TW - 1;
#pragma unroll
    for (int i = 0; i < CW_XPIECES; ++i) {
      const int u = (i * 8 + wave_u) * 64 + lane;
      const int pix = u / CC_CG, cg = u - pix * CC_CG;
      const int r = pix / (CC_TW + 2), cc = pix - r * (CC_TW + 2);
      const int iy = y0 + r, ix = x0 + cc, ch = chunk * CC_CK + cg * 8;
      const bool ok = u < CC_XUNITS && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W && ch < p.Cin;
      unsigned off = ok ? (unsigned)((((size_t)b * p.H + iy) * p.W + ix) * p.ldx + ch) * 2u : 0x80000000u;
      asm volatile("" : "+v"(off));
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsX, (lds_ptr)(Xs + (size_t)buf * CW_XBUF + (size_t)(i * 8 + wave_u) * 1024), 16, off, 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < CW_YPIECES; ++i) {
      const int u = (i * 8 + wave_u) * 64 + lane;           // 16-byte unit of the dY tile: pixel u / 8, channel group u % 8
      const int pix = u >> 3, cg = u & 7;
      const int oy = ty * CC_TH + pix / CC_TW, ox = tx * CC_TW + pix % CC_TW, ch = slice * CC_NS + cg * 8;
      const bool ok = oy < p.H && ox < p.W && ch < p.Cout;   // pixels outside the image / channels beyond Cout: zero gradient
      unsigned off = ok ? (unsigned)((((size_t)b * p.H + oy) * p.W + ox) * p.ldy + ch) * 2u : 0x80000000u;
      asm volatile("" : "+v"(off));
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsY, (lds_ptr)(Ys + (size_t)buf * CW_YBYTES + (size_t)(i * 8 + wave_u) * 1024), 16, off, 0, 0, 0);
    }
  };
  constexpr int NDMA = CW_XPIECES + CW_YPIECES;

  f32x16 acc[9];
#pragma unroll
  for (int a = 0; a < 9; ++a)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;

  int t = group;
  if (t < ntiles) issue_tile(t, 0);
  int buf = 0;
  for (; t < ntiles; t += p.groups) {
    const int tn = t + p.groups;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // (reads of the previous tile finished, not just issued)
    __builtin_amdgcn_s_barrier();
    if (tn < ntiles) {
      issue_tile(tn, buf ^ 1);
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NDMA) : "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();   // [no LDS reads pending]: none issued since the barrier above
    const char* xs = Xs + (size_t)buf * CW_XBUF;
    const char* ys = Ys + (size_t)buf * CW_YBYTES;
    // this wave's four K steps: pixels [16 s, 16 s + 16) of the tile, s = 4 * quarter .. + 3 (row s >> 1, columns 16 (s & 1) ..);
    // K-slot order of the fragments: half 0 -> pixels {0-3, 8-11}, half 1 -> {4-7, 12-15} of the step, for both operands
    bf16x8 fa[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int s_ = 4 * quarter + k;
      const char* base = ys + (size_t)(16 * s_ + 4 * half) * (CC_NS * 2);
      fa[k] = cc_join(cc_tr4(base, CC_NS * 2, 32 * nt + 16 * sub, lane), cc_tr4(base + 8 * CC_NS * 2, CC_NS * 2, 32 * nt + 16 * sub, lane));
    }
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int ky = tap / 3, kx = tap - ky * 3;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int s_ = 4 * quarter + k;
        const int prow = (s_ >> 1) + ky, pcol = 16 * (s_ & 1) + kx + 4 * half;
        const char* base = xs + ((size_t)(prow * (CC_TW + 2) + pcol) * CC_CK) * 2;
        const bf16x8 fb = cc_join(cc_tr4(base, CC_CK * 2, 16 * sub, lane), cc_tr4(base + 8 * CC_CK * 2, CC_CK * 2, 16 * sub, lane));
        acc[tap] = mvit_mfma32(fa[k], fb, acc[tap], 0, 0, 0);
      }
    }
    buf ^= 1;
  }
  // the four pixel quarters of an output-channel tile, added in quarter order through LDS (the ring is free now)
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  float* red = (float*)smem;                        // [2 channel tiles][9 taps][16 registers][64 lanes]
  for (int qd = 0; qd < 4; ++qd) {
    if (quarter == qd) {
#pragma unroll
      for (int a = 0; a < 9; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float* q = red + ((size_t)(nt * 9 + a) * 16 + r) * 64 + lane;
          *q = (qd == 0 ? 0.f : *q) + acc[a][r];
        }
    }
    __syncthreads();
  }
  // D[n][c]: column c = lane & 31 of the chunk, row n = (r&3) + 8*(r>>2) + 4*half of the channel tile.  512 threads walk the
  // 2 x 9 x 1024 sums; consecutive lanes = consecutive input channels = consecutive addresses of the output-channel-major result
  for (int e = tid; e < 2 * 9 * 1024; e += CW_THREADS) {
    const int ln = e & 63, r = (e >> 6) & 15, a = (e >> 10) % 9, ntile = e / (9 * 1024);
    const int c = chunk * CC_CK + (ln & 31), n = slice * CC_NS + ntile * 32 + (r & 3) + 8 * (r >> 2) + 4 * (ln >> 5);
    if (c < p.Cin && n < p.Cout) atomicAdd(p.dWn + (size_t)n * (9 * p.Cin_pad) + a * p.Cin_pad + c, red[e]);
  }
}

// nn.Conv2d weight W[Cout][Cin][3][3] (f32) -> [slices][chunks][9][64][40] bf16:
//   mode 0 (forward):  n = output channel, k = input channel:            W[n][k][ky][kx]
//   mode 1 (dgrad):    n = input channel of the forward conv, k = its output channel, taps flipped:  W[k][n][2-ky][2-kx]
// n / k beyond the tensor and the +8 row pad are zero.
__global__ __launch_bounds__(256) void pack_conv_chunked_kernel(const float* __restrict__ W, bf16_t* __restrict__ out, int Cout, int Cin,
                                                                int N, int K, int mode) {
  const int nchunk = (K + CC_CK - 1) / CC_CK, nslice = (N + CC_NS - 1) / CC_NS;
  const long long total = (long long)nslice * nchunk * 9 * CC_NS * CC_WROW;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int kl = (int)(i % CC_WROW);
    long long r = i / CC_WROW;
    const int nl = (int)(r % CC_NS);
    r /= CC_NS;
    const int tap = (int)(r % 9);
    r /= 9;
    const int chunk = (int)(r % nchunk), slice = (int)(r / nchunk);
    const int n = slice * CC_NS + nl, k = chunk * CC_CK + kl, ky = tap / 3, kx = tap - ky * 3;
    float v = 0.f;
    if (kl < CC_CK && n < N && k < K) {
      if (mode == 0) v = W[((size_t)n * Cin + k) * 9 + ky * 3 + kx];
      else v = W[((size_t)k * Cin + n) * 9 + (2 - ky) * 3 + (2 - kx)];
    }
    out[i] = f2bf(v);
  }
}

// all chunked operands of a step in ONE launch (blockIdx.y = descriptor): the three forward and three input-gradient operands of the
// wide fusion blocks used to be six launches and six allocations in the prologue of every training step
struct CcPackMulti {
  mvit_cc_pack_desc d[MVIT_CC_PACK_MAX];
};
__global__ __launch_bounds__(256) void pack_conv_chunked_multi_kernel(const CcPackMulti pm) {
  const mvit_cc_pack_desc& d = pm.d[blockIdx.y];
  const int N = d.mode == 0 ? d.Cout : d.Cin, K = d.mode == 0 ? d.Cin : d.Cout;
  const int nchunk = (K + CC_CK - 1) / CC_CK, nslice = (N + CC_NS - 1) / CC_NS;
  const long long total = (long long)nslice * nchunk * 9 * CC_NS * CC_WROW;
  bf16_t* __restrict__ out = (bf16_t*)d.out;
  const float* __restrict__ W = d.W;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int kl = (int)(i % CC_WROW);
    long long r = i / CC_WROW;
    const int nl = (int)(r % CC_NS);
    r /= CC_NS;
    const int tap = (int)(r % 9);
    r /= 9;
    const int chunk = (int)(r % nchunk), slice = (int)(r / nchunk);
    const int n = slice * CC_NS + nl, k = chunk * CC_CK + kl, ky = tap / 3, kx = tap - ky * 3;
    float v = 0.f;
    if (kl < CC_CK && n < N && k < K) {
      if (d.mode == 0) v = W[((size_t)n * d.Cin + k) * 9 + ky * 3 + kx];
      else v = W[((size_t)k * d.Cin + n) * 9 + (2 - ky) * 3 + (2 - kx)];
    }
    out[i] = f2bf(v);
  }
}

}  // namespace

extern "C" {

MVIT_API int mvit_conv3x3_chunked_pack_multi(const mvit_cc_pack_desc* descs, int n, mvit_stream_t stream) {
  MVIT_CLEAR_ERROR();
  if (!descs || n <= 0 || n > MVIT_CC_PACK_MAX) return MVIT_EINVAL;
  CcPackMulti pm;
  long long most = 0;
  for (int i = 0; i < n; ++i) {
    const mvit_cc_pack_desc& d = descs[i];
    if (!d.W || !d.out || d.Cout <= 0 || d.Cin <= 0 || (d.mode != 0 && d.mode != 1)) return MVIT_EINVAL;
    pm.d[i] = d;
    const long long t = mvit_conv3x3_chunked_pack_elems(d.mode == 0 ? d.Cout : d.Cin, d.mode == 0 ? d.Cin : d.Cout);
    most = t > most ? t : most;
  }
  const long long blocks = (most + 255) / 256;
  hipLaunchKernelGGL(pack_conv_chunked_multi_kernel, dim3((unsigned)(blocks > 2048 ? 2048 : blocks), n), dim3(256), 0, (hipStream_t)stream, pm);
  return MVIT_LAUNCH_CHECK();
}

MVIT_API long long mvit_conv3x3_chunked_pack_elems(int N, int K) {
  return (long long)((N + CC_NS - 1) / CC_NS) * ((K + CC_CK - 1) / CC_CK) * 9 * CC_NS * CC_WROW;
}

MVIT_API int mvit_conv3x3_chunked_pack(const float* W, void* out, int Cout, int Cin, int mode, mvit_stream_t stream) {
  MVIT_CLEAR_ERROR();
  if (!W || !out || Cout <= 0 || Cin <= 0 || (mode != 0 && mode != 1)) return MVIT_EINVAL;
  const int N = mode == 0 ? Cout : Cin, K = mode == 0 ? Cin : Cout;
  const long long total = mvit_conv3x3_chunked_pack_elems(N, K);
  const long long blocks = (total + 255) / 256;
  hipLaunchKernelGGL(pack_conv_chunked_kernel, dim3((unsigned)(blocks > 4096 ? 4096 : blocks)), dim3(256), 0, (hipStream_t)stream, W,
                     (bf16_t*)out, Cout, Cin, N, K, mode);
  return MVIT_LAUNCH_CHECK();
}

MVIT_API int mvit_conv3x3_chunked(const void* X, const void* Wp, void* Y, double* stats, int nslots, int B, int H, int W, int Cin,
                                  int ldx, int Cout, int ldy, mvit_stream_t stream) {
  MVIT_CLEAR_ERROR();
  if (!X || !Wp || !Y || B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || (Cin & 7) || (Cout & 7) || (ldx & 7) || ldx < Cin ||
      (ldy & 3) || ldy < Cout || (stats && nslots <= 0))
    return MVIT_EINVAL;
  if ((size_t)B * H * W * ldx * 2 >= 0x7fffffffull) return MVIT_EINVAL;   // 32-bit byte offsets of the raw buffer
  CcArgs a{(const bf16_t*)X, (const bf16_t*)Wp, (bf16_t*)Y, stats, B, H, W, ldx, ldy, Cin, Cout, nslots};
  static mvit_per_device_size raised;
  if (mvit_ensure_dynamic_lds((const void*)conv3x3_chunked_kernel, CC_LDS, raised) != MVIT_OK) return MVIT_EINVAL;
  const long long items = (long long)B * ((H + CC_TH - 1) / CC_TH) * ((W + CC_TW - 1) / CC_TW) * ((Cout + CC_NS - 1) / CC_NS);
  int blocks = items < mvit_num_cus() ? (int)items : mvit_num_cus();            // one persistent block per CU (LDS)
  // >= 256 statistic slots = the deterministic mode's "one writer block per slot" contract (slot = blockIdx.x % nslots): keep it on
  // parts with more CUs than slots, as launch_one of gemm_kernel.hpp does for its STATS epilogue
  if (stats && nslots >= 256 && blocks > nslots) blocks = nslots;
  hipLaunchKernelGGL(conv3x3_chunked_kernel, dim3(blocks), dim3(256), CC_LDS, (hipStream_t)stream, a);
  return MVIT_LAUNCH_CHECK();
}

MVIT_API int mvit_conv3x3_chunked_wgrad(const void* X, const void* dY, float* dWn, int B, int H, int W, int Cin, int Cin_pad, int ldx,
                                        int Cout, int ldy, mvit_stream_t stream) {
  MVIT_CLEAR_ERROR();
  if (!X || !dY || !dWn || B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || (Cin & 7) || (Cout & 7) || Cin_pad < Cin ||
      (ldx & 7) || ldx < Cin || (ldy & 7) || ldy < Cout)
    return MVIT_EINVAL;
  if ((size_t)B * H * W * ldx * 2 >= 0x7fffffffull || (size_t)B * H * W * ldy * 2 >= 0x7fffffffull) return MVIT_EINVAL;
  const int ntiles = B * ((H + CC_TH - 1) / CC_TH) * ((W + CC_TW - 1) / CC_TW);
  const int npair = ((Cin + CC_CK - 1) / CC_CK) * ((Cout + CC_NS - 1) / CC_NS);
  int groups = mvit_num_cus() / npair;                 // tile groups per (slice, chunk) pair: one block per CU in total
  groups = groups < 1 ? 1 : (groups > ntiles ? ntiles : groups);
  CwcArgs a{(const bf16_t*)X, (const bf16_t*)dY, dWn, B, H, W, ldx, ldy, Cin, Cout, Cin_pad, groups};
  static mvit_per_device_size raised;
  if (mvit_ensure_dynamic_lds((const void*)conv3x3_chunked_wgrad_kernel, CW_LDS, raised) != MVIT_OK) return MVIT_EINVAL;
  hipLaunchKernelGGL(conv3x3_chunked_wgrad_kernel, dim3(npair * groups), dim3(CW_THREADS), CW_LDS, (hipStream_t)stream, a);
  return MVIT_LAUNCH_CHECK();
}

}  // extern "C"
