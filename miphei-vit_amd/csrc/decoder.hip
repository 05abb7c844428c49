// Decoder-side data movement kernels (all HBM-bound, NHWC bf16, 16-byte accesses per lane):
//   tap-table resampler  - bilinear x2 upsample (Fusion_Block, src/generators/mipheivit.py:89), bicubic
//                          token-grid regrid (Encoder.forward, mipheivit.py:147-151,161-162) and the
//                          adjoints of both, with BatchNorm+ReLU optionally applied to the gathered source
//                          and the result written straight into a channel slice of the concat buffer
//   BatchNorm statistics finalisation / apply / backward (nn.BatchNorm2d in Basic_Conv3x3, mipheivit.py:33)
//   bf16 transpose (operand of the head-conv weight-gradient GEMM).
#include "common.hpp"
#include "../../include/miphei_hip.h"

namespace {

__device__ __forceinline__ void unpack8(const uint4& t, float (&f)[8]) {
  const uint32_t u[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    f[2 * j] = lo16f(u[j]);
    f[2 * j + 1] = hi16f(u[j]);
  }
}
__device__ __forceinline__ uint4 pack8f(const float (&f)[8]) {
  uint4 r;
  r.x = pack2bf(f[0], f[1]);
  r.y = pack2bf(f[2], f[3]);
  r.z = pack2bf(f[4], f[5]);
  r.w = pack2bf(f[6], f[7]);
  return r;
}

// ------------------------------------------------------------------ separable tap-table resampler
struct ResampleArgs {
  const bf16_t* src;
  bf16_t* dst;
  const int* ty_idx;
  const float* ty_w;
  const int* tx_idx;
  const float* tx_w;
  const float* scale;  // optional: relu(v*scale[c] + shift[c]) applied to every gathered source value
  const float* shift;
  int B, h, w, H, W, C, ld_src, ld_dst, T;
  long long src_bstride, dst_bstride;
};

// blockIdx.y = output row, blockIdx.z = batch: the row taps are block-uniform (scalar loads); the column taps of a thread's pixel are read
// once, in front of the tap loops (TT = taps per axis when it is <= 4, else 0: dynamic count).
template <int TT>
__global__ __launch_bounds__(256) void resample_kernel(const ResampleArgs a) {
  const int T = TT ? TT : a.T;
  const int cv = a.C >> 3;
  const int b = blockIdx.z, oy = blockIdx.y;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= a.W * cv) return;
  const int ox = i / cv, c8 = i - ox * cv;
  float acc[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) acc[j] = 0.f;
  float sc[8], sh[8];
  if (a.scale) {
#pragma unroll
    for (int j = 0; j < 8; ++j) sc[j] = a.scale[c8 * 8 + j], sh[j] = a.shift[c8 * 8 + j];
  }
  constexpr int TR = TT ? TT : 1;
  float wxr[TR];
  int ixr[TR];
  if (TT) {
#pragma unroll
    for (int tx = 0; tx < TR; ++tx) wxr[tx] = a.tx_w[ox * T + tx], ixr[tx] = a.tx_idx[ox * T + tx];
  }
  const bf16_t* sb = a.src + (size_t)b * a.src_bstride + c8 * 8;
#pragma unroll
  for (int ty = 0; ty < (TT ? TT : 16); ++ty) {
    if (!TT && ty >= T) break;
    const float wy = a.ty_w[oy * T + ty];
    if (wy == 0.f) continue;
    const int iy = a.ty_idx[oy * T + ty];
    auto tap = [&](float wx, int ix) __attribute__((always_inline)) {
      const uint4 t = *(const uint4*)(sb + ((size_t)iy * a.w + ix) * a.ld_src);
      float f[8];
      unpack8(t, f);
      const float wgt = wy * wx;
      if (a.scale) {
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] += wgt * fmaxf(f[j] * sc[j] + sh[j], 0.f);
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] += wgt * f[j];
      }
    };
    if (TT) {
#pragma unroll
      for (int tx = 0; tx < TR; ++tx)
        if (wxr[tx] != 0.f) tap(wxr[tx], ixr[tx]);
    } else {
      for (int tx = 0; tx < T; ++tx) {
        const float wx = a.tx_w[ox * T + tx];
        if (wx != 0.f) tap(wx, a.tx_idx[ox * T + tx]);
      }
    }
  }
  *(uint4*)(a.dst + (size_t)b * a.dst_bstride + ((size_t)oy * a.W + ox) * a.ld_dst + c8 * 8) = pack8f(acc);
}

// ------------------------------------------------------------------ bilinear x2 (align_corners=False), specialised
// One thread per CELL between four neighbouring source pixels (y, y+1) x (x, x+1), y in [-1, h-1], x in [-1, w-1] (clamped at the
// border) and 8 channels: the four outputs (2y+1, 2y+2) x (2x+1, 2x+2) depend on exactly these four sources with the weights
// 0.75 / 0.25 - four 16-byte loads and one BatchNorm+ReLU per source for four outputs, against 16 loads and 16 applications for
// the same four outputs in the tap-table kernel above (114 -> 45 us for the 64-channel 128^2 -> 256^2 stage at batch 16).
// extra8: optional NHWC bf16 buffer with 8 channels per output pixel, copied behind the C up-sampled channels (the image slice of
// the last concat buffer: D0 skip of Detail_Capture.forward, mipheivit.py:208-211).
struct Up2Args {
  const bf16_t* src;
  bf16_t* dst;
  const float* scale;
  const float* shift;
  const bf16_t* extra8;
  int B, h, w, C, ld_src, ld_dst;
  long long src_bstride, dst_bstride;
};
__global__ __launch_bounds__(256) void upsample2x_kernel(const Up2Args a) {
  const int cv = a.C >> 3;
  const int cw = a.w + 1, chh = a.h + 1;
  const long long total = (long long)a.B * chh * cw * cv;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int c8 = (int)(i % cv);
    long long r = i / cv;
    const int cx = (int)(r % cw) - 1;
    r /= cw;
    const int cy = (int)(r % chh) - 1, b = (int)(r / chh);
    const int y0 = max(cy, 0), y1 = min(cy + 1, a.h - 1), x0 = max(cx, 0), x1 = min(cx + 1, a.w - 1);
    const bf16_t* sb = a.src + (size_t)b * a.src_bstride + c8 * 8;
    float s[4][8];
    unpack8(*(const uint4*)(sb + ((size_t)y0 * a.w + x0) * a.ld_src), s[0]);
    unpack8(*(const uint4*)(sb + ((size_t)y0 * a.w + x1) * a.ld_src), s[1]);
    unpack8(*(const uint4*)(sb + ((size_t)y1 * a.w + x0) * a.ld_src), s[2]);
    unpack8(*(const uint4*)(sb + ((size_t)y1 * a.w + x1) * a.ld_src), s[3]);
    if (a.scale) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float sc = a.scale[c8 * 8 + j], sh = a.shift[c8 * 8 + j];
#pragma unroll
        for (int q = 0; q < 4; ++q) s[q][j] = fmaxf(s[q][j] * sc + sh, 0.f);
      }
    }
    // rows: t = 0.75 top + 0.25 bottom (output row 2cy+1), u = 0.25 top + 0.75 bottom (row 2cy+2); then the same across x
    float o[4][8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float tl = 0.75f * s[0][j] + 0.25f * s[2][j], tr = 0.75f * s[1][j] + 0.25f * s[3][j];
      const float bl = 0.25f * s[0][j] + 0.75f * s[2][j], br = 0.25f * s[1][j] + 0.75f * s[3][j];
      o[0][j] = 0.75f * tl + 0.25f * tr;
      o[1][j] = 0.25f * tl + 0.75f * tr;
      o[2][j] = 0.75f * bl + 0.25f * br;
      o[3][j] = 0.25f * bl + 0.75f * br;
    }
    const int W2 = 2 * a.w, H2 = 2 * a.h;
    bf16_t* db = a.dst + (size_t)b * a.dst_bstride;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int oy = 2 * cy + 1 + (q >> 1), ox = 2 * cx + 1 + (q & 1);
      if (oy < 0 || oy >= H2 || ox < 0 || ox >= W2) continue;
      bf16_t* px = db + ((size_t)oy * W2 + ox) * a.ld_dst;
      *(uint4*)(px + c8 * 8) = pack8f(o[q]);
      if (a.extra8 && c8 == 0) *(uint4*)(px + a.C) = *(const uint4*)(a.extra8 + (((size_t)b * H2 + oy) * W2 + ox) * 8);
    }
  }
}

// Adjoint of the x2 bilinear map (gradient w.r.t. its input): source pixel y collects output rows 2y-1 .. 2y+2 with the weights
// 0.25, 0.75, 0.75, 0.25; at the border the clamped tap folds back (row 0 counts 1.0 for y = 0, row 2h-1 counts 1.0 for y = h-1).
// Separable, no tap tables: 16 loads per output vector and nothing else to wait for (92 -> 40 us at 256^2 -> 128^2, 64 channels).
__global__ __launch_bounds__(256) void upsample2x_adjoint_kernel(const Up2Args a) {
  const int cv = a.C >> 3;
  const long long total = (long long)a.B * a.h * a.w * cv;
  const int W2 = 2 * a.w, H2 = 2 * a.h;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int c8 = (int)(i % cv);
    long long r = i / cv;
    const int x = (int)(r % a.w);
    r /= a.w;
    const int y = (int)(r % a.h), b = (int)(r / a.h);
    float wy[4], wx[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float base = (k == 0 || k == 3) ? 0.25f : 0.75f;
      const int ry = 2 * y - 1 + k, rx = 2 * x - 1 + k;
      wy[k] = (ry < 0 || ry >= H2) ? 0.f : ((k == 1 && y == 0) || (k == 2 && y == a.h - 1)) ? 1.f : base;
      wx[k] = (rx < 0 || rx >= W2) ? 0.f : ((k == 1 && x == 0) || (k == 2 && x == a.w - 1)) ? 1.f : base;
    }
    const bf16_t* sb = a.src + (size_t)b * a.src_bstride + c8 * 8;
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.f;
#pragma unroll
    for (int ky = 0; ky < 4; ++ky) {
      const int ry = min(max(2 * y - 1 + ky, 0), H2 - 1);   // (clamped rows / columns carry weight 0)
      float row[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) row[j] = 0.f;
#pragma unroll
      for (int kx = 0; kx < 4; ++kx) {
        const int rx = min(max(2 * x - 1 + kx, 0), W2 - 1);
        float f[8];
        unpack8(*(const uint4*)(sb + ((size_t)ry * W2 + rx) * a.ld_src), f);
#pragma unroll
        for (int j = 0; j < 8; ++j) row[j] += wx[kx] * f[j];
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] += wy[ky] * row[j];
    }
    *(uint4*)(a.dst + (size_t)b * a.dst_bstride + ((size_t)y * a.w + x) * a.ld_dst + c8 * 8) = pack8f(acc);
  }
}

// ------------------------------------------------------------------ NCHW f32 image -> NHWC bf16 channel slice
__global__ __launch_bounds__(256) void image_to_nhwc_kernel(const float* __restrict__ img, bf16_t* __restrict__ dst, int B,
                                                            int S, int C, int ld_dst, int nzero) {
  const long long total = (long long)B * S * S;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const long long pix = i % ((long long)S * S);
    const int b = (int)(i / ((long long)S * S));
    bf16_t* o = dst + (size_t)i * ld_dst;
    if (C + nzero == 8 && (ld_dst & 7) == 0 && ((uintptr_t)dst & 15) == 0) {   // the 8-channel image buffer: one 16-byte store per pixel
      float f[8];
#pragma unroll
      for (int c = 0; c < 8; ++c) f[c] = c < C ? img[((size_t)b * C + c) * S * S + pix] : 0.f;
      *(uint4*)o = pack8f(f);
      continue;
    }
    for (int c = 0; c < C; ++c) o[c] = f2bf(img[((size_t)b * C + c) * S * S + pix]);
    for (int c = 0; c < nzero; ++c) o[C + c] = 0;
  }
}

// ------------------------------------------------------------------ BatchNorm: statistics -> scale / shift
// stats: [nslots][2][C] doubles (sum, sum of squares) accumulated by the conv epilogue.
// 32 lanes per channel: lane k sums slots k, k + 32, ... and the partial sums meet through a shuffle tree (one load latency
// instead of nslots dependent ones: 13 -> 4 us for a kernel that runs 7 times per step).
__global__ __launch_bounds__(256) void bn_finalize_kernel(const double* __restrict__ stats, const float* __restrict__ gamma,
                                   const float* __restrict__ beta, float* __restrict__ rmean, float* __restrict__ rvar,
                                   float* __restrict__ scale, float* __restrict__ shift, float* __restrict__ mean_o,
                                   float* __restrict__ rstd_o, int C, int nslots, double count, float eps, float momentum,
                                   int training) {
  const int c = blockIdx.x * 8 + (threadIdx.x >> 5), k0 = threadIdx.x & 31;
  const bool ok = c < C;
  double mean = 0., var = 0.;
  if (training) {
    double s = 0., q = 0.;
    if (ok)
      for (int k = k0; k < nslots; k += 32) {
        s += stats[(size_t)k * 2 * C + c];
        q += stats[(size_t)k * 2 * C + C + c];
      }
#pragma unroll
    for (int d = 16; d >= 1; d >>= 1) {
      s += __shfl_xor(s, d, 32);
      q += __shfl_xor(q, d, 32);
    }
    mean = s / count;
    var = q / count - mean * mean;
    if (var < 0.) var = 0.;
  }
  if (!ok || k0 != 0) return;
  if (training) {
    const double unb = count > 1. ? var * count / (count - 1.) : var;
    rmean[c] = (float)((1. - momentum) * rmean[c] + momentum * mean);
    rvar[c] = (float)((1. - momentum) * rvar[c] + momentum * unb);
  } else {
    mean = rmean[c];
    var = rvar[c];
  }
  const double rstd = 1. / sqrt(var + (double)eps);
  const double sc = gamma[c] * rstd;
  scale[c] = (float)sc;
  shift[c] = (float)(beta[c] - mean * sc);
  if (mean_o) mean_o[c] = (float)mean;
  if (rstd_o) rstd_o[c] = (float)rstd;
}

// Counter-based dropout mask (nn.Dropout after the ReLU of Conv2DBlock / Deconv2DBlock, src/generators/unet.py:441-519): element
// e of a [M, C] activation is kept iff the 16-bit field (e & 3) of splitmix64(seed + (e >> 2) * golden) is >= p * 65536; kept
// values are scaled by 1 / (1 - p).  Forward and backward recompute the same mask from (seed, element index): nothing is stored.
struct DropCfg {
  unsigned long long seed;
  unsigned thresh;   // p * 65536 (0: dropout off)
  float inv_keep;
};
__device__ __forceinline__ unsigned long long splitmix64(unsigned long long z) {
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
// multipliers of 8 consecutive elements starting at element index e0 (e0 % 8 == 0)
__device__ __forceinline__ void drop8(const DropCfg& d, unsigned long long e0, float (&f)[8]) {
  if (d.thresh == 0) return;
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const unsigned long long z = splitmix64(d.seed + ((e0 >> 2) + h) * 0x9E3779B97F4A7C15ull);
#pragma unroll
    for (int j = 0; j < 4; ++j) f[4 * h + j] *= ((unsigned)(z >> (16 * j)) & 0xffffu) >= d.thresh ? d.inv_keep : 0.f;
  }
}

__global__ __launch_bounds__(256) void bn_relu_apply_kernel(const bf16_t* __restrict__ x, const float* __restrict__ scale,
                                                            const float* __restrict__ shift, bf16_t* __restrict__ out,
                                                            long long M, int C, int ld_x, int ld_out, DropCfg drop) {
  const int cv = C >> 3;
  const long long total = M * cv;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int c8 = (int)(i % cv);
    const long long m = i / cv;
    float f[8];
    unpack8(*(const uint4*)(x + (size_t)m * ld_x + c8 * 8), f);
#pragma unroll
    for (int j = 0; j < 8; ++j) f[j] = fmaxf(f[j] * scale[c8 * 8 + j] + shift[c8 * 8 + j], 0.f);
    drop8(drop, (unsigned long long)m * C + c8 * 8, f);
    *(uint4*)(out + (size_t)m * ld_out + c8 * 8) = pack8f(f);
  }
}

// ------------------------------------------------------------------ BatchNorm+ReLU backward
// reduce: per channel  s1 = sum g, s2 = sum g*xhat,  g = dy * [x*scale+shift > 0]
__global__ __launch_bounds__(256) void bn_relu_bwd_reduce_kernel(const bf16_t* __restrict__ dy, int ld_dy,
                                                                 const bf16_t* __restrict__ x, const float* __restrict__ scale,
                                                                 const float* __restrict__ shift,
                                                                 const float* __restrict__ mean, const float* __restrict__ rstd,
                                                                 double* __restrict__ stats, long long M, int C, int nslots,
                                                                 DropCfg drop) {
  // per-thread partial sums meet in a FIXED order (no LDS atomics): part[row group][2C], then one thread per statistic adds the
  // row groups in ascending order -> the block's contribution does not depend on wave scheduling
  __shared__ float part[256 * 16];
  const int cv = C >> 3;
  const int c8 = threadIdx.x % cv, rl = threadIdx.x / cv, rpb = 256 / cv;
  float s1[8], s2[8], sc[8], sh[8], mu[8], rs[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    s1[j] = s2[j] = 0.f;
    sc[j] = scale[c8 * 8 + j], sh[j] = shift[c8 * 8 + j], mu[j] = mean[c8 * 8 + j], rs[j] = rstd[c8 * 8 + j];
  }
  if (rl < rpb) {
    for (long long m = (long long)blockIdx.x * rpb + rl; m < M; m += (long long)gridDim.x * rpb) {
      float g[8], xv[8];
      unpack8(*(const uint4*)(dy + (size_t)m * ld_dy + c8 * 8), g);
      unpack8(*(const uint4*)(x + (size_t)m * C + c8 * 8), xv);
      drop8(drop, (unsigned long long)m * C + c8 * 8, g);   // gradient through the dropout that followed the ReLU
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float gg = (xv[j] * sc[j] + sh[j] > 0.f) ? g[j] : 0.f;
        s1[j] += gg;
        s2[j] += gg * (xv[j] - mu[j]) * rs[j];
      }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      part[rl * 2 * C + c8 * 8 + j] = s1[j];
      part[rl * 2 * C + C + c8 * 8 + j] = s2[j];
    }
  }
  __syncthreads();
  // one writer block per slot when nslots >= gridDim.x (the deterministic configuration); otherwise slots are shared
  double* st = stats + (size_t)(blockIdx.x % nslots) * 2 * C;
  for (int i = threadIdx.x; i < 2 * C; i += 256) {
    float t = 0.f;
    for (int r = 0; r < rpb; ++r) t += part[r * 2 * C + i];
    atomicAdd(st + i, (double)t);
  }
}

// apply: dx = gamma*rstd*(g - s1/n - xhat*s2/n); block 0 also emits dgamma += s2, dbeta += s1
__global__ __launch_bounds__(256) void bn_relu_bwd_apply_kernel(const bf16_t* __restrict__ dy, int ld_dy,
                                                                const bf16_t* __restrict__ x, const float* __restrict__ scale,
                                                                const float* __restrict__ shift, const float* __restrict__ mean,
                                                                const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                                const double* __restrict__ stats, float* __restrict__ dgamma,
                                                                float* __restrict__ dbeta, bf16_t* __restrict__ dx,
                                                                long long M, int C, int nslots, double count, DropCfg drop) {
  __shared__ float s1s[512], s2s[512];
  for (int c = threadIdx.x; c < C; c += 256) {
    double a = 0., b = 0.;
    for (int k = 0; k < nslots; ++k) {
      a += stats[(size_t)k * 2 * C + c];
      b += stats[(size_t)k * 2 * C + C + c];
    }
    s1s[c] = (float)(a / count);
    s2s[c] = (float)(b / count);
    if (blockIdx.x == 0) {
      dgamma[c] += (float)b;
      dbeta[c] += (float)a;
    }
  }
  __syncthreads();
  // a thread keeps ONE group of eight channels (as the reduce kernel): its per-channel coefficients live in registers,
  //   dx = A * [a > 0] * g + P * x + Q,   A = gamma * rstd,  P = -A * rstd * s2,  Q = -A * (s1 - rstd * mean * s2),  a = x * scale + shift
  // (per element the former loop fetched scale / shift / mean / rstd / gamma from memory: 40 scalar loads per 16-byte output)
  const int cv = C >> 3;
  const int c8 = threadIdx.x % cv, rl = threadIdx.x / cv, rpb = 256 / cv;
  float sc[8], sh[8], cA[8], cP[8], cQ[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int c = c8 * 8 + j;
    const float r = rstd[c], a = gamma[c] * r;
    sc[j] = scale[c], sh[j] = shift[c];
    cA[j] = a;
    cP[j] = -a * r * s2s[c];
    cQ[j] = -a * (s1s[c] - r * mean[c] * s2s[c]);
  }
  if (rl < rpb) {
    for (long long m = (long long)blockIdx.x * rpb + rl; m < M; m += (long long)gridDim.x * rpb) {
      float g[8], xv[8];
      unpack8(*(const uint4*)(dy + (size_t)m * ld_dy + c8 * 8), g);
      unpack8(*(const uint4*)(x + (size_t)m * C + c8 * 8), xv);
      drop8(drop, (unsigned long long)m * C + c8 * 8, g);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float gg = (xv[j] * sc[j] + sh[j] > 0.f) ? g[j] : 0.f;
        g[j] = cA[j] * gg + (cP[j] * xv[j] + cQ[j]);
      }
      *(uint4*)(dx + (size_t)m * C + c8 * 8) = pack8f(g);
    }
  }
}

// ------------------------------------------------------------------ dst[c][r] = src[r][c]   (64x64 LDS tiles)
__global__ __launch_bounds__(256) void transpose_kernel(const bf16_t* __restrict__ src, bf16_t* __restrict__ dst, int R,
                                                        int Cc, int ld_src, long long ld_dst) {
  __shared__ bf16_t tile[64][66];
  const int r0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
  for (int i = threadIdx.x; i < 64 * 64; i += 256) {
    const int r = i >> 6, c = i & 63;
    tile[r][c] = (r0 + r < R && c0 + c < Cc) ? src[(size_t)(r0 + r) * ld_src + c0 + c] : (bf16_t)0;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 64 * 64; i += 256) {
    const int c = i >> 6, r = i & 63;
    if (r0 + r < R && c0 + c < Cc) dst[(size_t)(c0 + c) * ld_dst + r0 + r] = tile[r][c];
  }
}

inline DropCfg make_drop(float p, unsigned long long seed) {
  DropCfg d;
  d.seed = seed;
  d.thresh = p > 0.f ? (unsigned)(p * 65536.f + 0.5f) : 0u;
  d.inv_keep = p > 0.f ? 1.f / (1.f - (float)d.thresh / 65536.f) : 1.f;
  return d;
}

inline int nblk(long long work, int per, int cap = 16384) {
  long long b = (work + per - 1) / per;
  return (int)(b < 1 ? 1 : (b > cap ? cap : b));
}

// ------------------------------------------------------------------ conv weight re-pack (once per optimizer step)
// W [Cout, Cin, 3, 3] f32 (nn.Conv2d layout)  ->  wk [Cout, 9*Cp] bf16, k = (ky*3+kx)*Cp + c   (B operand of the forward
// implicit GEMM) and wd [Cp, 9*Cout] bf16, k = (ky*3+kx)*Cout + co (B operand of the dgrad GEMM).  Channel c of the
// packed layout is source channel (c + rot) mod Cin (fus3 keeps its concat buffer as [up | image]); c >= Cin is zero pad.
// Up to 8 weights per launch (blockIdx.y = weight, blockIdx.z = 0: wk / 1: wd): the seven convolutions of the decoder are re-packed
// every step, and a dependent 6 us launch each is what they cost.
struct PackBatch { mvit_conv_pack_desc d[8]; };
__global__ __launch_bounds__(256) void pack_conv_w_kernel(const PackBatch pb) {
  const mvit_conv_pack_desc& q = pb.d[blockIdx.y];
  const float* __restrict__ W = q.W;
  bf16_t* __restrict__ wk = (bf16_t*)q.wk;
  bf16_t* __restrict__ wd = (bf16_t*)q.wd;
  const int Cout = q.Cout, Cin = q.Cin, Cp = q.Cp, rot = q.rot;
  if (blockIdx.z == 1 && !wd) return;
  const int total = Cout * Cp;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
    int co, c;
    if (blockIdx.z == 0) {  // c fastest: coalesced wk stores
      co = i / Cp, c = i - co * Cp;
    } else {                // co fastest: coalesced wd stores
      c = i / Cout, co = i - c * Cout;
    }
    float v[9];
    if (c < Cin) {
      const float* src = W + ((size_t)co * Cin + (c + rot) % Cin) * 9;
#pragma unroll
      for (int t = 0; t < 9; ++t) v[t] = src[t];
    } else {
#pragma unroll
      for (int t = 0; t < 9; ++t) v[t] = 0.f;
    }
    if (blockIdx.z == 0) {
#pragma unroll
      for (int t = 0; t < 9; ++t) wk[(size_t)co * 9 * Cp + (size_t)t * Cp + c] = f2bf(v[t]);
    } else {
#pragma unroll
      for (int t = 0; t < 9; ++t) wd[(size_t)c * 9 * Cout + (size_t)t * Cout + co] = f2bf(v[t]);
    }
  }
}

// ------------------------------------------------------------------ ConvTranspose2d(kernel 2, stride 2) pixel shuffle
// The transposed convolution is a GEMM [B*H*W, Cin] x [4*Cout, Cin]^T whose column (dy*2+dx)*C + c belongs to output
// pixel (2y+dy, 2x+dx), channel c (reference: nn.ConvTranspose2d in Deconv2DBlock / Decoder, src/generators/unet.py:304-372,
// 490-498).  forward: packed [M, 4C] -> NHWC [B, 2H, 2W, ld_dst] (a channel slice of a concat buffer);
// inverse: the adjoint gather of the output gradient back into the packed layout.
__global__ __launch_bounds__(256) void pixel_shuffle2x_kernel(bf16_t* __restrict__ packed, bf16_t* __restrict__ img, int B, int H,
                                                              int W, int C, long long ld_img, int inverse) {
  const int c8n = C / 8;
  const long long total = (long long)B * H * W * 4 * c8n;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int c8 = (int)(i % c8n);
    long long t = i / c8n;
    const int q = (int)(t & 3);
    const long long m = t >> 2;
    const int x = (int)(m % W);
    const long long t2 = m / W;
    const int y = (int)(t2 % H), b = (int)(t2 / H);
    const long long pix = ((long long)b * 2 * H + 2 * y + (q >> 1)) * (2 * W) + 2 * x + (q & 1);
    uint4* ps = (uint4*)(packed + (size_t)m * 4 * C + (size_t)q * C + c8 * 8);
    uint4* pi = (uint4*)(img + (size_t)pix * ld_img + c8 * 8);
    if (inverse)
      *ps = *pi;
    else
      *pi = *ps;
  }
}

}  // namespace

extern "C" {

MVIT_API int mvit_resample2d(const void* src, void* dst, const int* ty_idx, const float* ty_w, const int* tx_idx,
                             const float* tx_w, const float* scale, const float* shift, int B, int h, int w, int H, int W,
                             int C, int ld_src, int ld_dst, long long src_bstride, long long dst_bstride, int T,
                             mvit_stream_t stream) {
  MVIT_CLEAR_ERROR();
  if (B <= 0 || C <= 0 || (C & 7) || (ld_src & 7) || (ld_dst & 7) || T <= 0 || T > 16) return MVIT_EINVAL;
  if ((scale == nullptr) != (shift == nullptr)) return MVIT_EINVAL;
  ResampleArgs a{(const bf16_t*)src, (bf16_t*)dst, ty_idx, ty_w, tx_idx, tx_w, scale, shift, B, h, w, H, W, C,
                 ld_src, ld_dst, T, src_bstride, dst_bstride};
  if (H > 65535 || B > 65535) return MVIT_EINVAL;
  const dim3 grid((unsigned)((W * (C >> 3) + 255) / 256), (unsigned)H, (unsigned)B);
  hipStream_t s = (hipStream_t)stream;
  switch (T) {
    case 1: hipLaunchKernelGGL(resample_kernel<1>, grid, dim3(256), 0, s, a); break;
    case 2: hipLaunchKernelGGL(resample_kernel<2>, grid, dim3(256), 0, s, a); break;
    case 3: hipLaunchKernelGGL(resample_kernel<3>, grid, dim3(256), 0, s, a); break;
    case 4: hipLaunchKernelGGL(resample_kernel<4>, grid, dim3(256), 0, s, a); break;
    default: hipLaunchKernelGGL(resample_kernel<0>, grid, dim3(256), 0, s, a); break;
  }
  return MVIT_LAUNCH_CHECK();
}

MVIT_API int mvit_upsample2x_bilinear(const void* src, void* dst, const float* scale, const float* shift, const void* extra8,
                                      int B, int h, int w, int C, int ld_src, int ld_dst, long long src_bstride,
                                      long long dst_bstride, mvit_stream_t stream) {
  MVIT_CLEAR_ERROR();
  if (B <= 0 || h <= 0 || w <= 0 || C <= 0 || (C & 7) || (ld_src & 7) || (ld_dst & 7) || ld_src < C) return MVIT_EINVAL;
  if (ld_dst < C + (extra8 ? 8 : 0) || (scale == nullptr) != (shift == nullptr)) return MVIT_EINVAL;
  Up2Args a{(const bf16_t*)src, (bf16_t*)dst, scale, shift, (const bf16_t*)extra8, B, h, w, C, ld_src, ld_dst, src_bstride,
            dst_bstride};
  hipLaunchKernelGGL(upsample2x_kernel, dim3(nblk((long long)B * (h + 1) * (w + 1) * (C >> 3), 256)), dim3(256), 0,
                     (hipStream_t)stream, a);
  return MVIT_LAUNCH_CHECK();
}

MVIT_API int mvit_upsample2x_bilinear_bwd(const void* d_out, void* d_in, int B, int h, int w, int C, int ld_dout, int ld_din,
                                          long long dout_bstride, long long din_bstride, mvit_stream_t stream) {
  MVIT_CLEAR_ERROR();
  if (B <= 0 || h <= 0 || w <= 0 || C <= 0 || (C & 7) || (ld_dout & 7) || (ld_din & 7) || ld_dout < C || ld_din < C) return MVIT_EINVAL;
  Up2Args a{(const bf16_t*)d_out, (bf16_t*)d_in, nullptr, nullptr, nullptr, B, h, w, C, ld_dout, ld_din, dout_bstride, din_bstride};
  hipLaunchKernelGGL(upsample2x_adjoint_kernel, dim3(nblk((long long)B * h * w * (C >> 3), 256)), dim3(256), 0,
                     (hipStream_t)stream, a);
  return MVIT_LAUNCH_CHECK();
}

MVIT_API int mvit_image_to_nhwc(const float* img, void* dst, int B, int S, int C, int ld_dst, int nzero,
                                mvit_stream_t stream) {
  MVIT_CLEAR_ERROR();
  if (B <= 0 || S <= 0 || C <= 0 || C + nzero > ld_dst) return MVIT_EINVAL;
  hipLaunchKernelGGL(image_to_nhwc_kernel, dim3(nblk((long long)B * S * S, 256)), dim3(256), 0, (hipStream_t)stream, img,
                     (bf16_t*)dst, B, S, C, ld_dst, nzero);
  return MVIT_LAUNCH_CHECK();
}

MVIT_API int mvit_bn_finalize(const double* stats, const float* gamma, const float* beta, float* running_mean,
                              float* running_var, float* scale, float* shift, float* mean_out, float* rstd_out, int C,
                              int nslots, double count, float eps, float momentum, int training, mvit_stream_t stream) {
  MVIT_CLEAR_ERROR();
  if (C <= 0 || (training && (!stats || nslots <= 0 || count <= 0))) return MVIT_EINVAL;
  hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + 7) / 8), dim3(256), 0, (hipStream_t)stream, stats, gamma, beta,
                     running_mean, running_var, scale, shift, mean_out, rstd_out, C, nslots, count, eps, momentum, training);
  return MVIT_LAUNCH_CHECK();
}

MVIT_API int mvit_bn_relu_apply(const void* x, const float* scale, const float* shift, void* out, long long M, int C,
                                int ld_x, int ld_out, float drop_p, unsigned long long drop_seed, mvit_stream_t stream) {
  MVIT_CLEAR_ERROR();
  if (M <= 0 || C <= 0 || (C & 7) || (ld_x & 7) || (ld_out & 7) || !(drop_p >= 0.f && drop_p < 1.f)) return MVIT_EINVAL;
  hipLaunchKernelGGL(bn_relu_apply_kernel, dim3(nblk(M * (C >> 3), 256)), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)x, scale, shift, (bf16_t*)out, M, C, ld_x, ld_out, make_drop(drop_p, drop_seed));
  return MVIT_LAUNCH_CHECK();
}

MVIT_API int mvit_bn_relu_bwd_reduce(const void* dy, int ld_dy, const void* x, const float* scale, const float* shift,
                                     const float* mean, const float* rstd, double* stats, long long M, int C, int nslots,
                                     float drop_p, unsigned long long drop_seed, mvit_stream_t stream) {
  MVIT_CLEAR_ERROR();
  if (M <= 0 || C <= 0 || (C & 7) || C > 512 || (ld_dy & 7) || nslots <= 0 || !(drop_p >= 0.f && drop_p < 1.f)) return MVIT_EINVAL;
  const int rpb = 256 / (C >> 3);
  // (>= 256 slots: the grid is capped at the slot count, one writer block per slot -> run-to-run identical statistics)
  hipLaunchKernelGGL(bn_relu_bwd_reduce_kernel, dim3(nblk(M, rpb * 16, nslots >= 256 ? nslots : 2048)), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)dy, ld_dy, (const bf16_t*)x, scale, shift, mean, rstd, stats, M, C, nslots,
                     make_drop(drop_p, drop_seed));
  return MVIT_LAUNCH_CHECK();
}

MVIT_API int mvit_bn_relu_bwd_apply(const void* dy, int ld_dy, const void* x, const float* scale, const float* shift,
                                    const float* mean, const float* rstd, const float* gamma, const double* stats,
                                    float* dgamma, float* dbeta, void* dx, long long M, int C, int nslots, double count,
                                    float drop_p, unsigned long long drop_seed, mvit_stream_t stream) {
  MVIT_CLEAR_ERROR();
  if (M <= 0 || C <= 0 || (C & 7) || C > 512 || (ld_dy & 7) || nslots <= 0 || !(drop_p >= 0.f && drop_p < 1.f)) return MVIT_EINVAL;
  // (every block re-sums the statistic slots in its prologue: few, long-lived blocks)
  hipLaunchKernelGGL(bn_relu_bwd_apply_kernel, dim3(nblk(M, (256 / (C >> 3)) * 8, 1024)), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)dy, ld_dy, (const bf16_t*)x, scale, shift, mean, rstd, gamma, stats, dgamma, dbeta,
                     (bf16_t*)dx, M, C, nslots, count, make_drop(drop_p, drop_seed));
  return MVIT_LAUNCH_CHECK();
}

MVIT_API int mvit_transpose_bf16(const void* src, void* dst, int R, int Cc, int ld_src, long long ld_dst,
                                 mvit_stream_t stream) {
  MVIT_CLEAR_ERROR();
  if (R <= 0 || Cc <= 0) return MVIT_EINVAL;
  hipLaunchKernelGGL(transpose_kernel, dim3((R + 63) / 64, (Cc + 63) / 64), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)src, (bf16_t*)dst, R, Cc, ld_src, ld_dst);
  return MVIT_LAUNCH_CHECK();
}

MVIT_API int mvit_pack_conv3x3_weights(const float* W, void* wk, void* wd, int Cout, int Cin, int Cp, int rot,
                                       mvit_stream_t stream) {
  MVIT_CLEAR_ERROR();
  if (Cout <= 0 || Cin <= 0 || Cp < Cin || (Cp & 7) || rot < 0 || !wk) return MVIT_EINVAL;
  mvit_conv_pack_desc d{W, wk, wd, Cout, Cin, Cp, rot};
  return mvit_pack_conv3x3_weights_multi(&d, 1, stream);
}

MVIT_API int mvit_pack_conv3x3_weights_multi(const mvit_conv_pack_desc* descs, int n, mvit_stream_t stream) {
  MVIT_CLEAR_ERROR();
  if (!descs || n <= 0 || n > 8) return MVIT_EINVAL;
  PackBatch pb{};
  int most = 0, any_wd = 0;
  for (int i = 0; i < n; ++i) {
    const mvit_conv_pack_desc& q = descs[i];
    if (q.Cout <= 0 || q.Cin <= 0 || q.Cp < q.Cin || (q.Cp & 7) || q.rot < 0 || !q.wk || !q.W) return MVIT_EINVAL;
    pb.d[i] = q;
    most = max(most, q.Cout * q.Cp);
    any_wd |= q.wd != nullptr;
  }
  hipLaunchKernelGGL(pack_conv_w_kernel, dim3(min((most + 255) / 256, 1024), n, any_wd ? 2 : 1), dim3(256), 0, (hipStream_t)stream, pb);
  return MVIT_LAUNCH_CHECK();
}

MVIT_API int mvit_pixel_shuffle2x(void* packed, void* img, int B, int H, int W, int C, long long ld_img, int inverse,
                                  mvit_stream_t stream) {
  MVIT_CLEAR_ERROR();
  if (B <= 0 || H <= 0 || W <= 0 || C <= 0 || (C & 7) || (ld_img & 7) || ld_img < C) return MVIT_EINVAL;
  const long long total = (long long)B * H * W * 4 * (C / 8);
  hipLaunchKernelGGL(pixel_shuffle2x_kernel, dim3((unsigned)min((total + 255) / 256, (long long)8192)), dim3(256), 0,
                     (hipStream_t)stream, (bf16_t*)packed, (bf16_t*)img, B, H, W, C, ld_img, inverse);
  return MVIT_LAUNCH_CHECK();
}

}  // extern "C"
