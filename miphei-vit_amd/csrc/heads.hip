// Fused per-marker output heads (16 x SegmentationHead on the shared 32-channel feature map).
//
// Reference: AttentionBlock / SegmentationHead, src/generators/unet.py:407-438, instantiated once per
// marker at src/generators/mipheivit.py:198-205 and run at :213-218 (9 op launches per head, the
// 32-channel map re-read 16 times).  Here the map is read once per stage:
//   moments   : sum x and sum x x^T over all pixels  -> the train-mode BatchNorm statistics of every
//               head's 1x1 conv output follow analytically (mean = w.mu + b, var = w^T Cov w)
//   gate      : g[p,h] = sigmoid(w2 . relu(BN(W1 x + b1)) + b2) for all heads, one thread per pixel
//   conv      : y[p,h] = tanh(b3 + sum_{3x3} W3[h] . (x*g_h)), one thread per pixel, NCHW f32 output
// and the matching backward passes.  VALU kernels with LDS-broadcast weights; HBM traffic is the
// 32-channel map plus the 16-channel gate / output.
#include "common.hpp"
#include "../../include/miphei_hip.h"

namespace {

constexpr int XC = 32;   // feature channels
constexpr int HC = 16;   // hidden channels of the gate (F_int = 32 // 2)
constexpr int MAXH = 16; // heads
constexpr int NMOM = XC + XC * XC;
constexpr int DB3_SLOTS = 64; // db3 partial sums: [DB3_SLOTS][32] floats (one 128-byte line per slot)

__device__ __forceinline__ void load_x32(const bf16_t* __restrict__ p, float (&x)[XC]) {
#pragma unroll
  for (int v = 0; v < 4; ++v) {
    const uint4 t = ((const uint4*)p)[v];
    const uint32_t u[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      x[v * 8 + 2 * j] = __uint_as_float(u[j] << 16);
      x[v * 8 + 2 * j + 1] = __uint_as_float(u[j] & 0xffff0000u);
    }
  }
}
__device__ __forceinline__ void load_g16(const bf16_t* __restrict__ p, float (&g)[MAXH]) {
#pragma unroll
  for (int v = 0; v < 2; ++v) {
    const uint4 t = ((const uint4*)p)[v];
    const uint32_t u[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      g[v * 8 + 2 * j] = __uint_as_float(u[j] << 16);
      g[v * 8 + 2 * j + 1] = __uint_as_float(u[j] & 0xffff0000u);
    }
  }
}

// ------------------------------------------------------------------ moments: sum x (32), sum x x^T (32x32)
__global__ __launch_bounds__(256) void moments_kernel(const bf16_t* __restrict__ x, double* __restrict__ mom, long long M,
                                                      int nslots) {
  __shared__ float xs[64][XC + 1];
  const int tid = threadIdx.x, i = tid >> 3, j0 = (tid & 7) * 4;
  float acc[4] = {0.f, 0.f, 0.f, 0.f}, s = 0.f;
  for (long long r0 = (long long)blockIdx.x * 64; r0 < M; r0 += (long long)gridDim.x * 64) {
    __syncthreads();
    for (int e = tid; e < 64 * XC; e += 256) {
      const int r = e >> 5, c = e & 31;
      xs[r][c] = (r0 + r < M) ? bf2f(x[(size_t)(r0 + r) * XC + c]) : 0.f;
    }
    __syncthreads();
#pragma unroll 8
    for (int r = 0; r < 64; ++r) {
      const float a = xs[r][i];
#pragma unroll
      for (int q = 0; q < 4; ++q) acc[q] += a * xs[r][j0 + q];
      if (tid < XC) s += xs[r][tid];
    }
  }
  double* o = mom + (size_t)(blockIdx.x % nslots) * NMOM;
#pragma unroll
  for (int q = 0; q < 4; ++q) atomicAdd(o + XC + i * XC + j0 + q, (double)acc[q]);
  if (tid < XC) atomicAdd(o + tid, (double)s);
}

// ------------------------------------------------------------------ BN statistics of every gate channel from the moments
__global__ __launch_bounds__(256) void bn_from_moments_kernel(const double* __restrict__ mom, const float* __restrict__ W1,
                                                              const float* __restrict__ b1, const float* __restrict__ gamma,
                                                              const float* __restrict__ beta, float* __restrict__ rmean,
                                                              float* __restrict__ rvar, float* __restrict__ scale,
                                                              float* __restrict__ shift, float* __restrict__ mean_o,
                                                              float* __restrict__ rstd_o, double* __restrict__ mom_sum,
                                                              int NCH, int nslots, double count, float eps, float momentum,
                                                              int training) {
  __shared__ double ms[NMOM];
  const int ch = threadIdx.x;
  if (training) {
    for (int e = threadIdx.x; e < NMOM; e += 256) {
      double s = 0.;
      for (int k = 0; k < nslots; ++k) s += mom[(size_t)k * NMOM + e];
      ms[e] = s;
      if (mom_sum) mom_sum[e] = s;
    }
    __syncthreads();
  }
  if (ch >= NCH) return;
  double mean, var;
  if (training) {
    const float* w = W1 + (size_t)ch * XC;
    double wm = 0.;
    for (int k = 0; k < XC; ++k) wm += (double)w[k] * ms[k] / count;
    mean = wm + b1[ch];
    double v = 0.;
    for (int j = 0; j < XC; ++j) {
      double row = 0.;
      for (int k = 0; k < XC; ++k) row += (double)w[k] * (ms[XC + j * XC + k] / count - (ms[j] / count) * (ms[k] / count));
      v += (double)w[j] * row;
    }
    var = v < 0. ? 0. : v;
    const double unb = count > 1. ? var * count / (count - 1.) : var;
    rmean[ch] = (float)((1. - momentum) * rmean[ch] + momentum * mean);
    rvar[ch] = (float)((1. - momentum) * rvar[ch] + momentum * unb);
  } else {
    mean = rmean[ch];
    var = rvar[ch];
  }
  const double rstd = 1. / sqrt(var + (double)eps);
  const double sc = gamma[ch] * rstd;
  scale[ch] = (float)sc;
  shift[ch] = (float)(beta[ch] - mean * sc);
  if (mean_o) mean_o[ch] = (float)mean;
  if (rstd_o) rstd_o[ch] = (float)rstd;
}

// ------------------------------------------------------------------ gate forward (thread per pixel)
__global__ __launch_bounds__(256) void gate_fwd_kernel(const bf16_t* __restrict__ x, const float* __restrict__ W1,
                                                       const float* __restrict__ b1, const float* __restrict__ scale,
                                                       const float* __restrict__ shift, const float* __restrict__ W2,
                                                       const float* __restrict__ b2, bf16_t* __restrict__ G, long long M,
                                                       int NH) {
  __shared__ __attribute__((aligned(16))) float w1s[MAXH * HC * XC];
  __shared__ float b1s[MAXH * HC], w2s[MAXH * HC], b2s[MAXH];
  const int nch = NH * HC;
  for (int e = threadIdx.x; e < nch * XC; e += 256) w1s[e] = W1[e] * scale[e >> 5];  // BN folded into the 1x1 conv
  for (int e = threadIdx.x; e < nch; e += 256) {
    b1s[e] = b1[e] * scale[e] + shift[e];
    w2s[e] = W2[e];
  }
  if (threadIdx.x < NH) b2s[threadIdx.x] = b2[threadIdx.x];
  __syncthreads();
  for (long long p = (long long)blockIdx.x * 256 + threadIdx.x; p < M; p += (long long)gridDim.x * 256) {
    float xv[XC];
    load_x32(x + (size_t)p * XC, xv);
    float g[MAXH];
#pragma unroll
    for (int h = 0; h < MAXH; ++h) g[h] = 0.f;
    for (int h = 0; h < NH; ++h) {
      float psi = b2s[h];
#pragma unroll 4
      for (int c = 0; c < HC; ++c) {
        const float4* w = (const float4*)(w1s + (h * HC + c) * XC);
        float a = b1s[h * HC + c];
#pragma unroll
        for (int k = 0; k < XC / 4; ++k) {
          const float4 ww = w[k];
          a += ww.x * xv[4 * k] + ww.y * xv[4 * k + 1] + ww.z * xv[4 * k + 2] + ww.w * xv[4 * k + 3];
        }
        psi += w2s[h * HC + c] * fmaxf(a, 0.f);
      }
      const float gv = sigmoidf_(psi);
#pragma unroll
      for (int hh = 0; hh < MAXH; ++hh)
        if (hh == h) g[hh] = gv;
    }
    uint4 o[2];
    uint32_t* ou = (uint32_t*)o;
#pragma unroll
    for (int j = 0; j < 8; ++j) ou[j] = pack2bf(g[2 * j], g[2 * j + 1]);
    ((uint4*)(G + (size_t)p * MAXH))[0] = o[0];
    ((uint4*)(G + (size_t)p * MAXH))[1] = o[1];
  }
}

// ------------------------------------------------------------------ gated 3x3 conv + tanh (thread per pixel)
__global__ __launch_bounds__(256) void conv_fwd_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ G,
                                                       const float* __restrict__ W3, const float* __restrict__ b3,
                                                       float* __restrict__ out, int B, int H, int W, int NH) {
  __shared__ __attribute__((aligned(16))) float w3s[MAXH * 9 * XC];
  for (int e = threadIdx.x; e < NH * 9 * XC; e += 256) w3s[e] = W3[e];
  __syncthreads();
  const long long M = (long long)B * H * W;
  for (long long p = (long long)blockIdx.x * 256 + threadIdx.x; p < M; p += (long long)gridDim.x * 256) {
    const int px = (int)(p % W);
    const long long t = p / W;
    const int py = (int)(t % H), b = (int)(t / H);
    float y[MAXH];
#pragma unroll
    for (int h = 0; h < MAXH; ++h) y[h] = 0.f;
#pragma unroll 1
    for (int d = 0; d < 9; ++d) {
      const int qy = py + d / 3 - 1, qx = px + d % 3 - 1;
      if (qy < 0 || qy >= H || qx < 0 || qx >= W) continue;
      const size_t q = ((size_t)b * H + qy) * W + qx;
      float xv[XC], g[MAXH];
      load_x32(x + q * XC, xv);
      load_g16(G + q * MAXH, g);
#pragma unroll
      for (int h = 0; h < MAXH; ++h) {
        if (h < NH) {
          const float4* w = (const float4*)(w3s + (h * 9 + d) * XC);
          float a = 0.f;
#pragma unroll
          for (int k = 0; k < XC / 4; ++k) {
            const float4 ww = w[k];
            a += ww.x * xv[4 * k] + ww.y * xv[4 * k + 1] + ww.z * xv[4 * k + 2] + ww.w * xv[4 * k + 3];
          }
          y[h] += a * g[h];
        }
      }
    }
    const size_t plane = (size_t)H * W, pix = (size_t)py * W + px;
#pragma unroll
    for (int h = 0; h < MAXH; ++h)
      if (h < NH) out[((size_t)b * NH + h) * plane + pix] = tanhf(y[h] + b3[h]);
  }
}

// ------------------------------------------------------------------ backward of the gated conv (thread per pixel q)
// dz[p,h] = dY*(1-y^2);  ET[(h,d)][q] = g[q,h]*dz[q-d+1,h];  dG[q,h] = sum_c x_c v_hc;  dXc[q,c] = sum_h g_h v_hc
// with v_hc = sum_d dz[q-d+1,h] * W3[h][d][c].   db3[h] += sum_p dz[p,h].
__global__ __launch_bounds__(256) void conv_bwd_kernel(const float* __restrict__ dY, const float* __restrict__ Y,
                                                       const bf16_t* __restrict__ x, const bf16_t* __restrict__ G,
                                                       const float* __restrict__ W3, bf16_t* __restrict__ ET,
                                                       float* __restrict__ dG, float* __restrict__ dXc,
                                                       float* __restrict__ db3, int B, int H, int W, int NH) {
  __shared__ __attribute__((aligned(16))) float w3s[MAXH * 9 * XC];
  for (int e = threadIdx.x; e < NH * 9 * XC; e += 256) w3s[e] = W3[e];
  __syncthreads();
  const long long M = (long long)B * H * W;
  const size_t plane = (size_t)H * W;
  float db_acc[MAXH];
#pragma unroll
  for (int h = 0; h < MAXH; ++h) db_acc[h] = 0.f;
  for (long long q0 = (long long)blockIdx.x * 256; q0 < M; q0 += (long long)gridDim.x * 256) {
    const long long q = q0 + threadIdx.x;
    const bool live = q < M;
    const long long qq = live ? q : 0;
    const int qx = (int)(qq % W);
    const long long t = qq / W;
    const int qy = (int)(t % H), b = (int)(t / H);
    float xv[XC], g[MAXH], dx[XC];
    load_x32(x + (size_t)qq * XC, xv);
    load_g16(G + (size_t)qq * MAXH, g);
#pragma unroll
    for (int c = 0; c < XC; ++c) dx[c] = 0.f;
    for (int h = 0; h < NH; ++h) {
      float gh = 0.f;
#pragma unroll
      for (int hh = 0; hh < MAXH; ++hh)
        if (hh == h) gh = g[hh];
      float v[XC];
#pragma unroll
      for (int c = 0; c < XC; ++c) v[c] = 0.f;
      float dzc = 0.f;
      const float* dyp = dY + ((size_t)b * NH + h) * plane;
      const float* yp = Y + ((size_t)b * NH + h) * plane;
#pragma unroll 1
      for (int d = 0; d < 9; ++d) {
        const int py = qy - (d / 3) + 1, px = qx - (d % 3) + 1;
        float dz = 0.f;
        if (live && py >= 0 && py < H && px >= 0 && px < W) {
          const size_t pi = (size_t)py * W + px;
          const float yy = yp[pi];
          dz = dyp[pi] * (1.f - yy * yy);
        }
        if (d == 4) dzc = dz;
        if (live) ET[((size_t)h * 9 + d) * M + q] = f2bf(gh * dz);
        const float4* w = (const float4*)(w3s + (h * 9 + d) * XC);
#pragma unroll
        for (int k = 0; k < XC / 4; ++k) {
          const float4 ww = w[k];
          v[4 * k] += dz * ww.x;
          v[4 * k + 1] += dz * ww.y;
          v[4 * k + 2] += dz * ww.z;
          v[4 * k + 3] += dz * ww.w;
        }
      }
      float dg = 0.f;
#pragma unroll
      for (int c = 0; c < XC; ++c) {
        dg += xv[c] * v[c];
        dx[c] += gh * v[c];
      }
      if (live) dG[(size_t)q * MAXH + h] = dg;
#pragma unroll
      for (int hh = 0; hh < MAXH; ++hh)
        if (hh == h) db_acc[hh] += dzc;
    }
    if (live) {
#pragma unroll
      for (int k = 0; k < XC / 4; ++k)
        ((float4*)(dXc + (size_t)q * XC))[k] = make_float4(dx[4 * k], dx[4 * k + 1], dx[4 * k + 2], dx[4 * k + 3]);
    }
  }
  // db3[h] += sum_p dz[p,h]: one atomic per wave and head, spread over DB3_SLOTS cache lines (summed by the caller)
  const int slot = (blockIdx.x * 4 + (threadIdx.x >> 6)) % DB3_SLOTS;
#pragma unroll
  for (int h = 0; h < MAXH; ++h) {
    const float s = wave_sum(db_acc[h]);
    if ((threadIdx.x & 63) == 0 && h < NH) atomicAdd(db3 + slot * 32 + h, s);
  }
}

// ------------------------------------------------------------------ gate backward, reduction pass (thread per gate channel)
// per channel ch=(h,c): Sb = sum da, Sg = sum da*that, Z[k] = sum (da*gamma) x_k, dw2 = sum dpsi*r, (c==0) db2 = sum dpsi
constexpr int GR_ROWS = 32;
constexpr int GR_OUT = 4 + XC;  // Sb, Sg, dw2, db2, Z[32]
__global__ __launch_bounds__(256) void gate_bwd_reduce_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ G,
                                                              const float* __restrict__ dG, const float* __restrict__ W1,
                                                              const float* __restrict__ b1, const float* __restrict__ scale,
                                                              const float* __restrict__ shift, const float* __restrict__ mean,
                                                              const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                              const float* __restrict__ W2, double* __restrict__ red,
                                                              long long M, int NH, int nslots) {
  __shared__ __attribute__((aligned(16))) float xs[GR_ROWS][XC];
  __shared__ float dps[GR_ROWS][MAXH];
  const int ch = threadIdx.x, nch = NH * HC;
  const bool act = ch < nch;
  const int h = act ? ch / HC : 0;
  float w[XC];
#pragma unroll
  for (int k = 0; k < XC; ++k) w[k] = act ? W1[(size_t)ch * XC + k] : 0.f;
  const float bb = act ? b1[ch] : 0.f, sc = act ? scale[ch] : 0.f, sh = act ? shift[ch] : 0.f;
  const float mu = act ? mean[ch] : 0.f, rs = act ? rstd[ch] : 0.f, gm = act ? gamma[ch] : 0.f, w2 = act ? W2[ch] : 0.f;
  float Sb = 0.f, Sg = 0.f, dw2 = 0.f, db2 = 0.f, Z[XC];
#pragma unroll
  for (int k = 0; k < XC; ++k) Z[k] = 0.f;
  for (long long r0 = (long long)blockIdx.x * GR_ROWS; r0 < M; r0 += (long long)gridDim.x * GR_ROWS) {
    __syncthreads();
    for (int e = threadIdx.x; e < GR_ROWS * XC; e += 256) {
      const int r = e >> 5, c = e & 31;
      xs[r][c] = (r0 + r < M) ? bf2f(x[(size_t)(r0 + r) * XC + c]) : 0.f;
    }
    for (int e = threadIdx.x; e < GR_ROWS * MAXH; e += 256) {
      const int r = e >> 4, hh = e & 15;
      float v = 0.f;
      if (r0 + r < M && hh < NH) {
        const float gv = bf2f(G[(size_t)(r0 + r) * MAXH + hh]);
        v = dG[(size_t)(r0 + r) * MAXH + hh] * gv * (1.f - gv);
      }
      dps[r][hh] = v;
    }
    __syncthreads();
    if (act) {
#pragma unroll 2
      for (int r = 0; r < GR_ROWS; ++r) {
        const float4* xr = (const float4*)xs[r];
        float u = bb;
#pragma unroll
        for (int k = 0; k < XC / 4; ++k) {
          const float4 xx = xr[k];
          u += w[4 * k] * xx.x + w[4 * k + 1] * xx.y + w[4 * k + 2] * xx.z + w[4 * k + 3] * xx.w;
        }
        const float a = u * sc + sh;
        const float dpsi = dps[r][h];
        const float da = a > 0.f ? dpsi * w2 : 0.f;
        Sb += da;
        Sg += da * (u - mu) * rs;
        dw2 += dpsi * fmaxf(a, 0.f);
        db2 += dpsi;
        const float dt = da * gm;
#pragma unroll
        for (int k = 0; k < XC / 4; ++k) {
          const float4 xx = xr[k];
          Z[4 * k] += dt * xx.x;
          Z[4 * k + 1] += dt * xx.y;
          Z[4 * k + 2] += dt * xx.z;
          Z[4 * k + 3] += dt * xx.w;
        }
      }
    }
  }
  if (act) {
    double* o = red + ((size_t)(blockIdx.x % nslots) * nch + ch) * GR_OUT;
    atomicAdd(o + 0, (double)Sb);
    atomicAdd(o + 1, (double)Sg);
    atomicAdd(o + 2, (double)dw2);
    atomicAdd(o + 3, (double)db2);
#pragma unroll
    for (int k = 0; k < XC; ++k) atomicAdd(o + 4 + k, (double)Z[k]);
  }
}

// finalize: parameter gradients + the two per-channel coefficients of the BN backward
__global__ __launch_bounds__(256) void gate_bwd_finalize_kernel(const double* __restrict__ red, const double* __restrict__ mom_sum,
                                                                const float* __restrict__ W1, const float* __restrict__ b1,
                                                                const float* __restrict__ mean, const float* __restrict__ rstd,
                                                                const float* __restrict__ gamma, float* __restrict__ dW1,
                                                                float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                                float* __restrict__ dW2, float* __restrict__ db2,
                                                                float* __restrict__ coef, int NH, int nslots, double count) {
  const int ch = threadIdx.x, nch = NH * HC;
  if (ch >= nch) return;
  double s[GR_OUT];
  for (int k = 0; k < GR_OUT; ++k) {
    double a = 0.;
    for (int sl = 0; sl < nslots; ++sl) a += red[((size_t)sl * nch + ch) * GR_OUT + k];
    s[k] = a;
  }
  const double Sb = s[0], Sg = s[1], gm = gamma[ch], rs = rstd[ch], mu = mean[ch];
  const double c1 = gm * Sb / count, c2 = gm * Sg / count;
  coef[2 * ch] = (float)c1;
  coef[2 * ch + 1] = (float)c2;
  dgamma[ch] += (float)Sg;
  dbeta[ch] += (float)Sb;
  dW2[ch] += (float)s[2];
  if ((ch % HC) == 0) db2[ch / HC] += (float)s[3];
  const float* w = W1 + (size_t)ch * XC;
  for (int k = 0; k < XC; ++k) {
    double wm2 = 0.;
    for (int j = 0; j < XC; ++j) wm2 += (double)w[j] * mom_sum[XC + j * XC + k];
    const double tx = rs * (wm2 + ((double)b1[ch] - mu) * mom_sum[k]);  // sum_p that_p x_p[k]
    dW1[(size_t)ch * XC + k] += (float)(rs * (s[4 + k] - c1 * mom_sum[k] - c2 * tx));
  }
}

// apply pass (thread per pixel): dF3[p,k] = dXc[p,k] + sum_{h,c} du[h,c] W1[h,c,k]
__global__ __launch_bounds__(256) void gate_bwd_apply_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ G,
                                                             const float* __restrict__ dG, const float* __restrict__ dXc,
                                                             const float* __restrict__ W1, const float* __restrict__ b1,
                                                             const float* __restrict__ scale, const float* __restrict__ shift,
                                                             const float* __restrict__ mean, const float* __restrict__ rstd,
                                                             const float* __restrict__ gamma, const float* __restrict__ W2,
                                                             const float* __restrict__ coef, bf16_t* __restrict__ dF,
                                                             long long M, int NH) {
  __shared__ __attribute__((aligned(16))) float w1s[MAXH * HC * XC];
  __shared__ float ps[MAXH * HC][8];  // b1, scale, shift, mean, rstd, gamma, w2, pad
  __shared__ float cs[MAXH * HC][2];
  const int nch = NH * HC;
  for (int e = threadIdx.x; e < nch * XC; e += 256) w1s[e] = W1[e];
  for (int e = threadIdx.x; e < nch; e += 256) {
    ps[e][0] = b1[e], ps[e][1] = scale[e], ps[e][2] = shift[e], ps[e][3] = mean[e], ps[e][4] = rstd[e], ps[e][5] = gamma[e];
    ps[e][6] = W2[e];
    cs[e][0] = coef[2 * e], cs[e][1] = coef[2 * e + 1];
  }
  __syncthreads();
  for (long long p = (long long)blockIdx.x * 256 + threadIdx.x; p < M; p += (long long)gridDim.x * 256) {
    float xv[XC], g[MAXH], dx[XC];
    load_x32(x + (size_t)p * XC, xv);
    load_g16(G + (size_t)p * MAXH, g);
#pragma unroll
    for (int k = 0; k < XC / 4; ++k) {
      const float4 t = ((const float4*)(dXc + (size_t)p * XC))[k];
      dx[4 * k] = t.x, dx[4 * k + 1] = t.y, dx[4 * k + 2] = t.z, dx[4 * k + 3] = t.w;
    }
    for (int h = 0; h < NH; ++h) {
      float gh = 0.f;
#pragma unroll
      for (int hh = 0; hh < MAXH; ++hh)
        if (hh == h) gh = g[hh];
      const float dpsi = dG[(size_t)p * MAXH + h] * gh * (1.f - gh);
#pragma unroll 2
      for (int c = 0; c < HC; ++c) {
        const int ch = h * HC + c;
        const float4* w = (const float4*)(w1s + ch * XC);
        float u = ps[ch][0];
#pragma unroll
        for (int k = 0; k < XC / 4; ++k) {
          const float4 ww = w[k];
          u += ww.x * xv[4 * k] + ww.y * xv[4 * k + 1] + ww.z * xv[4 * k + 2] + ww.w * xv[4 * k + 3];
        }
        const float a = u * ps[ch][1] + ps[ch][2];
        const float dt = a > 0.f ? dpsi * ps[ch][6] * ps[ch][5] : 0.f;
        const float th = (u - ps[ch][3]) * ps[ch][4];
        const float du = ps[ch][4] * (dt - cs[ch][0] - th * cs[ch][1]);
#pragma unroll
        for (int k = 0; k < XC / 4; ++k) {
          const float4 ww = w[k];
          dx[4 * k] += du * ww.x;
          dx[4 * k + 1] += du * ww.y;
          dx[4 * k + 2] += du * ww.z;
          dx[4 * k + 3] += du * ww.w;
        }
      }
    }
    uint4 o[4];
    uint32_t* ou = (uint32_t*)o;
#pragma unroll
    for (int j = 0; j < 16; ++j) ou[j] = pack2bf(dx[2 * j], dx[2 * j + 1]);
#pragma unroll
    for (int v = 0; v < 4; ++v) ((uint4*)(dF + (size_t)p * XC))[v] = o[v];
  }
}

inline int nblk(long long work, int per, int cap) {
  long long b = (work + per - 1) / per;
  return (int)(b < 1 ? 1 : (b > cap ? cap : b));
}

}  // namespace

extern "C" {

MVIT_API int mvit_heads_moments(const void* x, double* mom, long long M, int nslots, mvit_stream_t stream) {
  MVIT_CLEAR_ERROR();
  if (M <= 0 || nslots <= 0) return MVIT_EINVAL;
  hipLaunchKernelGGL(moments_kernel, dim3(nblk(M, 64 * 8, 1024)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, mom, M,
                     nslots);
  return MVIT_LAUNCH_CHECK();
}

MVIT_API int mvit_heads_bn_from_moments(const double* mom, const float* W1, const float* b1, const float* gamma,
                                        const float* beta, float* running_mean, float* running_var, float* scale,
                                        float* shift, float* mean_out, float* rstd_out, double* mom_sum, int NH, int nslots,
                                        double count, float eps, float momentum, int training, mvit_stream_t stream) {
  MVIT_CLEAR_ERROR();
  if (NH <= 0 || NH > MAXH || (training && (!mom || nslots <= 0 || count <= 0))) return MVIT_EINVAL;
  hipLaunchKernelGGL(bn_from_moments_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, mom, W1, b1, gamma, beta,
                     running_mean, running_var, scale, shift, mean_out, rstd_out, mom_sum, NH * HC, nslots, count, eps,
                     momentum, training);
  return MVIT_LAUNCH_CHECK();
}

MVIT_API int mvit_heads_gate_fwd(const void* x, const float* W1, const float* b1, const float* scale, const float* shift,
                                 const float* W2, const float* b2, void* G, long long M, int NH, mvit_stream_t stream) {
  MVIT_CLEAR_ERROR();
  if (M <= 0 || NH <= 0 || NH > MAXH) return MVIT_EINVAL;
  hipLaunchKernelGGL(gate_fwd_kernel, dim3(nblk(M, 256, 4096)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, W1, b1,
                     scale, shift, W2, b2, (bf16_t*)G, M, NH);
  return MVIT_LAUNCH_CHECK();
}

MVIT_API int mvit_heads_conv_fwd(const void* x, const void* G, const float* W3, const float* b3, float* out, int B, int H,
                                 int W, int NH, mvit_stream_t stream) {
  MVIT_CLEAR_ERROR();
  if (B <= 0 || NH <= 0 || NH > MAXH) return MVIT_EINVAL;
  hipLaunchKernelGGL(conv_fwd_kernel, dim3(nblk((long long)B * H * W, 256, 8192)), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)x, (const bf16_t*)G, W3, b3, out, B, H, W, NH);
  return MVIT_LAUNCH_CHECK();
}

MVIT_API int mvit_heads_conv_bwd(const float* dY, const float* Y, const void* x, const void* G, const float* W3, void* ET,
                                 float* dG, float* dXc, float* db3, int B, int H, int W, int NH, mvit_stream_t stream) {
  MVIT_CLEAR_ERROR();
  if (B <= 0 || NH <= 0 || NH > MAXH) return MVIT_EINVAL;
  hipLaunchKernelGGL(conv_bwd_kernel, dim3(nblk((long long)B * H * W, 256, 2048)), dim3(256), 0, (hipStream_t)stream, dY, Y,
                     (const bf16_t*)x, (const bf16_t*)G, W3, (bf16_t*)ET, dG, dXc, db3, B, H, W, NH);
  return MVIT_LAUNCH_CHECK();
}

MVIT_API int mvit_heads_gate_bwd(const void* x, const void* G, const float* dG, const float* dXc, const float* W1,
                                 const float* b1, const float* scale, const float* shift, const float* mean,
                                 const float* rstd, const float* gamma, const float* W2, const double* mom_sum, double* red,
                                 float* coef, float* dW1, float* dgamma, float* dbeta, float* dW2, float* db2, void* dF,
                                 long long M, int NH, int nslots, double count, mvit_stream_t stream) {
  MVIT_CLEAR_ERROR();
  if (M <= 0 || NH <= 0 || NH > MAXH || nslots <= 0) return MVIT_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(gate_bwd_reduce_kernel, dim3(nblk(M, GR_ROWS * 16, 512)), dim3(256), 0, s, (const bf16_t*)x,
                     (const bf16_t*)G, dG, W1, b1, scale, shift, mean, rstd, gamma, W2, red, M, NH, nslots);
  hipLaunchKernelGGL(gate_bwd_finalize_kernel, dim3(1), dim3(256), 0, s, red, mom_sum, W1, b1, mean, rstd, gamma, dW1, dgamma,
                     dbeta, dW2, db2, coef, NH, nslots, count);
  hipLaunchKernelGGL(gate_bwd_apply_kernel, dim3(nblk(M, 256, 4096)), dim3(256), 0, s, (const bf16_t*)x, (const bf16_t*)G, dG,
                     dXc, W1, b1, scale, shift, mean, rstd, gamma, W2, coef, (bf16_t*)dF, M, NH);
  return MVIT_LAUNCH_CHECK();
}

}  // extern "C"
