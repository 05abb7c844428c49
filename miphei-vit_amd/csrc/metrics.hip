// Per-step image metrics of ModelModule.training_step / evaluation_step (reference src/models.py:35-52,140-143):
// torchmetrics 1.6.2 PeakSignalNoiseRatio(data_range=(-0.9, 0.9)) and StructuralSimilarityIndexMeasure(data_range=(-0.9, 0.9))
// state updates, on the device, without materialising the five Gaussian-filtered maps.
//   PSNR : sum_squared_error += sum (clamp(p) - clamp(t))^2 ; total += numel
//   SSIM : 11x11 Gaussian window (sigma 1.5), c1 = (0.01 R)^2, c2 = (0.03 R)^2, R = hi - lo; torchmetrics reflect-pads by 5,
//          filters, and crops the padded border again, i.e. only windows that lie fully inside the image count:
//          similarity += sum_b mean_{c, 5<=y<H-5, 5<=x<W-5} ssim ; total += B
#include "common.hpp"
#include "../../include/miphei_hip.h"

namespace {

constexpr int KS = 11, PAD = 5, TS = 32, RS = TS + KS - 1;  // window, tile of outputs, tile of inputs
// partial sums go to slotted accumulators, one 128-byte line per slot: f64 atomics that share a cache line serialise
// (~10 ns each), and there is one per wave
constexpr int SLOT_STRIDE = 16, SSIM_SLOTS = 32, PSNR_SLOTS = 64;

__global__ __launch_bounds__(256) void psnr_kernel(const float* __restrict__ p, const float* __restrict__ t,
                                                   double* __restrict__ slots, long long n, float lo, float hi) {
  double tot = 0.;
  for (long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += (long long)gridDim.x * 1024) {
    if (i + 3 < n) {
      const float4 a = *(const float4*)(p + i), b = *(const float4*)(t + i);
      const float d0 = fminf(fmaxf(a.x, lo), hi) - fminf(fmaxf(b.x, lo), hi), d1 = fminf(fmaxf(a.y, lo), hi) - fminf(fmaxf(b.y, lo), hi);
      const float d2 = fminf(fmaxf(a.z, lo), hi) - fminf(fmaxf(b.z, lo), hi), d3 = fminf(fmaxf(a.w, lo), hi) - fminf(fmaxf(b.w, lo), hi);
      tot += (double)(d0 * d0 + d1 * d1 + d2 * d2 + d3 * d3);
    } else {
      for (long long j = i; j < n; ++j) {
        const float d = fminf(fmaxf(p[j], lo), hi) - fminf(fmaxf(t[j], lo), hi);
        tot += (double)(d * d);
      }
    }
  }
  tot = wave_sum_d(tot);
  if ((threadIdx.x & 63) == 0) atomicAdd(slots + (size_t)((blockIdx.x * 4 + (threadIdx.x >> 6)) % PSNR_SLOTS) * SLOT_STRIDE, tot);
}

struct Gauss {
  float w[KS];
};

__global__ __launch_bounds__(256) void ssim_kernel(const float* __restrict__ p, const float* __restrict__ t,
                                                   double* __restrict__ per_image, int C, int H, int W, float lo, float hi,
                                                   float c1, float c2, Gauss gk) {
  __shared__ float ps[RS][RS + 1], ts[RS][RS + 1];
  __shared__ float hb[5][RS][TS];  // horizontally filtered p, t, p^2, t^2, p*t
  const int plane = blockIdx.z, b = plane / C;
  const int oy0 = blockIdx.y * TS, ox0 = blockIdx.x * TS;  // output tile origin (= input origin: output (y,x) uses rows y..y+10)
  const int OH = H - 2 * PAD, OW = W - 2 * PAD;
  const float* pp = p + (size_t)plane * H * W;
  const float* tp = t + (size_t)plane * H * W;
  for (int e = threadIdx.x; e < RS * RS; e += 256) {
    const int r = e / RS, c = e - r * RS;
    const int y = oy0 + r, x = ox0 + c;
    float a = 0.f, q = 0.f;
    if (y < H && x < W) {
      a = fminf(fmaxf(pp[(size_t)y * W + x], lo), hi);
      q = fminf(fmaxf(tp[(size_t)y * W + x], lo), hi);
    }
    ps[r][c] = a, ts[r][c] = q;
  }
  __syncthreads();
  // horizontal pass: one thread per (row, 8 adjacent columns) - 18 window inputs feed 8 outputs of each map
  for (int e = threadIdx.x; e < RS * (TS / 8); e += 256) {
    const int r = e / (TS / 8), c0 = (e - r * (TS / 8)) * 8;
    float a[8 + KS - 1], q[8 + KS - 1];
#pragma unroll
    for (int k = 0; k < 8 + KS - 1; ++k) a[k] = ps[r][c0 + k], q[k] = ts[r][c0 + k];
#pragma unroll
    for (int o = 0; o < 8; ++o) {
      float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f, s4 = 0.f;
#pragma unroll
      for (int k = 0; k < KS; ++k) {
        const float w = gk.w[k], x = a[o + k], y = q[o + k];
        s0 += w * x, s1 += w * y, s2 += w * x * x, s3 += w * y * y, s4 += w * x * y;
      }
      hb[0][r][c0 + o] = s0, hb[1][r][c0 + o] = s1, hb[2][r][c0 + o] = s2, hb[3][r][c0 + o] = s3, hb[4][r][c0 + o] = s4;
    }
  }
  __syncthreads();
  // vertical pass: one thread per (column, 4 adjacent rows) - 14 rows of each map feed 4 outputs
  float acc = 0.f;
  {
    const int c = threadIdx.x & (TS - 1), r0 = (threadIdx.x >> 5) * 4;
    float m[5][4];
#pragma unroll
    for (int f = 0; f < 5; ++f) {
      float v[4 + KS - 1];
#pragma unroll
      for (int k = 0; k < 4 + KS - 1; ++k) v[k] = hb[f][r0 + k][c];
#pragma unroll
      for (int o = 0; o < 4; ++o) {
        float sacc = 0.f;
#pragma unroll
        for (int k = 0; k < KS; ++k) sacc += gk.w[k] * v[o + k];
        m[f][o] = sacc;
      }
    }
#pragma unroll
    for (int o = 0; o < 4; ++o) {
      if (oy0 + r0 + o >= OH || ox0 + c >= OW) continue;
      const float mpp = m[0][o] * m[0][o], mtt = m[1][o] * m[1][o], mpt = m[0][o] * m[1][o];
      const float spp = fmaxf(m[2][o] - mpp, 0.f), stt = fmaxf(m[3][o] - mtt, 0.f), spt = m[4][o] - mpt;
      acc += ((2.f * mpt + c1) * (2.f * spt + c2)) / ((mpp + mtt + c1) * (spp + stt + c2));
    }
  }
  const float s = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) {
    const int slot = (blockIdx.x + blockIdx.y * gridDim.x + (threadIdx.x >> 6)) % SSIM_SLOTS;
    atomicAdd(per_image + ((size_t)b * SSIM_SLOTS + slot) * SLOT_STRIDE, (double)s);
  }
}

__global__ __launch_bounds__(64) void metrics_finalize_kernel(double* __restrict__ scratch, double* __restrict__ state, int B,
                                                               double count, double numel) {
  double* psnr_slots = scratch + (size_t)B * SSIM_SLOTS * SLOT_STRIDE;
  double s = 0., e = 0.;
  for (int i = threadIdx.x; i < B * SSIM_SLOTS; i += 64) {
    s += scratch[(size_t)i * SLOT_STRIDE] / count;
    scratch[(size_t)i * SLOT_STRIDE] = 0.;  // ready for the next update
  }
  for (int i = threadIdx.x; i < PSNR_SLOTS; i += 64) {
    e += psnr_slots[(size_t)i * SLOT_STRIDE];
    psnr_slots[(size_t)i * SLOT_STRIDE] = 0.;
  }
  s = wave_sum_d(s);
  e = wave_sum_d(e);
  if (threadIdx.x == 0) {
    state[0] += e;
    state[1] += numel;
    state[2] += s;
    state[3] += (double)B;
  }
}

}  // namespace

extern "C" MVIT_API long long mvit_pix_metrics_scratch_bytes(int B) {
  return (long long)(((size_t)B * SSIM_SLOTS + PSNR_SLOTS) * SLOT_STRIDE * sizeof(double));
}

extern "C" MVIT_API int mvit_pix_metrics_update(const float* pred, const float* target, double* state, double* per_image,
                                                long long scratch_bytes, int B, int C, int H, int W, float lo, float hi,
                                                mvit_stream_t stream) {
  MVIT_CLEAR_ERROR();
  if (B <= 0 || C <= 0 || H <= 2 * PAD || W <= 2 * PAD || !(hi > lo) || !state || !per_image) return MVIT_EINVAL;
  if (scratch_bytes < mvit_pix_metrics_scratch_bytes(B)) return MVIT_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  const long long n = (long long)B * C * H * W;
  long long nb = (n + 1023) / 1024;
  hipLaunchKernelGGL(psnr_kernel, dim3((unsigned)(nb > 2048 ? 2048 : nb)), dim3(256), 0, s, pred, target,
                     per_image + (size_t)B * SSIM_SLOTS * SLOT_STRIDE, n, lo, hi);
  Gauss gk;
  double sum = 0.;
  for (int k = 0; k < KS; ++k) {
    const double d = (double)(k - PAD) / 1.5;
    gk.w[k] = (float)exp(-d * d / 2.);
    sum += gk.w[k];
  }
  for (int k = 0; k < KS; ++k) gk.w[k] = (float)(gk.w[k] / sum);
  const float R = hi - lo, c1 = (0.01f * R) * (0.01f * R), c2 = (0.03f * R) * (0.03f * R);
  const int OH = H - 2 * PAD, OW = W - 2 * PAD;
  if ((long long)B * C > 65535) return MVIT_EINVAL;
  hipLaunchKernelGGL(ssim_kernel, dim3((OW + TS - 1) / TS, (OH + TS - 1) / TS, B * C), dim3(256), 0, s, pred, target, per_image, C, H,
                     W, lo, hi, c1, c2, gk);
  hipLaunchKernelGGL(metrics_finalize_kernel, dim3(1), dim3(64), 0, s, per_image, state, B, (double)C * OH * OW, (double)n);
  return MVIT_LAUNCH_CHECK();
}
