// Loss and optimiser kernels of the training step (HBM-bound elementwise / reductions).
//   WeightedMSELoss forward+backward   src/loss.py:47-57 (weights built at src/train.py:137-142)
//   global-norm clip + Adam            src/models.py:136-138, configure_optimizers src/models.py:359-371
#include "common.hpp"
#include "../../include/miphei_hip.h"

namespace {

__device__ __forceinline__ double block_sum_d(double v, double* sh) {
  v = wave_sum_d(v);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) sh[wave] = v;
  __syncthreads();
  double r = 0.;
  if (threadIdx.x == 0)
    for (int i = 0; i < (int)(blockDim.x >> 6); ++i) r += sh[i];
  return r;  // valid on thread 0
}

// grid: (chunks, B*C).  loss_acc += w_c * sum (p-t)^2 ; dY = coef * w_c * (p-t)
__global__ __launch_bounds__(256) void wmse_kernel(const float* __restrict__ pred, const float* __restrict__ target,
                                                   const float* __restrict__ w, double* __restrict__ loss_acc,
                                                   float* __restrict__ dY, int C, long long HW, float coef) {
  __shared__ double sh[4];
  const int plane = blockIdx.y, c = plane % C;
  const float wc = w[c];
  const float* p = pred + (size_t)plane * HW;
  const float* t = target + (size_t)plane * HW;
  float* d = dY ? dY + (size_t)plane * HW : nullptr;
  double acc = 0.;
  for (long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * 4; i < HW; i += (long long)gridDim.x * 1024) {
    if (i + 3 < HW) {
      const float4 a = *(const float4*)(p + i), b = *(const float4*)(t + i);
      const float e0 = a.x - b.x, e1 = a.y - b.y, e2 = a.z - b.z, e3 = a.w - b.w;
      acc += (double)(e0 * e0 + e1 * e1 + e2 * e2 + e3 * e3);
      if (d) *(float4*)(d + i) = make_float4(coef * wc * e0, coef * wc * e1, coef * wc * e2, coef * wc * e3);
    } else {
      for (long long j = i; j < HW; ++j) {
        const float e = p[j] - t[j];
        acc += (double)(e * e);
        if (d) d[j] = coef * wc * e;
      }
    }
  }
  const double r = block_sum_d(acc, sh);
  if (threadIdx.x == 0) atomicAdd(loss_acc, r * (double)wc);
}

__global__ __launch_bounds__(256) void sqnorm_kernel(const float* __restrict__ x, double* __restrict__ out, long long n) {
  __shared__ double sh[4];
  double acc = 0.;
  // four unguarded 16-byte loads per trip while they all fit (a bounds test around each load makes hipcc wait for it before the
  // next one: 26 dependent round trips per thread at the 6.7 M trainable parameters), then single groups, then the ragged end
  const long long stride = (long long)gridDim.x * 1024;
  long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
  for (; i + 3 * stride + 3 < n; i += 4 * stride) {
    const float4 a = *(const float4*)(x + i), b = *(const float4*)(x + i + stride), c = *(const float4*)(x + i + 2 * stride),
                 d = *(const float4*)(x + i + 3 * stride);
    acc += (double)(a.x * a.x + a.y * a.y) + (double)(a.z * a.z + a.w * a.w);
    acc += (double)(b.x * b.x + b.y * b.y) + (double)(b.z * b.z + b.w * b.w);
    acc += (double)(c.x * c.x + c.y * c.y) + (double)(c.z * c.z + c.w * c.w);
    acc += (double)(d.x * d.x + d.y * d.y) + (double)(d.z * d.z + d.w * d.w);
  }
  for (; i + 3 < n; i += stride) {
    const float4 a = *(const float4*)(x + i);
    acc += (double)(a.x * a.x + a.y * a.y) + (double)(a.z * a.z + a.w * a.w);
  }
  if (i < n)
    for (long long j = i; j < n; ++j) acc += (double)x[j] * x[j];
  const double r = block_sum_d(acc, sh);
  if (threadIdx.x == 0) atomicAdd(out, r);
}

// ordered variant (deterministic mode): per-block partial sums, then ONE block adds them in index order
__global__ __launch_bounds__(256) void sqnorm_partial_kernel(const float* __restrict__ x, double* __restrict__ partial, long long n) {
  __shared__ double sh[4];
  double acc = 0.;
  // four unguarded 16-byte loads per trip while they all fit (a bounds test around each load makes hipcc wait for it before the
  // next one: 26 dependent round trips per thread at the 6.7 M trainable parameters), then single groups, then the ragged end
  const long long stride = (long long)gridDim.x * 1024;
  long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
  for (; i + 3 * stride + 3 < n; i += 4 * stride) {
    const float4 a = *(const float4*)(x + i), b = *(const float4*)(x + i + stride), c = *(const float4*)(x + i + 2 * stride),
                 d = *(const float4*)(x + i + 3 * stride);
    acc += (double)(a.x * a.x + a.y * a.y) + (double)(a.z * a.z + a.w * a.w);
    acc += (double)(b.x * b.x + b.y * b.y) + (double)(b.z * b.z + b.w * b.w);
    acc += (double)(c.x * c.x + c.y * c.y) + (double)(c.z * c.z + c.w * c.w);
    acc += (double)(d.x * d.x + d.y * d.y) + (double)(d.z * d.z + d.w * d.w);
  }
  for (; i + 3 < n; i += stride) {
    const float4 a = *(const float4*)(x + i);
    acc += (double)(a.x * a.x + a.y * a.y) + (double)(a.z * a.z + a.w * a.w);
  }
  if (i < n)
    for (long long j = i; j < n; ++j) acc += (double)x[j] * x[j];
  const double r = block_sum_d(acc, sh);
  if (threadIdx.x == 0) partial[blockIdx.x] = r;
}
__global__ __launch_bounds__(64) void sqnorm_finish_kernel(const double* __restrict__ partial, int nparts, double* __restrict__ out) {
  if (threadIdx.x == 0) {
    double t = 0.;
    for (int i = 0; i < nparts; ++i) t += partial[i];
    *out += t;
  }
}

__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, const double* __restrict__ sqnorm, long long n,
                                                   float lr, float b1, float b2, float eps, float bc1, float bc2,
                                                   float max_norm, int* __restrict__ nonfinite) {
  float coef = 1.f;
  if (sqnorm) {
    // NaN guard on the device (reference: `torch.isnan(fake).any()` before backward, src/models.py:102-105): a NaN/Inf anywhere
    // in the generator output makes the gradient norm non-finite.  Such a step -- and, through the sticky flag, every later
    // one -- leaves p, m and v untouched, so the weights the host dumps when it notices are the last finite ones.
    const double sq = *sqnorm;
    const bool bad = !(sq == sq) || sq > 1.7e308 || (nonfinite && *nonfinite);
    if (bad) {
      if (nonfinite && blockIdx.x == 0 && threadIdx.x == 0) *nonfinite = 1;
      return;
    }
    if (max_norm > 0.f) {
      const float tn = (float)sqrt(sq);
      coef = fminf(1.f, max_norm / (tn + 1e-6f));
    }
  }
  const float step = lr / bc1, isb2 = rsqrtf(bc2);
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const float gg = g[i] * coef;
    const float mm = b1 * m[i] + (1.f - b1) * gg;
    const float vv = b2 * v[i] + (1.f - b2) * gg * gg;
    m[i] = mm;
    v[i] = vv;
    p[i] -= step * mm / (sqrtf(vv) * isb2 + eps);
  }
}

inline int nblk(long long work, int per, int cap) {
  long long b = (work + per - 1) / per;
  return (int)(b < 1 ? 1 : (b > cap ? cap : b));
}

}  // namespace

extern "C" {

MVIT_API int mvit_wmse_fwd_bwd(const float* pred, const float* target, const float* w, double* loss_acc, float* dY, int B,
                               int C, long long HW, float lambda_factor, mvit_stream_t stream) {
  MVIT_CLEAR_ERROR();
  if (B <= 0 || C <= 0 || HW <= 0) return MVIT_EINVAL;
  // loss = lambda/(C*B*HW) * sum_c w_c sum (p-t)^2  (caller scales loss_acc); dY = 2*lambda/(C*B*HW) * w_c * (p-t)
  const float coef = 2.f * lambda_factor / ((float)C * (float)B * (float)HW);
  hipLaunchKernelGGL(wmse_kernel, dim3(nblk(HW, 1024 * 16, 16), B * C), dim3(256), 0, (hipStream_t)stream, pred, target, w,
                     loss_acc, dY, C, HW, coef);
  return MVIT_LAUNCH_CHECK();
}

MVIT_API int mvit_sqnorm(const float* x, double* out, long long n, mvit_stream_t stream) {
  MVIT_CLEAR_ERROR();
  if (n <= 0) return MVIT_EINVAL;
  hipLaunchKernelGGL(sqnorm_kernel, dim3(nblk(n, 1024 * 16, 256)), dim3(256), 0, (hipStream_t)stream, x, out, n);
  return MVIT_LAUNCH_CHECK();
}

MVIT_API int mvit_sqnorm_ordered(const float* x, double* out, double* scratch256, long long n, mvit_stream_t stream) {
  MVIT_CLEAR_ERROR();
  if (n <= 0 || !scratch256 || !out) return MVIT_EINVAL;
  const int nb = nblk(n, 1024 * 16, 256);
  hipLaunchKernelGGL(sqnorm_partial_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, x, scratch256, n);
  hipLaunchKernelGGL(sqnorm_finish_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (const double*)scratch256, nb, out);
  return MVIT_LAUNCH_CHECK();
}

MVIT_API int mvit_adam_clip_step(float* p, const float* g, float* m, float* v, const double* sqnorm, long long n, float lr,
                                 float beta1, float beta2, float eps, float bias_c1, float bias_c2, float max_norm,
                                 int* nonfinite_flag, mvit_stream_t stream) {
  MVIT_CLEAR_ERROR();
  if (n <= 0) return MVIT_EINVAL;
  hipLaunchKernelGGL(adam_kernel, dim3(nblk(n, 256 * 4, 4096)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, sqnorm, n, lr,
                     beta1, beta2, eps, bias_c1, bias_c2, max_norm, nonfinite_flag);
  return MVIT_LAUNCH_CHECK();
}

}  // extern "C"
