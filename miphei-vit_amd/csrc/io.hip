// On-device input / output stage either side of the generator (HBM-bound byte work).
//   preprocess : uint8 RGB tiles [B,H,W,3] (as decoded from disk) -> normalised f32 NCHW (x - mean) / std
//                (NormalizationLayer mode "he", /root/reference/src/dataset.py:545-575, after ToTensor's HWC->CHW)
//   targets    : uint8 mIF [B,H,W,C] -> f32 NCHW  x/255*1.8 - 0.9   (mode "if", dataset.py:573)
//   export     : f32 NCHW predictions -> uint8 NCHW ((y+0.9)/1.8).clamp(0,1)*255 truncated
//                (SavePredictionsCallback.on_predict_batch_end, /root/reference/src/callbacks.py:345-346)
#include "common.hpp"
#include "../../include/miphei_hip.h"

namespace {

// one thread per 4 consecutive pixels of one output plane (coalesced 16-byte f32 stores)
__global__ __launch_bounds__(256) void u8_nhwc_to_f32_nchw_kernel(const uint8_t* __restrict__ src, float* __restrict__ dst,
                                                                  const float* __restrict__ a, const float* __restrict__ b,
                                                                  int B, int C, long long HW) {
  const long long total = (long long)B * C * HW / 4;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const long long e = i * 4;
    const long long pix = e % HW;
    const int c = (int)((e / HW) % C), bb = (int)(e / (HW * C));
    const uint8_t* s = src + ((size_t)bb * HW + pix) * C + c;
    const float sa = a[c], sb = b[c];
    float4 o;
    o.x = (float)s[0] * sa + sb;
    o.y = (float)s[(size_t)C] * sa + sb;
    o.z = (float)s[(size_t)2 * C] * sa + sb;
    o.w = (float)s[(size_t)3 * C] * sa + sb;
    *(float4*)(dst + e) = o;
  }
}

__global__ __launch_bounds__(256) void f32_to_u8_export_kernel(const float* __restrict__ src, uint8_t* __restrict__ dst,
                                                               long long n) {
  for (long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += (long long)gridDim.x * 1024) {
    const float4 v = *(const float4*)(src + i);
    const float f[4] = {v.x, v.y, v.z, v.w};
    uint32_t pk = 0;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float t = (f[e] + 0.9f) / 1.8f;
      t = fminf(fmaxf(t, 0.f), 1.f) * 255.f;
      pk |= ((uint32_t)t & 0xffu) << (8 * e);  // float -> uint8 truncates, as torch .to(torch.uint8)
    }
    *(uint32_t*)(dst + i) = pk;
  }
}

inline int nblk(long long work, int per, int cap) {
  long long b = (work + per - 1) / per;
  return (int)(b < 1 ? 1 : (b > cap ? cap : b));
}

}  // namespace

extern "C" {

MVIT_API int mvit_u8_nhwc_to_f32_nchw(const void* src_u8, float* dst, const float* scale, const float* shift, int B, int C,
                                      long long HW, mvit_stream_t stream) {
  MVIT_CLEAR_ERROR();
  if (B <= 0 || C <= 0 || HW <= 0 || (HW & 3)) return MVIT_EINVAL;
  hipLaunchKernelGGL(u8_nhwc_to_f32_nchw_kernel, dim3(nblk((long long)B * C * HW / 4, 256, 8192)), dim3(256), 0,
                     (hipStream_t)stream, (const uint8_t*)src_u8, dst, scale, shift, B, C, HW);
  return MVIT_LAUNCH_CHECK();
}

MVIT_API int mvit_f32_to_u8_export(const float* src, void* dst_u8, long long n, mvit_stream_t stream) {
  MVIT_CLEAR_ERROR();
  if (n <= 0 || (n & 3)) return MVIT_EINVAL;
  hipLaunchKernelGGL(f32_to_u8_export_kernel, dim3(nblk(n, 1024, 8192)), dim3(256), 0, (hipStream_t)stream, src,
                     (uint8_t*)dst_u8, n);
  return MVIT_LAUNCH_CHECK();
}

}  // extern "C"
