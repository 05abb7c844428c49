// On-device input / output stage either side of the generator (HBM-bound byte work).
//   preprocess : uint8 RGB tiles [B,H,W,3] (as decoded from disk) -> normalised f32 NCHW (x - mean) / std
//                (NormalizationLayer mode "he", /root/reference/src/dataset.py:545-575, after ToTensor's HWC->CHW)
//   targets    : uint8 mIF [B,H,W,C] -> f32 NCHW  x/255*1.8 - 0.9   (mode "if", dataset.py:573)
//   augment    : RandomCrop + HorizontalFlip + VerticalFlip + CoarseDropout applied JOINTLY to the uint8 image and target tile
//                (A.Compose([...], additional_targets={'image_target': 'image'}), /root/reference/src/dataset.py:458-468), then
//                the two normalisations above with the reference's own f32 operation order, straight into the f32 NCHW batch
//                tensors and (optionally) the engine's bf16 NHWC image buffer.  Draws are counter-based (splitmix64 of
//                seed, sample index, draw index): the host recomputes them (io_stage.augment_params)
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

__device__ __forceinline__ uint4 pack8f(const float (&f)[8]) {
  return make_uint4(pack2bf(f[0], f[1]), pack2bf(f[2], f[3]), pack2bf(f[4], f[5]), pack2bf(f[6], f[7]));
}

struct AugGeom {
  int B, C, Hs, Ws, H, W;                // source tile size, crop size
  unsigned long long seed, sample0;      // draw stream: sample index = sample0 + b
  float p_hflip, p_vflip, p_drop, hole_frac;
  float mean[3], std[3];                 // NormalizationLayer("he"): (x - mean) / std
};

__host__ __device__ inline unsigned long long aug_u64(unsigned long long seed, unsigned long long sample, int k) {
  unsigned long long z = seed + 0x9E3779B97F4A7C15ull * (16ull * sample + (unsigned long long)k + 1ull);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
// uniform double in [0, 1) from the top 53 bits
__host__ __device__ inline double aug_unit(unsigned long long u) { return (double)(u >> 11) * (1.0 / 9007199254740992.0); }

struct AugDraw { int oy, ox, hflip, vflip, drop, y1, x1, hh, hw; };
__host__ __device__ inline AugDraw aug_draw(const AugGeom& g, int b) {
  const unsigned long long n = g.sample0 + (unsigned long long)b;
  AugDraw d;
  d.oy = (int)(aug_u64(g.seed, n, 0) % (unsigned long long)(g.Hs - g.H + 1));     // A.RandomCrop
  d.ox = (int)(aug_u64(g.seed, n, 1) % (unsigned long long)(g.Ws - g.W + 1));
  d.hflip = aug_unit(aug_u64(g.seed, n, 2)) < (double)g.p_hflip;                  // A.HorizontalFlip(p)
  d.vflip = aug_unit(aug_u64(g.seed, n, 3)) < (double)g.p_vflip;                  // A.VerticalFlip(p)
  d.drop = aug_unit(aug_u64(g.seed, n, 4)) < (double)g.p_drop;                    // A.CoarseDropout(p), one hole, fill 0
  const int mh = (int)((double)g.hole_frac * g.H), mw = (int)((double)g.hole_frac * g.W);
  d.hh = (int)(aug_u64(g.seed, n, 5) % (unsigned long long)(mh + 1));             // hole size: integers in [0, frac * size]
  d.hw = (int)(aug_u64(g.seed, n, 6) % (unsigned long long)(mw + 1));
  d.y1 = (int)(aug_u64(g.seed, n, 7) % (unsigned long long)(g.H - d.hh + 1));     // hole position inside the (flipped) crop
  d.x1 = (int)(aug_u64(g.seed, n, 8) % (unsigned long long)(g.W - d.hw + 1));
  return d;
}

// One thread per 4 consecutive output pixels of one row: gathers the (cropped, flipped) uint8 pixels, zeroes the dropout hole,
// applies the reference's f32 arithmetic op by op (no FMA contraction: results are bit-identical to the numpy expressions) and
// writes 16-byte pieces of every output plane; the bf16 NHWC image buffer gets one 16-byte store per pixel.
#pragma clang fp contract(off)
__global__ __launch_bounds__(256) void augment_kernel(const uint8_t* __restrict__ img, const uint8_t* __restrict__ tgt,
                                                      float* __restrict__ out_img, float* __restrict__ out_tgt,
                                                      bf16_t* __restrict__ out_nhwc8, AugGeom g) {
  const int b = blockIdx.y;
  const AugDraw d = aug_draw(g, b);
  const int groups = g.W >> 2;
  const long long HW = (long long)g.H * g.W;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < g.H * groups; i += gridDim.x * 256) {
    const int y = i / groups, x0 = (i - y * groups) << 2;
    const int sy = d.oy + (d.vflip ? g.H - 1 - y : y);
    size_t src[4];
    bool hole[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int x = x0 + e;
      const int sx = d.ox + (d.hflip ? g.W - 1 - x : x);
      src[e] = ((size_t)b * g.Hs + sy) * g.Ws + sx;
      hole[e] = d.drop && y >= d.y1 && y < d.y1 + d.hh && x >= d.x1 && x < d.x1 + d.hw;
    }
    if (img) {
      float v[3][4];
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          const float px = hole[e] ? 0.f : (float)img[src[e] * 3 + c];
          v[c][e] = (px - g.mean[c]) / g.std[c];                       // dataset.py:570
        }
      if (out_img) {
#pragma unroll
        for (int c = 0; c < 3; ++c)
          *(float4*)(out_img + ((size_t)b * 3 + c) * HW + (size_t)y * g.W + x0) = make_float4(v[c][0], v[c][1], v[c][2], v[c][3]);
      }
      if (out_nhwc8) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float f[8] = {v[0][e], v[1][e], v[2][e], 0.f, 0.f, 0.f, 0.f, 0.f};
          *(uint4*)(out_nhwc8 + (((size_t)b * g.H + y) * g.W + x0 + e) * 8) = pack8f(f);
        }
      }
    }
    if (tgt) {
      for (int c = 0; c < g.C; ++c) {
        float o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float px = hole[e] ? 0.f : (float)tgt[src[e] * g.C + c];
          o[e] = px / 255.f * 1.8f - 0.9f;                             // dataset.py:573: np.float32(x) / 255 * 1.8 - 0.9
        }
        *(float4*)(out_tgt + ((size_t)b * g.C + c) * HW + (size_t)y * g.W + x0) = make_float4(o[0], o[1], o[2], o[3]);
      }
    }
  }
}
#pragma clang fp contract(fast)

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

MVIT_API int mvit_augment_tiles(const void* img_u8, const void* tgt_u8, float* out_img, float* out_tgt, void* out_nhwc8, int B,
                                int C, int Hs, int Ws, int H, int W, unsigned long long seed, unsigned long long sample0,
                                float p_hflip, float p_vflip, float p_drop, float hole_frac, const float* mean3,
                                const float* std3, mvit_stream_t stream) {
  MVIT_CLEAR_ERROR();
  if (B <= 0 || B > 65535 || H <= 0 || W <= 0 || (W & 3) || H > Hs || W > Ws || (!img_u8 && !tgt_u8)) return MVIT_EINVAL;
  if (img_u8 && ((!out_img && !out_nhwc8) || !mean3 || !std3)) return MVIT_EINVAL;
  if (tgt_u8 && (!out_tgt || C <= 0)) return MVIT_EINVAL;
  if (!(hole_frac >= 0.f && hole_frac <= 1.f)) return MVIT_EINVAL;
  AugGeom g;
  g.B = B, g.C = C, g.Hs = Hs, g.Ws = Ws, g.H = H, g.W = W, g.seed = seed, g.sample0 = sample0;
  g.p_hflip = p_hflip, g.p_vflip = p_vflip, g.p_drop = p_drop, g.hole_frac = hole_frac;
  for (int c = 0; c < 3; ++c) g.mean[c] = mean3 ? mean3[c] : 0.f, g.std[c] = std3 ? std3[c] : 1.f;
  const int bx = nblk((long long)H * (W >> 2), 256, 1024);
  hipLaunchKernelGGL(augment_kernel, dim3(bx, B), dim3(256), 0, (hipStream_t)stream, (const uint8_t*)img_u8,
                     (const uint8_t*)tgt_u8, out_img, out_tgt, (bf16_t*)out_nhwc8, g);
  return MVIT_LAUNCH_CHECK();
}

/* host-side recomputation of one sample's draws (same arithmetic as the kernel): out9 = oy, ox, hflip, vflip, drop, y1, x1, hh, hw */
MVIT_API int mvit_augment_draw(int Hs, int Ws, int H, int W, unsigned long long seed, unsigned long long sample, float p_hflip,
                               float p_vflip, float p_drop, float hole_frac, int* out9) {
  if (!out9 || H <= 0 || W <= 0 || H > Hs || W > Ws) return MVIT_EINVAL;
  AugGeom g;
  g.B = 1, g.C = 0, g.Hs = Hs, g.Ws = Ws, g.H = H, g.W = W, g.seed = seed, g.sample0 = sample;
  g.p_hflip = p_hflip, g.p_vflip = p_vflip, g.p_drop = p_drop, g.hole_frac = hole_frac;
  const AugDraw d = aug_draw(g, 0);
  const int v[9] = {d.oy, d.ox, d.hflip, d.vflip, d.drop, d.y1, d.x1, d.hh, d.hw};
  for (int i = 0; i < 9; ++i) out9[i] = v[i];
  return MVIT_OK;
}

MVIT_API int mvit_f32_to_u8_export(const float* src, void* dst_u8, long long n, mvit_stream_t stream) {
  MVIT_CLEAR_ERROR();
  if (n <= 0 || (n & 3)) return MVIT_EINVAL;
  hipLaunchKernelGGL(f32_to_u8_export_kernel, dim3(nblk(n, 1024, 8192)), dim3(256), 0, (hipStream_t)stream, src,
                     (uint8_t*)dst_u8, n);
  return MVIT_LAUNCH_CHECK();
}

}  // extern "C"
