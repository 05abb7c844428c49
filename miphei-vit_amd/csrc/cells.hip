// Per-nucleus mean intensities (validation-time cell extractor) as a segmented reduction.
// Reference: MeanCellExtrator.forward / extract_mean, /root/reference/src/utils.py:23-121 (optional area down-sampling, then per
// image torch.unique of the non-zero labels + scatter_add sums / counts) and CellMetrics.update, src/metrics.py:38-74.
//
// Nucleus ids are slide-global (sparse, up to millions) but spatially coherent, so nothing here is sized by the label VALUE:
//   1. cell_tile_kernel: one block per 16 x 128 tile of (down-sampled) pixels.  A thread walks a run of 8 pixels of one row and
//      keeps the channel sums of the current label in registers; a label change flushes the run into an LDS hash table of the
//      tile (key = label, 2C+1 f32 accumulators, ds_add_f32).  At the end every occupied slot becomes one partial record
//      {label, count, sums} of the image's record list (one global atomic per record, not per pixel and channel).
//      The area down-sampling (adaptive average windows) and the nearest-exact gather of the label map are done on the fly.
//   2. cell_merge_kernel: one block per image sorts the image's record keys (bitonic sort of (label, record) pairs in LDS),
//      finds the segment heads, and every head sums its few partial records: labels come out ascending (= torch.unique order),
//      compacted, with their pixel counts and per-channel means (or sums).
// HBM traffic: the images and the label map once; records are a few hundred KB.
#include "common.hpp"
#include "../../include/miphei_hip.h"

namespace {

constexpr int CT_ROWS = 16, CT_COLS = 128, CT_RUN = 8;   // tile geometry: 256 threads x 8 pixels
constexpr int CT_SLOTS = 256;                            // LDS hash slots per tile
constexpr int CM_THREADS = 1024;

struct CellGeom {
  int B, C, H, W, Ho, Wo;   // source and (down-sampled) working resolution
  float inv_scale;          // 1 / scale_factor (nearest-exact gather of the label map)
  int lab64;                // label map is int64 (else int32)
  int rmax;                 // record capacity per image
};

__device__ __forceinline__ int load_label(const void* nuclei, size_t i, int lab64) {
  if (lab64) {
    const long long v = ((const long long*)nuclei)[i];
    return v > 0x7fffffffLL ? -1 : (int)v;     // ids beyond int32: cells.py checks nuclei.max() on the host and raises before the launch
  }
  return ((const int*)nuclei)[i];
}

__global__ __launch_bounds__(256) void cell_tile_kernel(const float* __restrict__ pred, const float* __restrict__ target,
                                                        const void* __restrict__ nuclei, int* __restrict__ rec_count,
                                                        int* __restrict__ rec_key, float* __restrict__ rec_val, CellGeom g) {
  extern __shared__ __attribute__((aligned(16))) char lds_raw[];
  const int C = g.C, NV = 2 * C + 1;
  int* keys = (int*)lds_raw;                        // [CT_SLOTS]
  float* vals = (float*)(keys + CT_SLOTS);          // [CT_SLOTS][NV]: count, pred sums, target sums
  for (int i = threadIdx.x; i < CT_SLOTS * (NV + 1); i += 256) ((int*)lds_raw)[i] = 0;
  __syncthreads();
  const int tiles_x = (g.Wo + CT_COLS - 1) / CT_COLS;
  const int ty = blockIdx.x / tiles_x, tx = blockIdx.x - ty * tiles_x, b = blockIdx.y;
  const int oy = ty * CT_ROWS + (threadIdx.x >> 4), ox0 = tx * CT_COLS + (threadIdx.x & 15) * CT_RUN;
  const bool ident = g.H == g.Ho && g.W == g.Wo;
  const size_t HW = (size_t)g.H * g.W;
  const float* pb = pred + (size_t)b * C * HW;
  const float* tb = target ? target + (size_t)b * C * HW : nullptr;

  auto append_global = [&](int key, const float* v, bool from_lds) {
    const int r = atomicAdd(rec_count + b, 1);
    if (r < g.rmax) {
      rec_key[(size_t)b * g.rmax + r] = key;
      float* dst = rec_val + ((size_t)b * g.rmax + r) * NV;
      for (int c = 0; c < NV; ++c) dst[c] = v[c];
    }
    (void)from_lds;
  };

  if (oy < g.Ho) {
    int cur = 0;
    float cnt = 0.f;
    // run accumulators live in LDS-free registers only for small C; for generality they are flushed per pixel group below
    constexpr int CMAX = 32;
    float sp[CMAX], st[CMAX];
#pragma unroll
    for (int c = 0; c < CMAX; ++c) sp[c] = st[c] = 0.f;
    auto flush = [&]() {
      if (cur > 0 && cnt > 0.f) {
        unsigned h = ((unsigned)cur * 2654435761u) >> 24;   // 8 bits: CT_SLOTS = 256
        int slot = -1;
        for (int probe = 0; probe < CT_SLOTS; ++probe) {
          const int k = atomicCAS(keys + h, 0, cur);
          if (k == 0 || k == cur) {
            slot = (int)h;
            break;
          }
          h = (h + 1) & (CT_SLOTS - 1);
        }
        if (slot >= 0) {
          float* v = vals + (size_t)slot * NV;
          atomicAdd(v, cnt);
#pragma unroll
          for (int c = 0; c < CMAX; ++c)
            if (c < C) {
              atomicAdd(v + 1 + c, sp[c]);
              if (tb) atomicAdd(v + 1 + C + c, st[c]);
            }
        } else {   // more than CT_SLOTS distinct labels in one tile (per-pixel labels): the run goes out as its own record
          float tmp[2 * CMAX + 1];
          tmp[0] = cnt;
          for (int c = 0; c < C; ++c) tmp[1 + c] = sp[c], tmp[1 + C + c] = tb ? st[c] : 0.f;
          append_global(cur, tmp, false);
        }
      }
      cnt = 0.f;
#pragma unroll
      for (int c = 0; c < CMAX; ++c) sp[c] = st[c] = 0.f;
    };
    // window of the area down-sampling (adaptive average pooling): rows [y0, y1), and per pixel columns [x0, x1)
    const int y0 = ident ? oy : (int)(((long long)oy * g.H) / g.Ho);
    const int y1 = ident ? oy + 1 : (int)((((long long)(oy + 1)) * g.H + g.Ho - 1) / g.Ho);
    const int sy = ident ? oy : min((int)floorf(((float)oy + 0.5f) * g.inv_scale), g.H - 1);
    for (int k = 0; k < CT_RUN; ++k) {
      const int ox = ox0 + k;
      if (ox >= g.Wo) break;
      const int sx = ident ? ox : min((int)floorf(((float)ox + 0.5f) * g.inv_scale), g.W - 1);
      const int lab = load_label(nuclei, ((size_t)b * g.H + sy) * g.W + sx, g.lab64);
      if (lab != cur) {
        flush();
        cur = lab;
      }
      if (lab <= 0) continue;
      cnt += 1.f;
      if (ident) {
        const size_t o = (size_t)oy * g.W + ox;
#pragma unroll
        for (int c = 0; c < CMAX; ++c)
          if (c < C) {
            sp[c] += pb[c * HW + o];
            if (tb) st[c] += tb[c * HW + o];
          }
      } else {
        const int x0 = (int)(((long long)ox * g.W) / g.Wo), x1 = (int)((((long long)(ox + 1)) * g.W + g.Wo - 1) / g.Wo);
        const float inv = 1.f / (float)((y1 - y0) * (x1 - x0));
#pragma unroll
        for (int c = 0; c < CMAX; ++c)
          if (c < C) {
            float a = 0.f, t = 0.f;
            for (int y = y0; y < y1; ++y)
              for (int x = x0; x < x1; ++x) {
                a += pb[c * HW + (size_t)y * g.W + x];
                if (tb) t += tb[c * HW + (size_t)y * g.W + x];
              }
            sp[c] += a * inv;
            st[c] += t * inv;
          }
      }
    }
    flush();
  }
  __syncthreads();
  for (int s = threadIdx.x; s < CT_SLOTS; s += 256)
    if (keys[s] != 0) append_global(keys[s], vals + (size_t)s * NV, true);
}

// bitonic sort of n 64-bit (label << 32 | record) pairs padded to P (power of two) with ~0
__device__ void bitonic_sort(unsigned long long* a, int P) {
  for (int k = 2; k <= P; k <<= 1)
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = threadIdx.x; i < P; i += CM_THREADS) {
        const int l = i ^ j;
        if (l > i) {
          const unsigned long long x = a[i], y = a[l];
          const bool up = (i & k) == 0;
          if ((x > y) == up) a[i] = y, a[l] = x;
        }
      }
      __syncthreads();
    }
}

__global__ __launch_bounds__(CM_THREADS) void cell_merge_kernel(const int* __restrict__ rec_count, const int* __restrict__ rec_key,
                                                                const float* __restrict__ rec_val, int* __restrict__ n_unique,
                                                                int* __restrict__ out_ids, float* __restrict__ out_cnt,
                                                                float* __restrict__ out_p, float* __restrict__ out_t, int C,
                                                                int rmax, int want_sums) {
  extern __shared__ __attribute__((aligned(16))) char lds_raw[];
  __shared__ int wave_tot[CM_THREADS / 64];
  __shared__ int total_s;
  unsigned long long* pairs = (unsigned long long*)lds_raw;
  const int b = blockIdx.x, NV = 2 * C + 1;
  const int n = min(rec_count[b], rmax);
  int P = 1;
  while (P < n) P <<= 1;
  if (P < 2) P = 2;
  for (int i = threadIdx.x; i < P; i += CM_THREADS)
    pairs[i] = i < n ? ((unsigned long long)(unsigned)rec_key[(size_t)b * rmax + i] << 32) | (unsigned)i : ~0ull;
  __syncthreads();
  bitonic_sort(pairs, P);
  // segment heads -> compact index (block-wide exclusive scan over per-thread chunks)
  const int per = (P + CM_THREADS - 1) / CM_THREADS;
  const int lo = threadIdx.x * per, hi = min(lo + per, n);
  int mine = 0;
  for (int i = lo; i < hi; ++i) mine += (i == 0 || (pairs[i] >> 32) != (pairs[i - 1] >> 32)) ? 1 : 0;
  int incl = mine;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int v = __shfl_up(incl, o, 64);
    if (lane >= o) incl += v;
  }
  if (lane == 63) wave_tot[wave] = incl;
  __syncthreads();
  if (threadIdx.x == 0) {
    int run = 0;
    for (int w = 0; w < CM_THREADS / 64; ++w) {
      const int t = wave_tot[w];
      wave_tot[w] = run;
      run += t;
    }
    total_s = run;
    n_unique[b] = run;
  }
  __syncthreads();
  int u = wave_tot[wave] + incl - mine;
  for (int i = lo; i < hi; ++i) {
    const unsigned key = (unsigned)(pairs[i] >> 32);
    if (i != 0 && key == (unsigned)(pairs[i - 1] >> 32)) continue;
    // head of a segment: sum its partial records
    float cnt = 0.f;
    for (int j = i; j < n && (unsigned)(pairs[j] >> 32) == key; ++j)
      cnt += rec_val[((size_t)b * rmax + (unsigned)pairs[j]) * NV];
    const size_t o = (size_t)b * rmax + u;
    out_ids[o] = (int)key;
    out_cnt[o] = cnt;
    const float d = want_sums ? 1.f : 1.f / cnt;
    for (int c = 0; c < C; ++c) {
      float sp = 0.f, st = 0.f;
      for (int j = i; j < n && (unsigned)(pairs[j] >> 32) == key; ++j) {
        const float* v = rec_val + ((size_t)b * rmax + (unsigned)pairs[j]) * NV;
        sp += v[1 + c];
        st += v[1 + C + c];
      }
      out_p[o * C + c] = sp * d;
      if (out_t) out_t[o * C + c] = st * d;
    }
    ++u;
  }
}

}  // namespace

extern "C" {

MVIT_API int mvit_cell_means(const float* pred, const float* target, const void* nuclei, int label_is_int64, int B, int C, int H,
                             int W, float scale_factor, int rmax, int want_sums, int* rec_count, int* rec_key, float* rec_val,
                             int* n_unique, int* out_ids, float* out_count, float* out_pred, float* out_target,
                             mvit_stream_t stream) {
  MVIT_CLEAR_ERROR();
  if (!pred || !nuclei || B <= 0 || C <= 0 || C > 32 || H <= 0 || W <= 0 || !(scale_factor > 0.f) || scale_factor > 1.f ||
      rmax < 2 || rmax > 16384 || (rmax & (rmax - 1)) || !rec_count || !rec_key || !rec_val || !n_unique || !out_ids || !out_count || !out_pred)
    return MVIT_EINVAL;
  CellGeom g;
  g.B = B, g.C = C, g.H = H, g.W = W;
  g.Ho = scale_factor < 1.f ? (int)floor((double)H * (double)scale_factor) : H;
  g.Wo = scale_factor < 1.f ? (int)floor((double)W * (double)scale_factor) : W;
  if (g.Ho <= 0 || g.Wo <= 0) return MVIT_EINVAL;
  g.inv_scale = 1.f / scale_factor;
  g.lab64 = label_is_int64;
  g.rmax = rmax;
  hipStream_t s = (hipStream_t)stream;
  if (hipMemsetAsync(rec_count, 0, sizeof(int) * B, s) != hipSuccess) return MVIT_EINVAL;
  const int tiles = ((g.Ho + CT_ROWS - 1) / CT_ROWS) * ((g.Wo + CT_COLS - 1) / CT_COLS);
  const size_t lds1 = (size_t)CT_SLOTS * (2 * C + 2) * 4;
  static mvit_per_device_size raised1, raised2;
  if (mvit_ensure_dynamic_lds((const void*)cell_tile_kernel, lds1, raised1) != MVIT_OK) return MVIT_EINVAL;
  hipLaunchKernelGGL(cell_tile_kernel, dim3(tiles, B), dim3(256), lds1, s, pred, target, nuclei, rec_count, rec_key, rec_val, g);
  const size_t lds2 = (size_t)rmax * 8;
  if (mvit_ensure_dynamic_lds((const void*)cell_merge_kernel, lds2, raised2) != MVIT_OK) return MVIT_EINVAL;
  hipLaunchKernelGGL(cell_merge_kernel, dim3(B), dim3(CM_THREADS), lds2, s, rec_count, rec_key, rec_val, n_unique, out_ids,
                     out_count, out_pred, target ? out_target : nullptr, C, rmax, want_sums);
  return MVIT_LAUNCH_CHECK();
}

}  // extern "C"
