// Per-nucleus mean intensities (validation-time cell extractor).
// Reference: MeanCellExtrator.extract_mean, /root/reference/src/utils.py:49-121 (torch.unique + scatter_add per image).
// One pass over the label map: every pixel with label > 0 adds its C channel values of pred / target and 1 to the
// row of its label in dense per-image tables (f32 atomics; HBM-bound integer-label scan).  The host compacts the
// non-empty rows in label order (= torch.unique's sorted order) and divides.
#include "common.hpp"
#include "../../include/miphei_hip.h"

namespace {

__global__ __launch_bounds__(256) void cell_sums_kernel(const float* __restrict__ pred, const float* __restrict__ target,
                                                        const int* __restrict__ nuclei, float* __restrict__ sums_p,
                                                        float* __restrict__ sums_t, float* __restrict__ counts, int B,
                                                        int C, long long HW, int L) {
  const long long total = (long long)B * HW;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int b = (int)(i / HW);
    const long long pix = i - (long long)b * HW;
    const int lab = nuclei[i];
    if (lab <= 0 || lab > L) continue;
    const size_t row = (size_t)b * (L + 1) + lab;
    atomicAdd(counts + row, 1.f);
    for (int c = 0; c < C; ++c) {
      atomicAdd(sums_p + row * C + c, pred[((size_t)b * C + c) * HW + pix]);
      if (target) atomicAdd(sums_t + row * C + c, target[((size_t)b * C + c) * HW + pix]);
    }
  }
}

}  // namespace

extern "C" MVIT_API int mvit_cell_sums(const float* pred, const float* target, const int* nuclei, float* sums_pred,
                                       float* sums_target, float* counts, int B, int C, long long HW, int max_label,
                                       mvit_stream_t stream) {
  MVIT_CLEAR_ERROR();
  if (B <= 0 || C <= 0 || HW <= 0 || max_label <= 0) return MVIT_EINVAL;
  long long blocks = ((long long)B * HW + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(cell_sums_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, pred, target, nuclei,
                     sums_pred, sums_target, counts, B, C, HW, max_label);
  return MVIT_LAUNCH_CHECK();
}
