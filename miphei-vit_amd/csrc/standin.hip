// Multi-GPU pre-flight helper (round 5; not on the compute path): a kernel that HOLDS `blocks` workgroups of 256 threads on the
// chip for `usec` microseconds with the register footprint of a collective's ring kernel (64 VGPRs per lane, no LDS), doing no
// memory traffic.  bench.py --comm-standin launches it on a side stream at the five points of the backward pass where
// trainer.DataParallelSync issues its gradient buckets: on a one-GPU box it measures what RCCL's kernels cost the persistent,
// register-file-filling GEMM blocks in CU residency (a wave-specialised GEMM block takes 3 x 168 = 504 of a SIMD's 512 registers:
// nothing co-resides, so a CU held by the stand-in is a CU the next one-round GEMM launch does not get).
// Replaces nothing in the reference (single-device: /root/reference/src/train.py:205-207).
#include "common.hpp"
#include "../../include/miphei_hip.h"

namespace {
__global__ __launch_bounds__(256) void occupy_kernel(unsigned long long ticks, int* sink) {
  // 64 VGPRs allocated per lane (v63 named in the clobber list): the occupancy cost of a real ring kernel, not of an empty one
  asm volatile("v_mov_b32 v63, 0" ::: "v63");
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();      // 100 MHz, common to all CUs
  while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
  if (sink && ticks == ~0ull) *sink = 1;
}
}  // namespace

extern "C" MVIT_API int mvit_occupy_cus(int blocks, int usec, mvit_stream_t stream) {
  MVIT_CLEAR_ERROR();
  if (blocks <= 0 || usec < 0 || blocks > 1024 || usec > 100000) return MVIT_EINVAL;
  hipLaunchKernelGGL(occupy_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (unsigned long long)usec * 100ull, (int*)nullptr);
  return MVIT_LAUNCH_CHECK();
}
