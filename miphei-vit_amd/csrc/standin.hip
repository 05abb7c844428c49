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

// ---------------------------------------------------------------------------------------------------------------------------------
// XCD speed probe (round 5).  Under a dense bf16 MFMA load the eight XCDs of an MI355X settle at clocks several per cent apart (a
// property of the part: stable over a run, different from box to box; tools/ws_timing.py).  Every workgroup (one per CU, 512 threads =
// two waves per SIMD, as the GEMM's consumers) runs the same loop of v_mfma_f32_16x16x32_bf16 on pseudo-random register operands and
// records its own duration on the 100 MHz clock all CUs share; the host averages per XCD (workgroup L runs on XCD L % 8) and ranks
// them (miphei_vit_amd/xcd.py) -- the wave-specialised GEMM then gives the band items of a ragged launch to the fastest XCDs
// (mvit_set_xcd_rank).  Replaces nothing in the reference.
namespace {
__global__ __launch_bounds__(512) void xcd_probe_kernel(unsigned long long* __restrict__ out, int iters) {
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  unsigned s = threadIdx.x * 2654435761u + 12345u;
  bf16x8 a[4], b[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    union { uint32_t u[4]; bf16x8 v; } ua, ub;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      s = s * 1664525u + 1013904223u;
      ua.u[e] = (s & 0x807f807fu) | 0x3f003f00u;          // two bf16 in [0.5, 1) with random sign and mantissa
      s = s * 1664525u + 1013904223u;
      ub.u[e] = (s & 0x807f807fu) | 0x3f003f00u;
    }
    a[i] = ua.v, b[i] = ub.v;
  }
  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
  }
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) sum += acc[i][j][0] + acc[i][j][3];
  const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0 + (sum == 12345.678f ? 1ull : 0ull);
}
}  // namespace

// out[blocks]: duration of workgroup L in 10 ns ticks; iters x 16 MFMAs per wave (iters = 2000: ~0.3 ms)
extern "C" MVIT_API int mvit_xcd_probe(unsigned long long* out, int blocks, int iters, mvit_stream_t stream) {
  MVIT_CLEAR_ERROR();
  if (!out || blocks <= 0 || blocks > 4096 || iters <= 0 || iters > 1000000) return MVIT_EINVAL;
  hipLaunchKernelGGL(xcd_probe_kernel, dim3(blocks), dim3(512), 0, (hipStream_t)stream, out, iters);
  return MVIT_LAUNCH_CHECK();
}
