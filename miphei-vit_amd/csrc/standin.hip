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
// Infinity-Cache warmer (round 5).  The backward pass re-reads tensors the forward pass saved GBs of traffic ago (the packed fc1
// pre-activation of a block: 86 MB at batch 16): they come from HBM exactly when their consumer's epilogue waits for them with the
// matrix pipe idle, while during the GEMM main loops the HBM is mostly idle (the operand stream is served by L2 / Infinity Cache).
// This kernel shifts such reads into the GEMM phases: launched on a side stream one or two kernels ahead of the consumer, it touches
// one dword per 128-byte line of [p, p + bytes) with non-temporal loads -- the lines are allocated in the 256 MB memory-side cache
// on their way -- and discards the values.  One wave per block, TWO vector registers per lane, no LDS: a wave fits beside the three
// 168-register waves of a persistent GEMM block on a SIMD (504 + 8 <= 512), so it runs concurrently instead of waiting for a CU.
namespace {
__global__ __launch_bounds__(64) void warm_kernel(const char* __restrict__ p, unsigned long long bytes, unsigned long long per, int pace) {
  const unsigned long long beg = per * blockIdx.x;                                     // per: bytes per wave, whole 8 KB strides
  if (beg >= bytes) return;
  const unsigned long long len = bytes - beg < per ? bytes - beg : per;
  const char* base = p + beg;
  // descriptor over this wave's chunk: lanes beyond the end read nothing (range check), the stride rides on the scalar offset
  const unsigned long long v = (unsigned long long)base;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)v, 0, (int)(len > 0x7fffffffull ? 0x7fffffff : len), 0x00020000);
  const unsigned vo = (threadIdx.x & 63) * 128u;
  unsigned sink = 0;
  for (unsigned long long so = 0; so < len; so += 8192) {
    // (the destination register is reused without a wait: the values are never read)
    asm volatile("buffer_load_dword %0, %1, %2, %3 offen nt" : "=v"(sink) : "v"(vo), "s"(rs), "s"((unsigned)so) : "memory");
    // pacing: `pace` x 64 idle cycles between two 8 KB touches of a wave -- the warmer is meant to trickle (1-2 TB/s over the chip)
    // beside a GEMM's operand stream, not to race it for the fabric
    for (int i = 0; i < pace; ++i) __builtin_amdgcn_s_sleep(1);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
}  // namespace

extern "C" MVIT_API int mvit_prefetch_cache(const void* p, long long bytes, int waves, int pace, mvit_stream_t stream) {
  MVIT_CLEAR_ERROR();
  if (!p || bytes <= 0 || waves <= 0 || waves > 4096 || pace < 0 || pace > 100000) return MVIT_EINVAL;
  if ((unsigned long long)bytes / (unsigned)waves >= 0x7fffffffull) return MVIT_EINVAL;    // (32-bit offsets inside a wave's chunk)
  const unsigned long long per = (((unsigned long long)bytes / (unsigned)waves + 8191) / 8192) * 8192;
  hipLaunchKernelGGL(warm_kernel, dim3(waves), dim3(64), 0, (hipStream_t)stream, (const char*)p, (unsigned long long)bytes, per, pace);
  return MVIT_LAUNCH_CHECK();
}
