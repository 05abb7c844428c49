// Shared device helpers for the MIPHEI-ViT gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef uint16_t bf16_t;  // raw bf16 bits in memory
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

#define MVIT_OK 0
#define MVIT_EINVAL (-1)

// ---- the 16-bit operand type.  Every kernel stores its GEMM / convolution / attention operands as raw 16-bit patterns (`bf16_t`,
// fragments as `bf16x8`) and touches their VALUE only through the helpers below; the product library is built with bf16 operands
// (BASELINE.json's precision).  -DMVIT_F16 builds the same sources with IEEE fp16 operands on v_mfma_f32_*_f16 (libmiphei_hip_f16.so):
// the arithmetic type of the reference's evaluation convention `generator.eval().cuda().half()`
// (/root/reference/evaluation/eval_orion.py:191, 214-215), selected by the engine when the module's parameters are fp16 -- without it
// an fp16 model's weights were rounded a second time (fp16 -> bf16).
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
#ifdef MVIT_F16
typedef __attribute__((ext_vector_type(8))) _Float16 op16x8_t;
typedef __attribute__((ext_vector_type(2))) _Float16 op16x2_t;
// low / high 16-bit operand of a packed word as f32
__device__ __forceinline__ float lo16f(uint32_t u) { return (float)__builtin_bit_cast(_Float16, (uint16_t)(u & 0xffffu)); }
__device__ __forceinline__ float hi16f(uint32_t u) { return (float)__builtin_bit_cast(_Float16, (uint16_t)(u >> 16)); }
// fp32 pair -> packed operands, round-to-nearest-even (one v_cvt_pk_f16_f32)
__device__ __forceinline__ uint32_t pack2bf(float lo, float hi) {
  const f32x2_t v = {lo, hi};
  const op16x2_t r = __builtin_convertvector(v, op16x2_t);
  return __builtin_bit_cast(uint32_t, r);
}
#define MVIT_ONE2 0x3c003c00u          /* two packed 1.0 */
// c + a.lo * b.lo + a.hi * b.hi on packed operand words (v_dot2_f32_f16)
__device__ __forceinline__ float mvit_dot2(uint32_t a, uint32_t b, float c) {
  return __builtin_amdgcn_fdot2(__builtin_bit_cast(op16x2_t, a), __builtin_bit_cast(op16x2_t, b), c, false);
}
#define mvit_mfma16(a, b, c, x, y, z) __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(op16x8_t, a), __builtin_bit_cast(op16x8_t, b), c, x, y, z)
#define mvit_mfma32(a, b, c, x, y, z) __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(op16x8_t, a), __builtin_bit_cast(op16x8_t, b), c, x, y, z)
#else
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
__device__ __forceinline__ float lo16f(uint32_t u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float hi16f(uint32_t u) { return __uint_as_float(u & 0xffff0000u); }
// fp32 -> bf16, round-to-nearest-even, on the gfx950 hardware converter (one v_cvt_pk_bf16_f32 per pair)
__device__ __forceinline__ uint32_t pack2bf(float lo, float hi) {
  const f32x2_t v = {lo, hi};
  const bf16x2_t r = __builtin_convertvector(v, bf16x2_t);
  return *(const uint32_t*)&r;
}
#define MVIT_ONE2 0x3f803f80u
// c + a.lo * b.lo + a.hi * b.hi on packed operand words (v_dot2c_f32_bf16)
__device__ __forceinline__ float mvit_dot2(uint32_t a, uint32_t b, float c) {
  return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, a), __builtin_bit_cast(bf16x2_t, b), c, false);
}
#define mvit_mfma16(a, b, c, x, y, z) __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, x, y, z)
#define mvit_mfma32(a, b, c, x, y, z) __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, x, y, z)
#endif
__device__ __forceinline__ float bf2f(bf16_t v) { return lo16f((uint32_t)v); }
__device__ __forceinline__ bf16_t f2bf(float f) { return (bf16_t)(pack2bf(f, 0.f) & 0xffffu); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// sigmoid on the hardware reciprocal (v_rcp_f32, 1 ulp) instead of an IEEE division: `1.f / x` compiles to v_div_scale x 2 + v_rcp +
// 4 v_fma + v_div_fmas + v_div_fixup -- ten VALU instructions per element in epilogues whose run time is VALU issue (round 4:
// the d(SwiGLU) epilogue issued 3.4 VALU per MFMA over the whole kernel, profiles/r04_gemm_sq_counters_baseline.txt)
__device__ __forceinline__ float sigmoidf_(float x) { return __builtin_amdgcn_rcpf(1.f + __expf(-x)); }
__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.f + erff(x * 0.70710678118654752f)); }
__device__ __forceinline__ float gelu_erf_grad(float x) {
  const float cdf = 0.5f * (1.f + erff(x * 0.70710678118654752f));
  const float pdf = 0.3989422804014327f * __expf(-0.5f * x * x);
  return cdf + x * pdf;
}

// floor(x / d) without a division instruction sequence: mulhi(x, floor(2^32 / d) + 1), exact for x < 2^32 / d; d == 1 has no 32-bit
// constant.  A division by a run-time value costs a wave ~25 instructions and two trips through the vector unit (v_rcp_iflag +
// readfirstlane); in a kernel prologue every wave of the CU runs it at once on the CU's one scalar unit.
__device__ __forceinline__ unsigned mvit_fast_div(unsigned x, unsigned d, unsigned magic) { return d == 1 ? x : __umulhi(x, magic); }
static inline unsigned mvit_div_magic(unsigned d) { return d <= 1 ? 0u : (unsigned)(0x100000000ull / d) + 1u; }

// Host-side per-device state.  hipFuncSetAttribute(MaxDynamicSharedMemorySize) and the CU count are properties of the CURRENT
// device, so anything cached about them is keyed by hipGetDevice() (a process may drive several GPUs, from several threads).
// Relaxed atomics are enough: the cached value only ever grows and re-applying an attribute is idempotent.
#include <atomic>
#define MVIT_MAX_DEVICES 64
struct mvit_per_device_size {
  std::atomic<size_t> v[MVIT_MAX_DEVICES];
};
// Make sure kernel `fn` may be launched with `bytes` of dynamic LDS on the current device (no-op below the 64 KB default).
static inline int mvit_ensure_dynamic_lds(const void* fn, size_t bytes, mvit_per_device_size& raised) {
  if (bytes <= 64 * 1024) return MVIT_OK;
  if (bytes > 160 * 1024) return MVIT_EINVAL;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MVIT_MAX_DEVICES) return MVIT_EINVAL;
  if (bytes <= raised.v[dev].load(std::memory_order_relaxed)) return MVIT_OK;
  if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess) return MVIT_EINVAL;
  size_t cur = raised.v[dev].load(std::memory_order_relaxed);
  while (cur < bytes && !raised.v[dev].compare_exchange_weak(cur, bytes, std::memory_order_relaxed)) {
  }
  return MVIT_OK;
}

// CU count of the current device (cached per device)
static inline int mvit_num_cus() {
  static std::atomic<int> cus[MVIT_MAX_DEVICES];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MVIT_MAX_DEVICES) return 256;
  int cu = cus[dev].load(std::memory_order_relaxed);
  if (cu <= 0) {
    if (hipDeviceGetAttribute(&cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cu <= 0) cu = 256;
    cus[dev].store(cu, std::memory_order_relaxed);
  }
  return cu;
}

static inline int hip_ok(hipError_t e) { return e == hipSuccess ? MVIT_OK : (int)e; }
#define MVIT_LAUNCH_CHECK() hip_ok(hipGetLastError())
// hipGetLastError() reports (and resets) the last error of ANY earlier HIP call of this thread, including benign
// probes made by other libraries in the process: reset it on entry so the check after our launch is about our launch.
#define MVIT_CLEAR_ERROR() (void)hipGetLastError()
