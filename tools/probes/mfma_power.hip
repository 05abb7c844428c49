// Probe (not part of the library): MFMA-only loops on random bf16 register operands, run for a few seconds each, to compare
// what the chip sustains under its power cap for v_mfma_f32_32x32x16_bf16 vs v_mfma_f32_16x16x32_bf16 (same flops per
// instruction-byte, different accumulator / operand register traffic per flop).  One wave per SIMD (256 threads, 1 block per
// CU) or two (argv[1] = 2).  Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/probes/mfma_power.hip -o tools/probes/mfma_power
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void k32(const uint4* __restrict__ src, float* __restrict__ out, int iters) {
  bf16x8 a[4], b[4];
  for (int i = 0; i < 4; ++i) {
    uint4 t = src[(threadIdx.x * 8 + i) & 4095], u = src[(threadIdx.x * 8 + 4 + i) & 4095];
    a[i] = *(bf16x8*)&t; b[i] = *(bf16x8*)&u;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  for (int i = 0; i < 4; ++i) asm volatile("" : "+v"(a[i]), "+v"(b[i]));   // operands settled before the loop: it holds MFMAs only
  f32x16 acc[4][4];
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) s += acc[i][j][r];
  if (s == 12345.678f) out[0] = s;
}
__global__ __launch_bounds__(256) void k16(const uint4* __restrict__ src, float* __restrict__ out, int iters) {
  bf16x8 a[8], b[8];
  for (int i = 0; i < 8; ++i) {
    uint4 t = src[(threadIdx.x * 16 + i) & 4095], u = src[(threadIdx.x * 16 + 8 + i) & 4095];
    a[i] = *(bf16x8*)&t; b[i] = *(bf16x8*)&u;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  for (int i = 0; i < 8; ++i) asm volatile("" : "+v"(a[i]), "+v"(b[i]));
  f32x4 acc[8][8];
  for (int i = 0; i < 8; ++i) for (int j = 0; j < 8; ++j) for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < 8; ++i) for (int j = 0; j < 8; ++j) for (int r = 0; r < 4; ++r) s += acc[i][j][r];
  if (s == 12345.678f) out[0] = s;
}

__global__ __launch_bounds__(256) void k16b(const uint4* __restrict__ src, float* __restrict__ out, int iters) {
  bf16x8 a[8], b[8];
  for (int i = 0; i < 8; ++i) {
    uint4 t = src[(threadIdx.x * 16 + i) & 4095], u = src[(threadIdx.x * 16 + 8 + i) & 4095];
    a[i] = *(bf16x8*)&t; b[i] = *(bf16x8*)&u;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  for (int i = 0; i < 8; ++i) asm volatile("" : "+v"(a[i]), "+v"(b[i]));
  f32x4 acc[8][8];
  for (int i = 0; i < 8; ++i) for (int j = 0; j < 8; ++j) for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
        asm volatile("s_nop 0");
        __builtin_amdgcn_sched_barrier(0);
      }
  }
  float s = 0.f;
  for (int i = 0; i < 8; ++i) for (int j = 0; j < 8; ++j) for (int r = 0; r < 4; ++r) s += acc[i][j][r];
  if (s == 12345.678f) out[0] = s;
}
__global__ __launch_bounds__(256) void k16c(const uint4* __restrict__ src, float* __restrict__ out, int iters) {
  bf16x8 a[8], b[8];
  for (int i = 0; i < 8; ++i) {
    uint4 t = src[(threadIdx.x * 16 + i) & 4095], u = src[(threadIdx.x * 16 + 8 + i) & 4095];
    a[i] = *(bf16x8*)&t; b[i] = *(bf16x8*)&u;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  for (int i = 0; i < 8; ++i) asm volatile("" : "+v"(a[i]), "+v"(b[i]));
  f32x4 acc[4][4];
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int h = 0; h < 4; ++h)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i + 4 * (h & 1)], b[j + 4 * (h >> 1)], acc[i][j], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
        }
  }
  float s = 0.f;
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) for (int r = 0; r < 4; ++r) s += acc[i][j][r];
  if (s == 12345.678f) out[0] = s;
}

int main(int argc, char** argv) {
  const int wps = argc > 1 ? atoi(argv[1]) : 1;
  const bool zeros = getenv("RR_ZEROS") != nullptr;
  std::vector<uint16_t> h(4096 * 8);
  uint32_t s = 777;
  for (auto& v : h) { s = s * 1664525u + 1013904223u; float f = ((int)(s >> 9) % 2001 - 1000) / 1000.f; uint32_t u; memcpy(&u, &f, 4); v = zeros ? 0 : (uint16_t)(u >> 16); }
  uint4* d; float* o; hipMalloc(&d, h.size() * 2); hipMalloc(&o, 4);
  hipMemcpy(d, h.data(), h.size() * 2, hipMemcpyHostToDevice);
  const int grid = 256 * wps, iters = 20000;   // per launch: 16 (64) MFMAs x iters per wave
  for (int which = 0; which < 4; ++which) {
    auto launch = [&]() { if (which == 0) hipLaunchKernelGGL(k32, dim3(grid), dim3(256), 0, 0, d, o, iters); else if (which == 1) hipLaunchKernelGGL(k16, dim3(grid), dim3(256), 0, 0, d, o, iters); else if (which == 2) hipLaunchKernelGGL(k16b, dim3(grid), dim3(256), 0, 0, d, o, iters); else hipLaunchKernelGGL(k16c, dim3(grid), dim3(256), 0, 0, d, o, iters); };
    launch(); hipDeviceSynchronize();
    auto t0 = std::chrono::steady_clock::now();
    int n = 0;
    double dt = 0;
    while (dt < 3.0) { launch(); hipDeviceSynchronize(); ++n; dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); }
    // per wave and iteration: 16 x 32x32x16 (32768 flop each) for kernel 0, 64 x 16x16x32 (16384 flop each) for the others.
    // (Round 1-2 credited every kernel with 16 x 32768: the 16x16x32 loops were under-reported by 2x -- the 'half the rate' puzzle of
    // DESIGN.md section 6c was this line, not the instruction.)
    const double flops = (double)n * grid * 4 /*waves*/ * iters * (which == 0 ? 16 * 32768.0 : 64 * 16384.0);
    printf("%s %s, %d wave(s)/SIMD: %.1f TF/s sustained over %.1f s\n", which == 0 ? "v_mfma_f32_32x32x16_bf16" : which == 1 ? "v_mfma_f32_16x16x32_bf16 (A held)" : which == 2 ? "v_mfma_f32_16x16x32_bf16 (B held, s_nop between)" : "v_mfma_f32_16x16x32_bf16 (16 accumulators)",
           zeros ? "zeros" : "random", wps, flops / dt / 1e12, dt);
    fflush(stdout);
  }
  return 0;
}
