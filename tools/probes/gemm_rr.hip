// Probe (not part of the library): C[M,N] = A[M,K] * B[N,K]^T in bf16 on a one-wave-per-SIMD 256x256x64 tile whose K tile is
// REGISTER RESIDENT: all fragments of a K tile (4 sub-steps x (4 A + 4 B) x 16 B per lane = 128 VGPRs) are read out of LDS
// early, which frees the A half of an LDS buffer a quarter into the K step and the B half at its middle, so two 64 KB
// buffers give the refill of tile t+2 more than one K step of lead (the library's two-stage 256x256 tile has < 1; see
// DESIGN.md section 6, "What the K step costs").  Instruction order is explicit, as in gemm_kernel.hpp.
// Per K tile t (buffer t&1), fragments of sub-steps 0/1 already in registers:
//   phase 1: 16 MFMAs (s=0) | read A fragments of s=2,3            -> lgkmcnt(0), barrier  (A half of the buffer is free)
//   phase 2: 16 MFMAs (s=1) | read B fragments of s=2,3, DMA A(t+2) -> lgkmcnt(0), barrier  (B half is free)
//   phase 3: 16 MFMAs (s=2) | DMA B(t+2)                            -> vmcnt(16), barrier   (tile t+1 has landed)
//   phase 4: 16 MFMAs (s=3) | read fragments s=0,1 of tile t+1
// Measured (8192^3, same box): 856 us = 1.28 PFLOP/s with a plain epilogue and no prologue overlap, against 844 us for the
// library's 8-wave two-stage 256x256 tile and 677 us for hipBLASLt: the pipeline shape alone does not close the gap (bunching
// the fragment reads into the first half of a phase is 6 % slower than one read per two MFMAs; a v_mfma_f32_16x16x32_bf16
// build of the same loop, with this swizzle, ran at 1.0 PFLOP/s).
// Requires M, N multiples of 256 and K a multiple of 64 (>= 192).  Build:
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/probes/gemm_rr.hip -o tools/probes/gemm_rr ; run: tools/probes/gemm_rr [M N K]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <type_traits>
#include <vector>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef uint16_t bf16_t;
typedef __attribute__((address_space(3))) void* lds_ptr;

constexpr int BM = 256, BN = 256, BK = 64;
constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, BUF_BYTES = A_BYTES + B_BYTES;

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* ptr) {
  const unsigned long long v = (unsigned long long)ptr;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
  return __builtin_amdgcn_make_buffer_rsrc((void*)(((unsigned long long)hi << 32) | lo), 0, 0x7fffffff, 0x00020000);
}
__device__ __forceinline__ unsigned pack2bf(float a, float b) {
  unsigned ua = __float_as_uint(a), ub = __float_as_uint(b);
  ua += 0x7fffu + ((ua >> 16) & 1u);
  ub += 0x7fffu + ((ub >> 16) & 1u);
  return (ua >> 16) | (ub & 0xffff0000u);
}

#define FENCE() __builtin_amdgcn_sched_barrier(0)
// MFMA order inside a sub-step: raster (i major, j minor), or with -DRR_SERP serpentine (j backwards on odd i): then every
// pair of consecutive MFMAs shares one operand fragment
#ifdef RR_SERP
#define JJ(m) ((((m) >> 2) & 1) ? 3 - ((m) & 3) : ((m) & 3))
#else
#define JJ(m) ((m) & 3)
#endif
// -DRR_TIMING: s_memtime stamps around the four MFMA phases and the three hand-overs of every K tile, summed per wave and
// written to `prof` by every wave of block 0 (7 sums + tile count); costs ~10 % itself, ratios are what matters
// -DRR_NO_READS / -DRR_NO_DMA: compile the fragment reads / the refill DMA out of the K loop (results are garbage): with
// RR_SECONDS=n (run back to back for n seconds) this gives each part's share of the power budget at the cap
#ifdef RR_NO_READS
#define RD(dst, src) asm volatile("" : "+v"(dst))
#else
#define RD(dst, src) dst = *(const bf16x8*)(src)
#endif
#ifdef RR_TIMING
#define STAMP(k) { const long long now_ = (long long)__builtin_readcyclecounter(); tsum[k] += now_ - tlast; tlast = now_; }
#else
#define STAMP(k)
#endif

template <int DUMMY>
__global__ __launch_bounds__(256) void gemm_rr_kernel(const bf16_t* __restrict__ A, const bf16_t* __restrict__ B,
                                                      bf16_t* __restrict__ C, int M, int N, int K, long long* __restrict__ prof) {
#ifdef RR_TIMING
  long long tsum[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = 0;
#endif
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const int wm = wave >> 1, wn = wave & 1;
  const int frag_row = lane & 31, half = lane >> 5;
  const int c8 = tid & 7, row_base = tid >> 3;
  const int c8s = c8 ^ ((row_base >> 1) & 7);
  const int tiles_m = M / BM, tiles_n = N / BN, ntiles = tiles_m * tiles_n, nk = K / BK;

  // fragment addresses: lane offset per sub-step; +i*4096 selects the 32-row fragment
  unsigned aoff[4], boff[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const unsigned sw = (unsigned)(((s * 2 + half) ^ ((frag_row >> 1) & 7)) << 4);
    aoff[s] = (unsigned)(wm * 128 + frag_row) * 128u + sw;
    boff[s] = (unsigned)A_BYTES + (unsigned)(wn * 128 + frag_row) * 128u + sw;
  }

  for (int vt = blockIdx.x; vt < ntiles; vt += gridDim.x) {
    int wg;
    {
      const int q = ntiles >> 3, r = ntiles & 7, xcd = vt & 7;
      wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (vt >> 3);
    }
    constexpr int GROUP_M = 4;
    const int per_group = GROUP_M * tiles_n;
    const int first_m = (wg / per_group) * GROUP_M;
    const int gsz = min(tiles_m - first_m, GROUP_M);
    const int tile_m = first_m + (wg % per_group) % gsz, tile_n = (wg % per_group) / gsz;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const __amdgpu_buffer_rsrc_t rsA = make_rsrc(A + (size_t)m0 * K), rsB = make_rsrc(B + (size_t)n0 * K);
    unsigned poff[8];  // this lane's source offset of piece j (rows row_base + 32 j), same for A and B (lda = ldb = K)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      poff[j] = ((unsigned)(row_base + 32 * j) * (unsigned)K + (unsigned)c8s * 8u) * 2u;
      asm volatile("" : "+v"(poff[j]));
    }
    auto dma = [&](int t, int buf, int p, auto) __attribute__((always_inline)) {  // piece p < 8: A rows, else B rows
      char* dst = smem + buf * BUF_BYTES + (p < 8 ? 0 : A_BYTES) + (wave_u * 8 + 32 * (p & 7)) * 128;
      const unsigned off = poff[p & 7];
      if (p < 8)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_ptr)dst, 16, off, t * (BK * 2), 0, 0);
      else
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lds_ptr)dst, 16, off, t * (BK * 2), 0, 0);
    };

    f32x16 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    bf16x8 fa[4][4], fb[4][4];  // [sub-step][fragment]

    // prologue: tiles 0 and 1 in flight, fragments of sub-steps 0, 1 of tile 0
#pragma unroll
    for (int p = 0; p < 16; ++p) dma(0, 0, p, 0);
#pragma unroll
    for (int p = 0; p < 16; ++p) dma(1, 1, p, 0);
    asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        fa[s][i] = *(const bf16x8*)(smem + aoff[s] + i * 4096);
        fb[s][i] = *(const bf16x8*)(smem + boff[s] + i * 4096);
      }

    auto ktile = [&](int t, auto refill_tag, auto next_tag) __attribute__((always_inline)) {
      constexpr bool REFILL = decltype(refill_tag)::value;  // tile t+2 exists
      constexpr bool NEXT = decltype(next_tag)::value;      // tile t+1 exists
      const int b = t & 1;
      const char* cur = smem + b * BUF_BYTES;
      const char* nxt = smem + (b ^ 1) * BUF_BYTES;
#ifdef RR_TIMING
      tlast = (long long)__builtin_readcyclecounter();
#endif
      // ---- phase 1: s = 0 | A fragments of s = 2, 3
#pragma unroll
      for (int m = 0; m < 16; ++m) {
        acc[m >> 2][JJ(m)] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0][m >> 2], fb[0][JJ(m)], acc[m >> 2][JJ(m)], 0, 0, 0);
        if ((m & 1) == 0) {  // one read per two MFMAs (bunching them into the first eight measured 6 % slower)
          const int r = m >> 1;  // 0..7
          RD(fa[2 + (r >> 2)][r & 3], cur + aoff[2 + (r >> 2)] + (r & 3) * 4096);
        }
        FENCE();
      }
      STAMP(0)
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      STAMP(1)
      FENCE();
      // ---- phase 2: s = 1 | B fragments of s = 2, 3 and the A half of tile t+2
#pragma unroll
      for (int m = 0; m < 16; ++m) {
        acc[m >> 2][JJ(m)] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[1][m >> 2], fb[1][JJ(m)], acc[m >> 2][JJ(m)], 0, 0, 0);
        if ((m & 1) == 0) {
          const int r = m >> 1;
          RD(fb[2 + (r >> 2)][r & 3], cur + boff[2 + (r >> 2)] + (r & 3) * 4096);
        } else if (REFILL) {
#ifndef RR_NO_DMA
          dma(t + 2, b, m >> 1, 0);
#endif
        }
        FENCE();
      }
      STAMP(2)
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      STAMP(3)
      FENCE();
      // ---- phase 3: s = 2 | the B half of tile t+2
#pragma unroll
      for (int m = 0; m < 16; ++m) {
        acc[m >> 2][JJ(m)] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[2][m >> 2], fb[2][JJ(m)], acc[m >> 2][JJ(m)], 0, 0, 0);
#ifndef RR_NO_DMA
        if (REFILL && (m & 1)) dma(t + 2, b, 8 + (m >> 1), 0);
#endif
        FENCE();
      }
      STAMP(4)
      if (NEXT) {
        if (REFILL)
          asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        else
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        FENCE();
      }
      STAMP(5)
      // ---- phase 4: s = 3 | fragments of s = 0, 1 of tile t+1 (order of use: A0, B0..B3, A1..A3)
#pragma unroll
      for (int m = 0; m < 16; ++m) {
        acc[m >> 2][JJ(m)] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[3][m >> 2], fb[3][JJ(m)], acc[m >> 2][JJ(m)], 0, 0, 0);
        if (NEXT) {
          const int s = m >> 3, r = m & 7;
          if (r == 0)
            RD(fa[s][0], nxt + aoff[s]);
          else if (r <= 4)
            RD(fb[s][r - 1], nxt + boff[s] + (r - 1) * 4096);
          else
            RD(fa[s][r - 4], nxt + aoff[s] + (r - 4) * 4096);
        }
        FENCE();
      }
      STAMP(6)
#ifdef RR_TIMING
      tsum[7] += 1;
#endif
    };
    int t = 0;
    for (; t + 2 < nk; ++t) ktile(t, std::true_type{}, std::true_type{});
    ktile(t++, std::false_type{}, std::true_type{});
    ktile(t++, std::false_type{}, std::false_type{});

    // ---- epilogue (plain): accumulators -> bf16 tile in LDS (row stride 512 B) -> 16-byte coalesced stores
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = wm * 128 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
          const int col = wn * 128 + j * 32 + frag_row;
          ((bf16_t*)smem)[row * 256 + col] = (bf16_t)(pack2bf(acc[i][j][r], 0.f) & 0xffffu);
        }
    __syncthreads();
#pragma unroll 4
    for (int c = tid; c < 256 * 32; c += 256) {
      const int row = c >> 5, ch = c & 31;
      *(uint4*)(C + (size_t)(m0 + row) * N + n0 + ch * 8) = *(const uint4*)(smem + row * 512 + ch * 16);
    }
    __syncthreads();
  }
#ifdef RR_TIMING
  if (blockIdx.x == 0 && lane == 0)
    for (int k = 0; k < 8; ++k) prof[wave * 8 + k] = tsum[k];
#endif
}

static uint16_t f2bf(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}
static float bf2f(uint16_t h) {
  uint32_t u = (uint32_t)h << 16;
  float f;
  memcpy(&f, &u, 4);
  return f;
}

int main(int argc, char** argv) {
  const int M = argc > 3 ? atoi(argv[1]) : 8192, N = argc > 3 ? atoi(argv[2]) : 8192, K = argc > 3 ? atoi(argv[3]) : 8192;
  auto kern = gemm_rr_kernel<0>;
  if (M % 256 || N % 256 || K % 64 || K < 192) { printf("need M,N %% 256 == 0, K %% 64 == 0, K >= 192\n"); return 1; }
  std::vector<uint16_t> hA((size_t)M * K), hB((size_t)N * K);
  uint32_t s = 12345;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((int)(s >> 9) % 2001 - 1000) / 1000.f; };
  const bool zeros = getenv("RR_ZEROS") != nullptr;  // all-zero operands: same instruction stream, far less switching power
  for (auto& v : hA) v = zeros ? 0 : f2bf(rnd());
  for (auto& v : hB) v = zeros ? 0 : f2bf(rnd());
  bf16_t *dA, *dB, *dC;
  hipMalloc(&dA, hA.size() * 2); hipMalloc(&dB, hB.size() * 2); hipMalloc(&dC, (size_t)M * N * 2);
  hipMemcpy(dA, hA.data(), hA.size() * 2, hipMemcpyHostToDevice);
  hipMemcpy(dB, hB.data(), hB.size() * 2, hipMemcpyHostToDevice);
  long long* dProf; hipMalloc(&dProf, 32 * 8); hipMemset(dProf, 0, 32 * 8);
  const int lds = 2 * BUF_BYTES;
  hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  const int ntiles = (M / 256) * (N / 256);
  const int grid = ntiles < 256 ? ntiles : 256;
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, 0, dA, dB, dC, M, N, K, dProf);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int it = 20;
  hipEventRecord(e0);
  for (int i = 0; i < it; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, 0, dA, dB, dC, M, N, K, dProf);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= it;
  if (const char* sec = getenv("RR_SECONDS")) {   // sustained rate at the power cap
    const double want = atof(sec);
    hipDeviceSynchronize();
    auto t0 = std::chrono::steady_clock::now();
    long n = 0; double dt = 0;
    while (dt < want) {
      for (int i = 0; i < 50; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, 0, dA, dB, dC, M, N, K, dProf);
      hipDeviceSynchronize(); n += 50;
      dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    }
    printf("sustained over %.1f s: %.1f TF/s\n", dt, 2.0 * M * N * K * n / dt / 1e12);
  }
  if (hipGetLastError() != hipSuccess) { printf("launch failed\n"); return 1; }
  std::vector<uint16_t> hC((size_t)M * N);
  hipMemcpy(hC.data(), dC, hC.size() * 2, hipMemcpyDeviceToHost);
  double worst = 0;
  for (int q = 0; q < 256; ++q) {
    s = s * 1664525u + 1013904223u; const int r = (s >> 8) % M;
    s = s * 1664525u + 1013904223u; const int c = (s >> 8) % N;
    double ref = 0;
    for (int k = 0; k < K; ++k) ref += (double)bf2f(hA[(size_t)r * K + k]) * bf2f(hB[(size_t)c * K + k]);
    const double got = bf2f(hC[(size_t)r * N + c]);
    const double err = fabs(got - ref) / (fabs(ref) + 1.0);
    if (err > worst) worst = err;
  }
  printf("gemm_rr M=%d N=%d K=%d: %.1f us  %.1f TF/s  worst sampled rel err %.2e %s\n", M, N, K, ms * 1e3,
         2.0 * M * N * K / (ms * 1e-3) / 1e12, worst, worst < 2e-2 ? "OK" : "MISMATCH");
#ifdef RR_TIMING
  long long hp[32]; hipMemcpy(hp, dProf, sizeof(hp), hipMemcpyDeviceToHost);
  const char* names[7] = {"phase1 MFMA+A reads", "hand-over 1 (lgkm, barrier)", "phase2 MFMA+B reads+DMA A", "hand-over 2 (lgkm, barrier)",
                          "phase3 MFMA+DMA B", "hand-over 3 (vmcnt, barrier)", "phase4 MFMA+next reads"};
  for (int w = 0; w < 4; ++w) {
    printf("wave %d (block 0), cycles per K tile over %lld tiles:", w, hp[w * 8 + 7]);
    long long tot = 0;
    for (int k = 0; k < 7; ++k) tot += hp[w * 8 + k];
    for (int k = 0; k < 7; ++k) printf("  %s %.0f", w == 0 ? names[k] : "", (double)hp[w * 8 + k] / hp[w * 8 + 7]);
    printf("  | total %.0f\n", (double)tot / hp[w * 8 + 7]);
  }
#endif
  return worst < 2e-2 ? 0 : 2;
}
