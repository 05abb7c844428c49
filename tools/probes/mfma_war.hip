// Probe (not part of the library): does a VALU write to a source register of an MFMA issued just before it corrupt that MFMA?
// Found in round 4 in attn_fwd_kernel (three waves per SIMD): v_cvt_pk_bf16_f32 into the B operand registers three instructions behind
// the v_mfma_f32_32x32x16_bf16 reading them produced a wrong product about once in 10^7 MFMAs; hipcc inserts no wait states there.
// Every wave runs, in one assembly block with fixed registers,
//     B <- X;  loop { acc = mfma(A, B, acc); GAP x s_nop;  B <- Y (4 x v_mov_b32);  acc = mfma(A, B, acc); GAP x s_nop;  B <- X }
// so acc must equal iters * (A.X + A.Y) bit for bit, whatever GAP is; the result is compared with a 16-state gap.  BLOCKS_PER_CU
// 256-thread blocks per CU (1..3 waves per SIMD queueing on the matrix pipe).
//   hipcc -O3 --offload-arch=gfx950 tools/probes/mfma_war.hip -o tools/probes/mfma_war && tools/probes/mfma_war
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define NOPS_0
#define NOPS_1 "s_nop 0\n\t"
#define NOPS_2 "s_nop 1\n\t"
#define NOPS_4 "s_nop 3\n\t"
#define NOPS_8 "s_nop 7\n\t"
#define NOPS_16 "s_nop 7\n\ts_nop 7\n\t"

#define WAR_KERNEL(NAME, MFMA, ACCHI, NOPS)                                                                                            \
  __global__ __launch_bounds__(256) void NAME(const uint4* __restrict__ src, uint4* __restrict__ out, int iters) {                     \
    const uint4* pa = src + (threadIdx.x & 63);                                                                                        \
    const uint4* px = pa + 64;                                                                                                         \
    const uint4* py = pa + 128;                                                                                                        \
    uint4* po = out + ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;                                                                    \
    asm volatile(                                                                                                                      \
        "global_load_dwordx4 v[4:7], %[pa], off\n\t"                                                                                   \
        "global_load_dwordx4 v[8:11], %[px], off\n\t"                                                                                  \
        "global_load_dwordx4 v[12:15], %[py], off\n\t"                                                                                 \
        "s_waitcnt vmcnt(0)\n\t"                                                                                                       \
        "v_mov_b32 v20, 0\n\tv_mov_b32 v21, 0\n\tv_mov_b32 v22, 0\n\tv_mov_b32 v23, 0\n\t"                                             \
        "v_mov_b32 v24, 0\n\tv_mov_b32 v25, 0\n\tv_mov_b32 v26, 0\n\tv_mov_b32 v27, 0\n\t"                                             \
        "v_mov_b32 v28, 0\n\tv_mov_b32 v29, 0\n\tv_mov_b32 v30, 0\n\tv_mov_b32 v31, 0\n\t"                                             \
        "v_mov_b32 v32, 0\n\tv_mov_b32 v33, 0\n\tv_mov_b32 v34, 0\n\tv_mov_b32 v35, 0\n\t"                                             \
        "v_mov_b32 v16, v8\n\tv_mov_b32 v17, v9\n\tv_mov_b32 v18, v10\n\tv_mov_b32 v19, v11\n\t"                                       \
        "s_mov_b32 s20, %[n]\n\t"                                                                                                      \
        "s_nop 7\n\t"                                                                                                                  \
        "1:\n\t" MFMA " v[20:" ACCHI "], v[4:7], v[16:19], v[20:" ACCHI "]\n\t" NOPS                                                   \
        "v_mov_b32 v16, v12\n\tv_mov_b32 v17, v13\n\tv_mov_b32 v18, v14\n\tv_mov_b32 v19, v15\n\t"                                     \
        "s_nop 7\n\t" MFMA " v[20:" ACCHI "], v[4:7], v[16:19], v[20:" ACCHI "]\n\t" NOPS                                              \
        "v_mov_b32 v16, v8\n\tv_mov_b32 v17, v9\n\tv_mov_b32 v18, v10\n\tv_mov_b32 v19, v11\n\t"                                       \
        "s_nop 7\n\t"                                                                                                                  \
        "s_sub_u32 s20, s20, 1\n\t"                                                                                                    \
        "s_cmp_lg_u32 s20, 0\n\t"                                                                                                      \
        "s_cbranch_scc1 1b\n\t"                                                                                                        \
        "s_nop 7\n\ts_nop 7\n\ts_nop 7\n\t"                                                                                            \
        "global_store_dwordx4 %[po], v[20:23], off\n\t"                                                                                \
        "global_store_dwordx4 %[po], v[24:27], off offset:16\n\t"                                                                      \
        "global_store_dwordx4 %[po], v[28:31], off offset:32\n\t"                                                                      \
        "global_store_dwordx4 %[po], v[32:35], off offset:48\n\t"                                                                      \
        "s_waitcnt vmcnt(0)\n\t"                                                                                                       \
        :                                                                                                                              \
        : [pa] "v"(pa), [px] "v"(px), [py] "v"(py), [po] "v"(po), [n] "s"(iters)                                                       \
        : "memory", "s20", "scc", "v4", "v5", "v6", "v7", "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18",   \
          "v19", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35");       \
  }

WAR_KERNEL(war32_g0, "v_mfma_f32_32x32x16_bf16", "35", NOPS_0)
WAR_KERNEL(war32_g1, "v_mfma_f32_32x32x16_bf16", "35", NOPS_1)
WAR_KERNEL(war32_g2, "v_mfma_f32_32x32x16_bf16", "35", NOPS_2)
WAR_KERNEL(war32_g4, "v_mfma_f32_32x32x16_bf16", "35", NOPS_4)
WAR_KERNEL(war32_g8, "v_mfma_f32_32x32x16_bf16", "35", NOPS_8)
WAR_KERNEL(war32_g16, "v_mfma_f32_32x32x16_bf16", "35", NOPS_16)
WAR_KERNEL(war16_g0, "v_mfma_f32_16x16x32_bf16", "23", NOPS_0)
WAR_KERNEL(war16_g1, "v_mfma_f32_16x16x32_bf16", "23", NOPS_1)
WAR_KERNEL(war16_g2, "v_mfma_f32_16x16x32_bf16", "23", NOPS_2)
WAR_KERNEL(war16_g4, "v_mfma_f32_16x16x32_bf16", "23", NOPS_4)
WAR_KERNEL(war16_g16, "v_mfma_f32_16x16x32_bf16", "23", NOPS_16)


// LDS-return variant: two MFMAs back to back (so that the second can queue behind the first and behind the other waves' MFMAs), then a
// ds_read_b128 that reloads the SECOND MFMA's B registers with the other value; lgkmcnt(0) before the registers are used again.
#define WARL_KERNEL(NAME, NOPS)                                                                                                        \
  __global__ __launch_bounds__(256) void NAME(const uint4* __restrict__ src, uint4* __restrict__ out, int iters) {                     \
    __shared__ uint4 lds[2 * 64 * 4];                                                                                                  \
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;                                                                        \
    lds[(wave * 2 + 0) * 64 + lane] = src[64 + lane];                                                                                  \
    lds[(wave * 2 + 1) * 64 + lane] = src[128 + lane];                                                                                 \
    __syncthreads();                                                                                                                   \
    const uint4* pa = src + lane;                                                                                                      \
    const unsigned lx = (unsigned)(size_t)(__attribute__((address_space(3))) uint4*)&lds[(wave * 2 + 0) * 64 + lane];                  \
    const unsigned ly = (unsigned)(size_t)(__attribute__((address_space(3))) uint4*)&lds[(wave * 2 + 1) * 64 + lane];                  \
    uint4* po = out + ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;                                                                    \
    asm volatile(                                                                                                                      \
        "global_load_dwordx4 v[4:7], %[pa], off\n\t"                                                                                   \
        "ds_read_b128 v[8:11], %[lx]\n\t"                                                                                              \
        "ds_read_b128 v[16:19], %[lx]\n\t"                                                                                             \
        "s_waitcnt vmcnt(0) lgkmcnt(0)\n\t"                                                                                            \
        "v_mov_b32 v20, 0\n\tv_mov_b32 v21, 0\n\tv_mov_b32 v22, 0\n\tv_mov_b32 v23, 0\n\t"                                             \
        "v_mov_b32 v24, 0\n\tv_mov_b32 v25, 0\n\tv_mov_b32 v26, 0\n\tv_mov_b32 v27, 0\n\t"                                             \
        "v_mov_b32 v28, 0\n\tv_mov_b32 v29, 0\n\tv_mov_b32 v30, 0\n\tv_mov_b32 v31, 0\n\t"                                             \
        "v_mov_b32 v32, 0\n\tv_mov_b32 v33, 0\n\tv_mov_b32 v34, 0\n\tv_mov_b32 v35, 0\n\t"                                             \
        "v_mov_b32 v36, 0\n\tv_mov_b32 v37, 0\n\tv_mov_b32 v38, 0\n\tv_mov_b32 v39, 0\n\t"                                             \
        "v_mov_b32 v40, 0\n\tv_mov_b32 v41, 0\n\tv_mov_b32 v42, 0\n\tv_mov_b32 v43, 0\n\t"                                             \
        "v_mov_b32 v44, 0\n\tv_mov_b32 v45, 0\n\tv_mov_b32 v46, 0\n\tv_mov_b32 v47, 0\n\t"                                             \
        "v_mov_b32 v48, 0\n\tv_mov_b32 v49, 0\n\tv_mov_b32 v50, 0\n\tv_mov_b32 v51, 0\n\t"                                             \
        "s_mov_b32 s20, %[n]\n\t"                                                                                                      \
        "s_nop 7\n\t"                                                                                                                  \
        "1:\n\t"                                                                                                                       \
        "v_mfma_f32_32x32x16_bf16 v[36:51], v[4:7], v[8:11], v[36:51]\n\t"                                                             \
        "v_mfma_f32_32x32x16_bf16 v[20:35], v[4:7], v[16:19], v[20:35]\n\t" NOPS                                                       \
        "ds_read_b128 v[16:19], %[ly]\n\t"                                                                                             \
        "s_waitcnt lgkmcnt(0)\n\t"                                                                                                     \
        "s_nop 1\n\t"                                                                                                                  \
        "v_mfma_f32_32x32x16_bf16 v[36:51], v[4:7], v[8:11], v[36:51]\n\t"                                                             \
        "v_mfma_f32_32x32x16_bf16 v[20:35], v[4:7], v[16:19], v[20:35]\n\t" NOPS                                                       \
        "ds_read_b128 v[16:19], %[lx]\n\t"                                                                                             \
        "s_waitcnt lgkmcnt(0)\n\t"                                                                                                     \
        "s_nop 1\n\t"                                                                                                                  \
        "s_sub_u32 s20, s20, 1\n\t"                                                                                                    \
        "s_cmp_lg_u32 s20, 0\n\t"                                                                                                      \
        "s_cbranch_scc1 1b\n\t"                                                                                                        \
        "s_nop 7\n\ts_nop 7\n\ts_nop 7\n\t"                                                                                            \
        "global_store_dwordx4 %[po], v[20:23], off\n\t"                                                                                \
        "global_store_dwordx4 %[po], v[24:27], off offset:16\n\t"                                                                      \
        "global_store_dwordx4 %[po], v[28:31], off offset:32\n\t"                                                                      \
        "global_store_dwordx4 %[po], v[32:35], off offset:48\n\t"                                                                      \
        "s_waitcnt vmcnt(0)\n\t"                                                                                                       \
        :                                                                                                                              \
        : [pa] "v"(pa), [lx] "v"(lx), [ly] "v"(ly), [po] "v"(po), [n] "s"(iters)                                                       \
        : "memory", "s20", "scc", "v4", "v5", "v6", "v7", "v8", "v9", "v10", "v11", "v16", "v17", "v18", "v19", "v20", "v21", "v22",   \
          "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39",       \
          "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51");                                         \
  }
WARL_KERNEL(warl_g0, NOPS_0)
WARL_KERNEL(warl_g2, NOPS_2)
WARL_KERNEL(warl_g8, NOPS_8)
WARL_KERNEL(warl_g16, NOPS_16 NOPS_16 NOPS_16 NOPS_16)

static uint16_t f2bf(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 20000;
  std::vector<uint16_t> h(3 * 64 * 8);
  uint32_t s = 7;
  for (auto& v : h) {
    s = s * 1664525u + 1013904223u;
    v = f2bf(((int)(s >> 9) % 2001 - 1000) / 1000.f);
  }
  uint4 *dsrc, *dout, *dref;
  hipMalloc(&dsrc, h.size() * 2);
  hipMemcpy(dsrc, h.data(), h.size() * 2, hipMemcpyHostToDevice);
  typedef void (*kern_t)(const uint4*, uint4*, int);
  struct { const char* name; kern_t k; kern_t ref; int nacc; } tests[] = {
      {"32x32x16 gap 0", war32_g0, war32_g16, 16}, {"32x32x16 gap 1", war32_g1, war32_g16, 16}, {"32x32x16 gap 2", war32_g2, war32_g16, 16},
      {"32x32x16 gap 4", war32_g4, war32_g16, 16}, {"32x32x16 gap 8", war32_g8, war32_g16, 16},
      {"16x16x32 gap 0", war16_g0, war16_g16, 4},  {"16x16x32 gap 1", war16_g1, war16_g16, 4},  {"16x16x32 gap 2", war16_g2, war16_g16, 4},
      {"16x16x32 gap 4", war16_g4, war16_g16, 4},
      {"LDS reload gap 0", warl_g0, warl_g16, 16}, {"LDS reload gap 2", warl_g2, warl_g16, 16}, {"LDS reload gap 8", warl_g8, warl_g16, 16}};
  for (int per_cu = 1; per_cu <= 3; per_cu += 2) {
    const int blocks = 256 * per_cu;
    const size_t n16 = (size_t)blocks * 256 * 4;
    hipMalloc(&dout, n16 * 16);
    hipMalloc(&dref, n16 * 16);
    std::vector<uint32_t> ho(n16 * 4), hr(n16 * 4);
    for (auto& t : tests) {
      hipMemset(dout, 0, n16 * 16);
      hipMemset(dref, 0, n16 * 16);
      hipLaunchKernelGGL(t.ref, dim3(blocks), dim3(256), 0, 0, dsrc, dref, iters);
      long bad_total = 0, runs = 20;
      hipMemcpy(hr.data(), dref, n16 * 16, hipMemcpyDeviceToHost);
      for (int r = 0; r < runs; ++r) {
        hipLaunchKernelGGL(t.k, dim3(blocks), dim3(256), 0, 0, dsrc, dout, iters);
        hipMemcpy(ho.data(), dout, n16 * 16, hipMemcpyDeviceToHost);
        long bad = 0;
        for (size_t thr = 0; thr < (size_t)blocks * 256; ++thr)
          for (int e = 0; e < t.nacc; ++e)
            if (ho[thr * 16 + e] != hr[thr * 16 + e]) { ++bad; break; }
        bad_total += bad;
      }
      // reference must equal itself across waves too (every wave computes the same thing)
      long refdiff = 0;
      for (size_t thr = 64; thr < (size_t)blocks * 256; ++thr)
        for (int e = 0; e < t.nacc; ++e)
          if (hr[thr * 16 + e] != hr[(thr & 63) * 16 + e]) { ++refdiff; break; }
      printf("%d block(s)/CU  %-17s: %ld lanes with a wrong accumulator over %ld launches x %d waves x %d MFMAs (reference self-check: %ld)\n",
             per_cu, t.name, bad_total, runs, blocks * 4, 2 * iters, refdiff);
    }
    hipFree(dout);
    hipFree(dref);
  }
  return 0;
}
