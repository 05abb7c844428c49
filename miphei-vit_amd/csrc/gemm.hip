// Dispatcher of the bf16 MFMA GEMM / implicit-GEMM convolution (kernel in gemm_kernel.hpp).
#include <cstdlib>
#include "gemm_kernel.hpp"

static int dispatch(const mvit_gemm_args& a, hipStream_t s);

namespace mvit_gemm {
int gemm_num_cus() { return mvit_num_cus(); }
bool ws_supported(const mvit_gemm_args& a);           // gemm_ws.hip: the wave-specialised 256x128 kernel (round 4)
int launch_ws(const mvit_gemm_args& a, hipStream_t s, int band_knob, int resid_single_knob, int dsw_reg_knob, int pack_store_knob);
bool ws4_supported(const mvit_gemm_args& a);          // gemm_ws4.hip: the same pipeline with one 128x64 consumer wave per SIMD (plain bf16 stores)
int launch_ws4(const mvit_gemm_args& a, hipStream_t s);
}  // namespace mvit_gemm

extern "C" MVIT_API int mvit_gemm_bf16(const mvit_gemm_args* args, mvit_stream_t stream) {
  MVIT_CLEAR_ERROR();
  using namespace mvit_gemm;
  if (!args) return MVIT_EINVAL;
  const mvit_gemm_args& a = *args;
  if (a.M <= 0 || a.N <= 0 || a.K <= 0) return MVIT_EINVAL;
  if ((a.K & 7) || (a.ldb & 7)) return MVIT_EINVAL;
  if (a.amode == MVIT_A_DENSE && (a.lda & 7)) return MVIT_EINVAL;
  if (a.amode == MVIT_A_PATCH) {
    if (a.conv_C != 8 || a.conv_ld != 8 || a.conv_stride <= 0 || a.K != a.conv_stride * a.conv_stride * 8 || a.A2 ||
        a.epi != MVIT_EPI_PATCH || a.conv_OH * a.conv_stride > a.conv_H || a.conv_OW * a.conv_stride > a.conv_W)
      return MVIT_EINVAL;
  } else if (a.amode != MVIT_A_DENSE && ((a.conv_C & 7) || (a.conv_ld & 7) || a.K != 9 * a.conv_C || a.A2)) {
    return MVIT_EINVAL;
  }
  if (a.A2 && ((a.K2 & 7) || (a.lda2 & 7) || (a.ldb2 & 7) || !a.B2)) return MVIT_EINVAL;
  if (a.ksplit > 1 && !(a.flags & MVIT_ATOMIC)) return MVIT_EINVAL;
  if ((a.flags & MVIT_RELU) && (a.amode != MVIT_A_CONV3 || a.epi != MVIT_EPI_STORE || (a.flags & (MVIT_OUT_F32 | MVIT_ATOMIC | MVIT_ACCUM_BF16))))
    return MVIT_EINVAL;
  if (a.epi == MVIT_EPI_STATS && (!a.stats || a.nslots <= 0)) return MVIT_EINVAL;
  if ((a.epi == MVIT_EPI_DSWIGLU || a.epi == MVIT_EPI_DGELU) && !a.aux) return MVIT_EINVAL;
  if (a.epi == MVIT_EPI_PATCH && (!a.pos || a.patch_P <= 0)) return MVIT_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  mvit_gemm_args al = a;  // vector (8/16-byte) epilogue I/O needs aligned pointers and leading dimensions
  {
    auto mis = [](const void* q, int ld) { return q && ((((uintptr_t)q) & 15) || (ld & 7)); };
    if (mis(a.C, a.ldc) || mis(a.aux, a.ldaux) || mis(a.bias, 0) || mis(a.gamma, 0) || mis(a.pos, a.N)) al.flags |= 0x400;
    if (a.epi == MVIT_EPI_SWIGLU && (a.ldc & 7)) al.flags |= 0x400;
  }
  return dispatch(al, s);
}

// Measurement knobs of the tile dispatch.  The product build compiles them to their constants: which kernel runs depends only on
// the problem.  `make DEBUG_KNOBS=1` (-DMVIT_DEBUG_KNOBS) reads them from the environment for A/B runs (tools/).
#ifdef MVIT_DEBUG_KNOBS
#define MVIT_KNOB(var, env, dflt) static const int var = [] { const char* e = getenv(env); return e ? atoi(e) : (dflt); }()
#else
#define MVIT_KNOB(var, env, dflt) constexpr int var = (dflt)
#endif

// tile variant for a problem: (BM << 20) | (BN << 8) | (WAVES_M << 4) | WAVES_N, or -1
static int select_variant(const mvit_gemm_args& a) {
  const bool dense = a.amode == MVIT_A_DENSE;
  MVIT_KNOB(big_tile, "MVIT_GEMM_BIG_TILE", 1);
  const bool big = big_tile && a.M >= 1024;  // 8-wave 256x128 tile, 3-stage DMA pipeline
  // 8-wave 256x256 tile (2 stages): 1.5x the arithmetic intensity per DMA'd byte, but its two stages leave the refill less
  // than one K step of lead and M = 5264 quantises badly on it; measured inside the model it wins from about 2.6 rounds of
  // tiles over the chip: fc1 of the batch-16 training step (672 tiles: +0.45 % on the step, same box, two pairs of runs) and
  // everything at batch-64 inference (+0.6 %); with qkv (378 tiles) and dfc2 (336) on it the step is 2 % slower
  MVIT_KNOB(huge_min_tiles, "MVIT_GEMM_HUGE_MIN_TILES", 600);
  const long long tiles256 = (long long)((a.M + 255) / 256) * ((a.N + 255) / 256);
  const bool huge = big && dense && (a.N % 256 == 0) && tiles256 >= huge_min_tiles && a.ksplit <= 1;
  // one-wave-per-SIMD variants (4 waves, 128-row sub-tiles), kept for measurement: MVIT_GEMM_W4 bit 0 sends the 256x256
  // problems there, bit 1 every dense 256x128 problem, bit 3 (8) the long-K (>= 4096) ones.  Off by default: since the
  // 8-wave tiles run the same explicitly ordered, register-double-buffered K step they are as fast in the main loop and
  // cheaper in the epilogue (tools/bench_vs_blas.py; whole step 41.3 vs 42.1 ms).
  // (the 4-wave instantiations -- gemm_dense_w4.hip -- are only compiled into the measurement library, `make dbg`)
  auto id = [](int bm, int bn, int wm, int wn) { return (bm << 20) | (bn << 8) | (wm << 4) | wn; };
#ifdef MVIT_DEBUG_KNOBS
  MVIT_KNOB(w4, "MVIT_GEMM_W4", 0);
  if (huge && (w4 & 1)) return id(256, 256, 2, 2);
  if (big && dense && (a.N % 128 == 0) && (a.epi != MVIT_EPI_SWIGLU || (w4 & 16)) && a.ksplit <= 1 &&
      ((w4 & 2) || ((w4 & 8) && a.K >= 4096)))
    return id(256, 128, 2, 2);
#endif
  if (huge) return id(256, 256, 2, 4);
  if (a.epi == MVIT_EPI_SWIGLU) {
    if ((a.N % 128) || !dense) return -1;
    return big ? id(256, 128, 4, 2) : id(128, 128, 2, 2);
  }
  // N between 65 and 255 that is no multiple of 128 (the decoder's 72- and 176-channel dgrads, K = 288 ... 1584 at a million
  // rows): 256-row tiles even though up to 44 % of their columns are padding - half as many tiles, and a K loop of 5-25 steps
  // is mostly per-tile fill and epilogue (+0.3 % on the step)
  MVIT_KNOB(wide_min, "MVIT_GEMM_WIDE_MIN", 65);
  if (a.N % 128 == 0 || a.N >= 256 || (big && a.N >= wide_min)) return big ? id(256, 128, 4, 2) : id(128, 128, 2, 2);
  if (a.N > 32) return id(128, 64, 2, 2);
  return id(128, 32, 4, 1);
}

static bool takes_ws(const mvit_gemm_args& a, int v) {
  // Wave-specialised kernel (gemm_ws.hip: producer waves own the operand DMA, consumer waves never wait on the vector-memory counter):
  // takes the dense problems of the 8-wave 256-row tiles it supports.  MVIT_GEMM_WS = bit mask over epilogues (1 store, 2 SwiGLU,
  // 4 residual, 8 d(SwiGLU)); which kernel runs depends only on the problem.
  MVIT_KNOB(ws_mask, "MVIT_GEMM_WS", 15);
  auto id = [](int bm, int bn, int wm, int wn) { return (bm << 20) | (bn << 8) | (wm << 4) | wn; };
  if (a.amode != MVIT_A_DENSE || (v != id(256, 128, 4, 2) && v != id(256, 256, 2, 4)) || !mvit_gemm::ws_supported(a)) return false;
  const int bit = a.epi == MVIT_EPI_STORE ? 1 : a.epi == MVIT_EPI_SWIGLU ? 2 : a.epi == MVIT_EPI_RESID ? 4 : 8;
  return (ws_mask & bit) != 0;
}


extern "C" MVIT_API int mvit_gemm_variant(const mvit_gemm_args* args) {
  if (!args || args->M <= 0 || args->N <= 0 || args->K <= 0) return MVIT_EINVAL;
  const int v = select_variant(*args);
  if (v < 0) return v;
  // bit 30: the wave-specialised kernel runs this problem (its tile is 256x128, 8 MFMA waves as 4 x 2, plus 4 DMA waves)
  return takes_ws(*args, v) ? (((256 << 20) | (128 << 8) | (4 << 4) | 2) | (1 << 30)) : v;
}

static int dispatch(const mvit_gemm_args& a, hipStream_t s) {
  using namespace mvit_gemm;
  const bool dense = a.amode == MVIT_A_DENSE;
  auto id = [](int bm, int bn, int wm, int wn) { return (bm << 20) | (bn << 8) | (wm << 4) | wn; };
  const int v = select_variant(a);
  if (v < 0) return MVIT_EINVAL;
  // (Measured and dropped in round 4: a ragged-M split -- rows [0, 5120) of fc1 as exactly five rounds of 256-row tiles on this kernel,
  // the remaining 144 rows as a second launch on the 128-row tiles -- 137.8 vs 133-135 us: the small launch costs what the sixth,
  // three-quarters-empty round costs.)
  MVIT_KNOB(ws_band, "MVIT_GEMM_WS_BAND", 1);     // 0: the ragged last tile row always as whole 256-row tiles; 2: band items spread evenly over the XCDs (measurement)
  MVIT_KNOB(ws_rsingle, "MVIT_GEMM_WS_RSINGLE", 1);   // 0: one-round residual GEMMs keep the consumer-side epilogue (measurement)
  MVIT_KNOB(ws_dswreg, "MVIT_GEMM_WS_DSWREG", 1);     // 0: the d(SwiGLU) operand as DMA'd pseudo tiles (two steps ahead) instead of the register prefetch
  MVIT_KNOB(ws_packst, "MVIT_GEMM_WS_PACKST", 1);     // 0: plain bf16 stores through the f32 panel (measurement)
#ifdef MVIT_DEBUG_KNOBS
  // measurement library only (gemm_ws4.hip): one-wave-per-SIMD consumers for the plain bf16 store problems: bit 0 = every such
  // problem, bit 1 = only one-round launches (tiles <= CUs), bit 2 = only multi-round launches.  Measured in round 5: proj 26.2 vs
  // 24.2 us, dfc1 120.9 vs 116.7, qkv 72.9 vs 69.5; step 468.6 vs 474.3 tiles/s with every store problem on it (-1.2 %), -0.5 %
  // with either half: two consumer waves per SIMD stay
  MVIT_KNOB(ws4, "MVIT_GEMM_WS4", 0);
  if (ws4 && takes_ws(a, v) && ws4_supported(a)) {
    const long long tiles = (long long)((a.M + 255) / 256) * (a.N / 128);
    const bool one_round = tiles <= gemm_num_cus();
    if ((ws4 & 1) || ((ws4 & 2) && one_round) || ((ws4 & 4) && !one_round)) return launch_ws4(a, s);
  }
#endif
  if (takes_ws(a, v)) return launch_ws(a, s, ws_band, ws_rsingle, ws_dswreg, ws_packst);
#ifdef MVIT_DEBUG_KNOBS
  if (v == id(256, 256, 2, 2)) return launch_dense<256, 256, 2, 2>(a, s);
  if (v == id(256, 128, 2, 2)) return launch_dense<256, 128, 2, 2>(a, s);
#endif
  if (v == id(256, 256, 2, 4)) return launch_dense<256, 256, 2, 4>(a, s);
  if (v == id(256, 128, 4, 2)) return dense ? launch_dense<256, 128, 4, 2>(a, s) : launch_conv<256, 128, 4, 2>(a, s);
  if (v == id(128, 128, 2, 2)) return dense ? launch_dense<128, 128, 2, 2>(a, s) : launch_conv<128, 128, 2, 2>(a, s);
  if (v == id(128, 64, 2, 2)) return dense ? launch_dense<128, 64, 2, 2>(a, s) : launch_conv<128, 64, 2, 2>(a, s);
  return dense ? launch_dense<128, 32, 4, 1>(a, s) : launch_conv<128, 32, 4, 1>(a, s);
}
