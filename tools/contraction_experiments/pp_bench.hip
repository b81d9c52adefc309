// The ping-pong contraction kernel (csrc/spq_gemm_pp.h) against the 128x128 kernel: bit-identity of the outputs on random operands
// and interleaved timing at a given shape (kernel tuning only).
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off [-DPP_DIAG=<bits>] [-DPP_NT=<4|6>] tools/contraction_experiments/pp_bench.hip -o tools/pp_bench
//   tools/pp_bench [M N K R [AL]]
#include <stdarg.h>
#include <vector>
#include <algorithm>
#include <random>
#include <string.h>
#include "../../llm-qat-on-gpt2_amd/csrc/spq_f16x2.hip"
namespace spq {
#include "spq_gemm_pp.h"
}
#ifndef PP_NT
#define PP_NT 6
#endif
namespace spq {
void set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fprintf(stderr, "\n"); }
int check_launch(const char* what) { hipError_t e = hipGetLastError(); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", what, hipGetErrorString(e)); return -2; } return 0; }
}
using namespace spq;
template <int AL> int run(int M, int N, int K, int R) {
  const int Mp = (M + 255) / 256 * 256, Np = (N + 127) / 128 * 128, Kp = (K + 63) / 64 * 64, Rp = (R + 63) / 64 * 64;
  std::mt19937 rng(1);
  auto fill = [&](size_t rows, size_t cols, size_t vrows, size_t vcols, int kind) {
    std::vector<_Float16> h(rows * cols, (_Float16)0.f);
    std::uniform_int_distribution<int> lv(-7, 7); std::normal_distribution<float> nd(0.f, 3000.f), lo(0.f, 1.5f);
    if (!getenv("PP_ZERO")) for (size_t r = 0; r < vrows; ++r) for (size_t c = 0; c < vcols; ++c)
      h[r * cols + c] = kind == 0 ? (_Float16)(float)lv(rng) : kind == 1 ? (_Float16)nd(rng) : (_Float16)lo(rng);
    _Float16* d; hipMalloc(&d, h.size() * 2); hipMemcpy(d, h.data(), h.size() * 2, hipMemcpyHostToDevice); return d;
  };
  GemmF16Args g; memset(&g, 0, sizeof g);
  g.qx = fill(Mp, Kp, M, K, AL == 2 ? 1 : 0); g.xl = AL == 2 ? fill(Mp, Kp, M, K, 2) : nullptr;
  if (R) { g.thi = fill(Mp, Rp, M, R, 1); g.tlo = fill(Mp, Rp, M, R, 2); g.Bhi = fill(Np, Rp, N, R, 1); g.Blo = fill(Np, Rp, N, R, 2); }
  g.Whi = fill(Np, Kp, N, K, 1); g.Wlo = fill(Np, Kp, N, K, 2);
  std::vector<float> hri(Mp), hrs(Np), hb(N), hxs = {4.f, 0.25f};
  std::uniform_real_distribution<float> u(0.5f, 2.f);
  for (auto& v : hri) v = u(rng); for (auto& v : hrs) v = u(rng) * 1e-4f; for (auto& v : hb) v = u(rng) - 1.f;
  float *ri, *rs, *bias, *y0, *y1, *xs;
  hipMalloc(&ri, Mp * 4); hipMalloc(&rs, Np * 4); hipMalloc(&bias, N * 4); hipMalloc(&xs, 8);
  hipMalloc(&y0, (size_t)M * N * 4); hipMalloc(&y1, (size_t)M * N * 4);
  hipMemcpy(ri, hri.data(), Mp * 4, hipMemcpyHostToDevice); hipMemcpy(rs, hrs.data(), Np * 4, hipMemcpyHostToDevice);
  hipMemcpy(bias, hb.data(), N * 4, hipMemcpyHostToDevice); hipMemcpy(xs, hxs.data(), 8, hipMemcpyHostToDevice);
  g.rowinv = ri; g.rowscale = rs; g.bias = bias; g.M = M; g.N = N; g.Kp = Kp; g.Rp = R ? Rp : 0;
  g.tiles_m = Mp / GM; g.tiles_n = Np / GN; g.xscale = xs; g.a_limbs = AL;
  unsigned long long* dbg = nullptr;
  hipMalloc(&dbg, 4096 * 8 * 16 * 8); hipMemset(dbg, 0, 4096 * 8 * 16 * 8); g.dbg = dbg;

  auto k0 = gemm_f16x2_t128_kernel<AL, 0>;
#ifdef PP_USE_PINGPONG
  auto k1 = gemm_f16x2_pp_kernel<PP_NT, AL, 0>;
  using C = PPCfg<PP_NT>;
  constexpr int K1_THREADS = 512;
#else
  auto k1 = gemm_f16x2_lc_kernel<PP_NT, AL, 0>;
  using C = LCCfg<PP_NT>;
  constexpr int K1_THREADS = 768;
#endif
  hipFuncSetAttribute((const void*)k0, hipFuncAttributeMaxDynamicSharedMemorySize, T128_LDS);
  hipFuncSetAttribute((const void*)k1, hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS);
  const int nt0 = 2 * g.tiles_m * g.tiles_n;
  const unsigned cap0 = T128_WGS * gemm_grid(1 << 30);
  const unsigned grid0 = (unsigned)nt0 < cap0 ? (unsigned)nt0 : cap0;
  const int nt1 = g.tiles_m * ((N + PPCfg<PP_NT>::BN - 1) / PPCfg<PP_NT>::BN);
  const unsigned grid1 = gemm_grid(nt1);
  printf("M=%d N=%d K=%d R=%d AL=%d | t128: %d tiles grid %u | pp NT=%d: %d tiles grid %u, LDS %d\n", M, N, K, R, AL, nt0, grid0, PP_NT, nt1, grid1, C::LDS);
  GemmF16Args g0 = g, g1 = g; g0.y = y0; g1.y = y1;
  hipMemset(y0, 0xff, (size_t)M * N * 4); hipMemset(y1, 0xee, (size_t)M * N * 4);
  k0<<<grid0, 256, T128_LDS>>>(g0);
  k1<<<grid1, K1_THREADS, C::LDS>>>(g1);
  if (hipDeviceSynchronize() != hipSuccess) { fprintf(stderr, "launch failed: %s\n", hipGetErrorString(hipGetLastError())); return 2; }
  int rc = 0;
  if (!(PP_DIAG & ~8)) {
    std::vector<float> a((size_t)M * N), b((size_t)M * N);
    hipMemcpy(a.data(), y0, a.size() * 4, hipMemcpyDeviceToHost); hipMemcpy(b.data(), y1, b.size() * 4, hipMemcpyDeviceToHost);
    size_t bad = 0, first = 0; double ss = 0;
    for (size_t i = 0; i < a.size(); ++i) { ss += (double)a[i] * a[i]; if (memcmp(&a[i], &b[i], 4)) { if (!bad) first = i; ++bad; } }
    printf("bit-identical: %s (%zu of %zu differ%s) rms %.4g\n", bad ? "NO" : "yes", bad, a.size(), bad ? "" : "", sqrt(ss / a.size()));
    if (bad) { printf("  first at m=%zu n=%zu: t128 %.9g pp %.9g\n", first / N, first % N, a[first], b[first]); rc = 1; }
    // run-to-run determinism of the new kernel
    hipMemset(y0, 0, (size_t)M * N * 4); g0.y = y0;
    GemmF16Args g2 = g; g2.y = y0;
    k1<<<grid1, K1_THREADS, C::LDS>>>(g2); hipDeviceSynchronize();
    hipMemcpy(a.data(), y0, a.size() * 4, hipMemcpyDeviceToHost);
    printf("run-to-run identical: %s\n", memcmp(a.data(), b.data(), a.size() * 4) ? "NO" : "yes");
    if (memcmp(a.data(), b.data(), a.size() * 4)) rc = 1;
  }
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const double flop = 2.0 * M * N * (double)K + (R ? 2.0 * M * (double)R * N : 0.0);
  for (int rep = 0; rep < 4; ++rep) {
    float ms0, ms1;
    for (int i = 0; i < 10; ++i) k0<<<grid0, 256, T128_LDS>>>(g0);
    hipEventRecord(e0); for (int i = 0; i < 50; ++i) k0<<<grid0, 256, T128_LDS>>>(g0); hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms0, e0, e1);
    for (int i = 0; i < 10; ++i) k1<<<grid1, K1_THREADS, C::LDS>>>(g1);
    hipEventRecord(e0); for (int i = 0; i < 50; ++i) k1<<<grid1, K1_THREADS, C::LDS>>>(g1); hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms1, e0, e1);
    printf("t128 %.1f us (%.0f TF)   pp %.1f us (%.0f TF)\n", ms0 / 50 * 1e3, flop / (ms0 / 50 * 1e-3) * 1e-12, ms1 / 50 * 1e3, flop / (ms1 / 50 * 1e-3) * 1e-12);
  }
  if (PP_DIAG & 32) {
    std::vector<unsigned long long> h((size_t)grid1 * 8 * 16);
    hipMemcpy(h.data(), dbg, h.size() * 8, hipMemcpyDeviceToHost);
    for (int grp = 0; grp < 2; ++grp) {
      double d[8] = {0}; const double nw = (double)grid1 * 4;
      for (unsigned b = 0; b < grid1; ++b) for (int w = 4 * grp; w < 4 * grp + 4; ++w) {
        const unsigned long long* o = &h[((size_t)b * 8 + w) * 16];
        d[0] += (double)(o[1] - o[0]); d[1] += (double)(o[2] - o[1]); d[2] += (double)(o[3] - o[2]); d[3] += (double)(o[4] - o[3]);
        d[4] += (double)(o[5] - o[4]); d[5] += (double)(o[12] - o[5]); d[6] += (double)(o[12] - o[0]);
      }
      if (PP_DIAG & 8) { double a = 0, b = 0; for (unsigned bb = 0; bb < grid1; ++bb) for (int w = 4 * grp; w < 4 * grp + 4; ++w) { a += (double)h[((size_t)bb * 8 + w) * 16 + 8]; b += (double)h[((size_t)bb * 8 + w) * 16 + 9]; }
        const double st = (double)nt1 / grid1 * ((R ? Rp / 64 * 4 : 0) + (AL == 1 ? Kp / 32 : Kp / 64 * 4));
        printf("group %d per stage: at the barrier %.0f | barrier -> MFMAs done %.0f\n", grp, a / nw / st, b / nw / st); }
      printf("group %d coarse stamps (cycles per wave): prologue %.0f | tile 0 stages %.0f, epilogue %.0f | tile 1 stages %.0f, epilogue %.0f | exit %.0f || total %.0f\n", grp,
             d[0] / nw, d[1] / nw, d[2] / nw, d[3] / nw, d[4] / nw, d[5] / nw, d[6] / nw);
    }
  }
#ifndef PP_USE_PINGPONG
  if (PP_DIAG & 32) {
    std::vector<unsigned long long> h((size_t)grid1 * 12 * 16);
    hipMemcpy(h.data(), dbg, h.size() * 8, hipMemcpyDeviceToHost);
    double d[3] = {0, 0, 0};
    for (unsigned b = 0; b < grid1; ++b) for (int lw = 0; lw < 4; ++lw) for (int i = 0; i < 3; ++i) d[i] += (double)h[((size_t)grid1 * 8 + (size_t)b * 4 + lw) * 16 + i];
    const double nl = (double)grid1 * 4, halves = 1.0 * (double)nt1 / grid1 * ((R ? Rp / 64 * 4 : 0) + (AL == 1 ? Kp / 32 : Kp / 64 * 4));
    printf("loader waves (cycles per stage): waiting for a free slot %.0f | issuing 2 x %d pieces %.0f | landing wait + FULL %.0f\n", d[0] / nl / halves,
           PPCfg<PP_NT>::P, d[1] / nl / halves, d[2] / nl / halves);
  }
#endif
  if (PP_DIAG & 8) {
    std::vector<unsigned long long> h((size_t)grid1 * 8 * 16);
    hipMemcpy(h.data(), dbg, h.size() * 8, hipMemcpyDeviceToHost);
    for (int grp = 0; grp < 2; ++grp) {
      double sum[16] = {0};
      for (unsigned b = 0; b < grid1; ++b) for (int w = 4 * grp; w < 4 * grp + 4; ++w) for (int i = 0; i < 16; ++i) sum[i] += (double)h[((size_t)b * 8 + w) * 16 + i];
      const double nw = (double)grid1 * 4, tiles = (double)nt1 / grid1;
      const int T = (R ? Rp / 64 * 4 : 0) + (AL == 1 ? Kp / 32 : Kp / 64 * 4);
      const double stages = tiles * T;
      printf("group %d stamps (cycles per stage and wave; %.1f stages): LOAD %.0f | barrier-1 %.0f | MFMA %.0f | barrier-2 %.0f || wave %.0f cycles, outside stages per tile %.0f, clock %.0f MHz\n",
             grp, stages, sum[0] / nw / stages, sum[1] / nw / stages, sum[2] / nw / stages, sum[3] / nw / stages, sum[4] / nw,
             (sum[4] - sum[0] - sum[1] - sum[2] - sum[3]) / nw / tiles, sum[4] / sum[5] * 100.0);
      printf("   LOAD = fragment-read issue %.0f + wait for last phase's copies %.0f + copy issue %.0f + lgkmcnt(0) %.0f\n", sum[8] / nw / stages, sum[9] / nw / stages,
             sum[10] / nw / stages, sum[11] / nw / stages);
      printf("   per tile: epilogue operand loads %.0f, whole epilogue %.0f, deferred barrier %.0f\n", sum[12] / nw / tiles, sum[13] / nw / tiles, sum[14] / nw / tiles);
    }
  }
  return rc;
}
int main(int argc, char** argv) {
  int M = 8192, N = 3072, K = 768, R = 64, AL = 1;
  if (argc > 4) { M = atoi(argv[1]); N = atoi(argv[2]); K = atoi(argv[3]); R = atoi(argv[4]); }
  if (argc > 5) AL = atoi(argv[5]);
  int rc = AL == 2 ? run<2>(M, N, K, R) : run<1>(M, N, K, R);
  return (hipGetLastError() == hipSuccess) ? rc : 3;
}
