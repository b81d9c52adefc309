// Timing and bit-for-bit check of the 256x128 contraction kernel against the 128x128 one (kernel tuning only):
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off [-DT256_DIAG=<bits>] tools/t256_bench.hip -o <bin>;  <bin> [M N K R]
#include <stdarg.h>
#include <vector>
#include <random>
#include <string.h>
#include "../llm-qat-on-gpt2_amd/csrc/spq_f16x2.hip"
#include "t256_kernel.h"
namespace spq {
void set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fprintf(stderr, "\n"); }
int check_launch(const char* what) { hipError_t e = hipGetLastError(); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", what, hipGetErrorString(e)); return -2; } return 0; }
}
using namespace spq;
int main(int argc, char** argv) {
  const int M = argc > 1 ? atoi(argv[1]) : 8192, N = argc > 2 ? atoi(argv[2]) : 3072, K = argc > 3 ? atoi(argv[3]) : 768, R = argc > 4 ? atoi(argv[4]) : 64;
  const int Mp = (M + 255) / 256 * 256, Np = (N + 127) / 128 * 128;
  std::mt19937 rng(1);
  auto fill = [&](size_t n, int kind) {
    std::vector<_Float16> h(n);
    std::uniform_int_distribution<int> lv(-7, 7); std::normal_distribution<float> nd(0.f, 3000.f);
    for (auto& v : h) v = kind == 0 ? (_Float16)(float)lv(rng) : (_Float16)nd(rng);
    _Float16* d; (void)hipMalloc(&d, n * 2); (void)hipMemcpy(d, h.data(), n * 2, hipMemcpyHostToDevice); return d;
  };
  auto fillf = [&](size_t n, float lo, float hi) {
    std::vector<float> h(n); std::uniform_real_distribution<float> u(lo, hi);
    for (auto& v : h) v = u(rng);
    float* d; (void)hipMalloc(&d, n * 4); (void)hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice); return d;
  };
  GemmF16Args g;
  g.qx = fill((size_t)Mp * K, 0); g.thi = fill((size_t)Mp * R, 1); g.tlo = fill((size_t)Mp * R, 1);
  g.Whi = fill((size_t)Np * K, 1); g.Wlo = fill((size_t)Np * K, 1); g.Bhi = fill((size_t)Np * R, 1); g.Blo = fill((size_t)Np * R, 1);
  float *y0, *y1;
  (void)hipMalloc(&y0, (size_t)M * N * 4); (void)hipMalloc(&y1, (size_t)M * N * 4);
  g.rowinv = fillf(Mp, 0.5f, 2.f); g.rowscale = fillf(Np, 1e-4f, 2e-4f); g.bias = fillf(N, -1.f, 1.f); g.y = y0; g.M = M; g.N = N; g.Kp = K; g.Rp = R;
  g.tiles_m = Mp / GM; g.tiles_n = Np / GN; g.dbg = nullptr; g.xl = nullptr; g.xscale = nullptr; g.a_limbs = 1;
  unsigned long long* dbg = nullptr;
  if (T256_DIAG & 8) { (void)hipMalloc(&dbg, 4096 * 4 * 8 * 8); (void)hipMemset(dbg, 0, 4096 * 4 * 8 * 8); g.dbg = dbg; }
  auto k128 = gemm_f16x2_t128_kernel<1, 0>;
  auto k256 = gemm_f16x2_t256_kernel<1, 0>;
  (void)hipFuncSetAttribute((const void*)k128, hipFuncAttributeMaxDynamicSharedMemorySize, T128_LDS);
  (void)hipFuncSetAttribute((const void*)k256, hipFuncAttributeMaxDynamicSharedMemorySize, T256_LDS);
  const int cus = (int)gemm_grid(1 << 30);
  const int ntiles = 2 * g.tiles_m * g.tiles_n;
  const unsigned grid128 = (unsigned)std::min(ntiles, T128_WGS * cus);
  const int grid256 = std::min(2 * cus, g.tiles_m * g.tiles_n);
  const T256Plan pl = t256_plan(g.tiles_m, g.tiles_n, grid256);
  printf("M %d N %d K %d R %d: %d CUs; t256 plan: %d whole tiles (%d bands) + %d half tiles on %d workgroups\n", M, N, K, R, cus, pl.n_full, pl.full_bands, pl.n_half, pl.grid);
  // correctness: bit for bit against the 128x128 kernel
  g.y = y0; k128<<<grid128, 256, T128_LDS>>>(g);
  g.y = y1; (void)hipMemset(y1, 0xff, (size_t)M * N * 4); k256<<<grid256, 256, T256_LDS>>>(g, pl);
  if (hipDeviceSynchronize() != hipSuccess) { fprintf(stderr, "launch failed: %s\n", hipGetErrorString(hipGetLastError())); return 2; }
  if (!(T256_DIAG & 7)) {
    std::vector<float> h0((size_t)M * N), h1((size_t)M * N);
    (void)hipMemcpy(h0.data(), y0, h0.size() * 4, hipMemcpyDeviceToHost); (void)hipMemcpy(h1.data(), y1, h1.size() * 4, hipMemcpyDeviceToHost);
    size_t bad = 0, first = 0;
    for (size_t i = 0; i < h0.size(); ++i) if (memcmp(&h0[i], &h1[i], 4)) { if (!bad) first = i; ++bad; }
    printf("t256 vs t128: %zu of %zu outputs differ%s\n", bad, h0.size(), bad ? "" : " (bit-identical)");
    if (bad) { printf("  first at m %zu n %zu: %g vs %g\n", first / N, first % N, h0[first], h1[first]); return 3; }
  }
  hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  for (int rep = 0; rep < 4; ++rep) {
    for (int which = 0; which < 2; ++which) {
      g.y = which ? y1 : y0;
      for (int i = 0; i < 10; ++i) { if (which) k256<<<grid256, 256, T256_LDS>>>(g, pl); else k128<<<grid128, 256, T128_LDS>>>(g); }
      (void)hipEventRecord(a);
      for (int i = 0; i < 100; ++i) { if (which) k256<<<grid256, 256, T256_LDS>>>(g, pl); else k128<<<grid128, 256, T128_LDS>>>(g); }
      (void)hipEventRecord(b); (void)hipEventSynchronize(b);
      float ms; (void)hipEventElapsedTime(&ms, a, b);
      printf("%s DIAG=%d: %.1f us\n", which ? "t256" : "t128", T256_DIAG, ms / 100 * 1e3f);
    }
  }
  if (T256_DIAG & 8) {
    std::vector<unsigned long long> h((size_t)grid256 * 4 * 8);
    (void)hipMemcpy(h.data(), dbg, h.size() * 8, hipMemcpyDeviceToHost);
    double sum[6] = {0, 0, 0, 0, 0, 0};
    unsigned long long s_min = ~0ull, e_min = ~0ull, e_max = 0;
    for (int bk = 0; bk < grid256; ++bk) for (int w = 0; w < 4; ++w) {
      for (int i = 0; i < 6; ++i) sum[i] += (double)h[((size_t)bk * 4 + w) * 8 + i];
      const unsigned long long st = h[((size_t)bk * 4 + w) * 8 + 6], en = h[((size_t)bk * 4 + w) * 8 + 7];
      s_min = std::min(s_min, st); e_min = std::min(e_min, en); e_max = std::max(e_max, en);
    }
    const double nw = (double)grid256 * 4;
    const double stages = (double)(K / 64 + 2 * (R / 64)) * (pl.n_full + pl.n_half) / grid256;
    printf("first end %.2f us, last end %.2f us after the first start\n", (e_min - s_min) / 100.0, (e_max - s_min) / 100.0);
    printf("stamps per wave: %.1f stages: wait %.0f | reads + MFMAs %.0f | barrier %.0f | copy issue %.0f (cycles per stage); epilogues %.0f cycles per wave; wave lifetime %.0f cycles\n",
           stages, sum[0] / nw / stages, sum[1] / nw / stages, sum[2] / nw / stages, sum[3] / nw / stages, sum[4] / nw, sum[5] / nw);
  }
  return hipGetLastError() == hipSuccess ? 0 : 1;
}
