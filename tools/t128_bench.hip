// Timing of the 128x128 contraction kernel at the headline shape under compile-time variants (kernel tuning only):
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -DT128_DIAG=<bits> -DT128_GROUP_M=<n> -DT128_WGS=<2|3> tools/t128_bench.hip -o <bin>
#include <stdarg.h>
#include <vector>
#include <algorithm>
#include <random>
#include "../llm-qat-on-gpt2_amd/csrc/spq_f16x2.hip"
namespace spq {
void set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fprintf(stderr, "\n"); }
int check_launch(const char* what) { hipError_t e = hipGetLastError(); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", what, hipGetErrorString(e)); return -2; } return 0; }
}
using namespace spq;
int main(int argc, char** argv) {
  const int M = 8192, N = 3072, K = 768, R = 64;
  std::mt19937 rng(1);
  auto fill = [&](size_t n, int kind) {
    std::vector<_Float16> h(n);
    std::uniform_int_distribution<int> lv(-7, 7); std::normal_distribution<float> nd(0.f, 3000.f);
    for (auto& v : h) v = kind == 0 ? (_Float16)(float)lv(rng) : (_Float16)nd(rng);
    _Float16* d; hipMalloc(&d, n * 2); hipMemcpy(d, h.data(), n * 2, hipMemcpyHostToDevice); return d;
  };
  GemmF16Args g;
  g.qx = fill((size_t)M * K, 0); g.thi = fill((size_t)M * R, 1); g.tlo = fill((size_t)M * R, 1);
  g.Whi = fill((size_t)N * K, 1); g.Wlo = fill((size_t)N * K, 1); g.Bhi = fill((size_t)N * R, 1); g.Blo = fill((size_t)N * R, 1);
  float *ri, *rs, *bias, *y;
  hipMalloc(&ri, M * 4); hipMalloc(&rs, N * 4); hipMalloc(&bias, N * 4); hipMalloc(&y, (size_t)M * N * 4);
  hipMemset(ri, 0, M * 4); hipMemset(rs, 0, N * 4); hipMemset(bias, 0, N * 4);
  g.rowinv = ri; g.rowscale = rs; g.bias = bias; g.y = y; g.M = M; g.N = N; g.Kp = K; g.Rp = R;
  g.tiles_m = M / GM; g.tiles_n = N / GN; g.dbg = nullptr; g.xl = nullptr; g.xscale = nullptr; g.a_limbs = 1;
  unsigned long long* dbg = nullptr;
  if (T128_DIAG & 8) { hipMalloc(&dbg, 4096 * 4 * 8 * 8); hipMemset(dbg, 0, 4096 * 4 * 8 * 8); g.dbg = dbg; }
  auto k = gemm_f16x2_t128_kernel<1, 0>;
  constexpr int LDS_USED = T128_LDS;
  hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_USED);
  const int ntiles = 2 * g.tiles_m * g.tiles_n;
  const unsigned cap = T128_WGS * gemm_grid(1 << 30);
  const unsigned grid = (unsigned)ntiles < cap ? (unsigned)ntiles : cap;
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int rep = 0; rep < 3; ++rep) {
    for (int i = 0; i < 10; ++i) k<<<grid, 256, LDS_USED>>>(g);
    hipEventRecord(a);
    for (int i = 0; i < 100; ++i) k<<<grid, 256, LDS_USED>>>(g);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("t128 WGS=%d GROUP_M=%d DIAG=%d ORDER=%d: %.1f us\n", T128_WGS, T128_GROUP_M, T128_DIAG, T128_ORDER, ms / 100 * 1e3f);
  }
  if (T128_DIAG & 8) {
    std::vector<unsigned long long> h((size_t)grid * 4 * 8);
    hipMemcpy(h.data(), dbg, h.size() * 8, hipMemcpyDeviceToHost);
    double sum[6] = {0, 0, 0, 0, 0, 0};
    for (unsigned b = 0; b < grid; ++b) for (int w = 0; w < 4; ++w) for (int i = 0; i < 6; ++i) sum[i] += (double)h[((size_t)b * 4 + w) * 8 + i];
    unsigned long long s_min = ~0ull, s_max = 0, e_min = ~0ull, e_max = 0;
    for (unsigned b = 0; b < grid; ++b) for (int w = 0; w < 4; ++w) {
      const unsigned long long st = h[((size_t)b * 4 + w) * 8 + 6], en = h[((size_t)b * 4 + w) * 8 + 7];
      s_min = st < s_min ? st : s_min; s_max = st > s_max ? st : s_max; e_min = en < e_min ? en : e_min; e_max = en > e_max ? en : e_max;
    }
    printf("last launch, 100-MHz counter, us after the first wave's start: last start %.2f, first end %.2f, last end %.2f\n", (s_max - s_min) / 100.0,
           (e_min - s_min) / 100.0, (e_max - s_min) / 100.0);
    for (unsigned lo = 0; lo < grid; lo += 256) {
      double e = 0, st = 0; unsigned long long emin = ~0ull, emax = 0; int n = 0;
      for (unsigned b = lo; b < lo + 256 && b < grid; ++b) {
        const unsigned long long en = h[((size_t)b * 4) * 8 + 7] - s_min;
        e += (double)en; st += (double)(h[((size_t)b * 4) * 8 + 6] - s_min); emin = en < emin ? en : emin; emax = en > emax ? en : emax; ++n;
      }
      printf("  blocks %u..%u: start %.2f us, end mean %.2f min %.2f max %.2f us\n", lo, lo + n - 1, st / n / 100.0, e / n / 100.0, emin / 100.0, emax / 100.0);
    }
    {   // end-time histogram (2-us bins) and the late finishers by XCD (block index mod 8)
      int hist[64] = {0}, late_xcd[8] = {0}, n_xcd[8] = {0};
      std::vector<double> ends;
      for (unsigned b = 0; b < grid; ++b) ends.push_back((double)(h[((size_t)b * 4) * 8 + 7] - s_min) / 100.0);
      std::vector<double> sorted = ends; std::sort(sorted.begin(), sorted.end());
      const double p90 = sorted[(size_t)(0.9 * sorted.size())];
      for (unsigned b = 0; b < grid; ++b) { int bin = (int)(ends[b] / 2.0); if (bin > 63) bin = 63; ++hist[bin]; ++n_xcd[b & 7]; if (ends[b] > p90) ++late_xcd[b & 7]; }
      printf("  end-time histogram (us: workgroups):");
      for (int i = 0; i < 64; ++i) if (hist[i]) printf(" %d-%d:%d", 2 * i, 2 * i + 2, hist[i]);
      printf("\n  slowest 10%% (> %.1f us) by XCD:", p90);
      for (int x = 0; x < 8; ++x) printf(" %d", late_xcd[x]);
      printf("\n");
    }
    printf("wave lifetime %.0f shader cycles = %.2f us of the 100-MHz counter: shader clock %.0f MHz\n", sum[4] / ((double)grid * 4),
           sum[5] / ((double)grid * 4) / 100.0, sum[4] / sum[5] * 100.0);
    const double nw = (double)grid * 4, stages = (double)ntiles / grid * (K / 64 + 2 * (R / 64));
    printf("stamps (cycles per stage and wave; %.1f stages per workgroup): wait-for-copies %.0f | fragment reads + MFMAs %.0f | barrier %.0f | copy issue %.0f"
           " || kernel per wave %.0f cycles, epilogue + rest per tile %.0f\n", stages, sum[0] / nw / stages, sum[1] / nw / stages, sum[2] / nw / stages,
           sum[3] / nw / stages, sum[4] / nw, (sum[4] - sum[0] - sum[1] - sum[2] - sum[3]) / nw / ((double)ntiles / grid));
  }
  return hipGetLastError() == hipSuccess ? 0 : 1;
}
