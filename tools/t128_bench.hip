// Timing of the 128x128 contraction kernel at the headline shape under compile-time variants (kernel tuning only):
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -DT128_DIAG=<bits> -DT128_GROUP_M=<n> -DT128_WGS=<2|3> tools/t128_bench.hip -o <bin>
#include <stdarg.h>
#include <vector>
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
  auto k = gemm_f16x2_t128_kernel<1, 0>;
  hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, T128_LDS);
  const int ntiles = 2 * g.tiles_m * g.tiles_n;
  const unsigned cap = T128_WGS * gemm_grid(1 << 30);
  const unsigned grid = (unsigned)ntiles < cap ? (unsigned)ntiles : cap;
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int rep = 0; rep < 3; ++rep) {
    for (int i = 0; i < 10; ++i) k<<<grid, 256, T128_LDS>>>(g);
    hipEventRecord(a);
    for (int i = 0; i < 100; ++i) k<<<grid, 256, T128_LDS>>>(g);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("t128 WGS=%d GROUP_M=%d DIAG=%d: %.1f us\n", T128_WGS, T128_GROUP_M, T128_DIAG, ms / 100 * 1e3f);
  }
  return hipGetLastError() == hipSuccess ? 0 : 1;
}
