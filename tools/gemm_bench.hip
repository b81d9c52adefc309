// Stand-alone timing harness for the f16x2 contraction kernel (kernel tuning only; not part of the product).
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off tools/gemm_bench.hip -o tools/gemm_bench
#include <stdarg.h>
#include <vector>
#include <random>
#include "../llm-qat-on-gpt2_amd/csrc/spq_f16x2.hip"
#include "variants/gemm_256x128.h"     // the round-1 kernels this tool dissects (no longer in the library)
#include "variants/gemm_u8x2.h"
namespace spq {
void set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fprintf(stderr, "\n"); }
int check_launch(const char* what) { hipError_t e = hipGetLastError(); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", what, hipGetErrorString(e)); return -2; } return 0; }
}
using namespace spq;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int DIAG>
float run(const GemmF16Args& g, int iters) {
  hipFuncSetAttribute((const void*)gemm_f16x2_kernel<DIAG>, hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_LDS);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int i = 0; i < 5; ++i) gemm_f16x2_kernel<DIAG><<<gemm_grid(g.tiles_m * g.tiles_n), GEMM_THREADS, GEMM_LDS>>>(g);
  hipEventRecord(a);
  for (int i = 0; i < iters; ++i) gemm_f16x2_kernel<DIAG><<<gemm_grid(g.tiles_m * g.tiles_n), GEMM_THREADS, GEMM_LDS>>>(g);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  return ms / iters * 1e3f;
}

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
  g.tiles_m = M / GM; g.tiles_n = N / GN;
  {
    GemmU8Args u;
    std::vector<unsigned char> hb((size_t)M * K); for (auto& v : hb) v = (unsigned char)(121 + rng() % 15);
    unsigned char* dq; hipMalloc(&dq, hb.size()); hipMemcpy(dq, hb.data(), hb.size(), hipMemcpyHostToDevice);
    u.qx = dq; u.thi = g.thi; u.tlo = g.tlo; u.Whi = g.Whi; u.Wlo = g.Wlo; u.Bhi = g.Bhi; u.Blo = g.Blo;
    u.rowinv = ri; u.rowscale = rs; u.bias = bias; u.y = y; u.M = M; u.N = N; u.Kp = K; u.Rp = R; u.tiles_m = M / GM; u.tiles_n = N / GN;
    auto runu = [&](auto kern) {
      hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, U8_LDS);
      hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
      for (int i = 0; i < 5; ++i) kern<<<gemm_grid(u.tiles_m * u.tiles_n), GEMM_THREADS, U8_LDS>>>(u);
      hipEventRecord(a);
      for (int i = 0; i < 50; ++i) kern<<<gemm_grid(u.tiles_m * u.tiles_n), GEMM_THREADS, U8_LDS>>>(u);
      hipEventRecord(b); hipEventSynchronize(b);
      float ms; hipEventElapsedTime(&ms, a, b); return ms / 50 * 1e3f;
    };
    { GemmU8Args v = u; GemmU8Args keep = u; v.Kp = 64; u = v; printf("skeleton only, K=64 (5 stages/tile): %.1f us; ", runu(gemm_u8x2_kernel<26>)); v.Rp = 0; u = v; printf("K=64, no LoRA (1 stage/tile): %.1f us\n", runu(gemm_u8x2_kernel<26>)); u = keep; }
    for (int rep = 0; rep < 2; ++rep)
      printf("u8x2: full %.1f | no-copies %.1f | no-compute %.1f | no-stores %.1f | no-copies,no-stores %.1f | stores only %.1f | skeleton only %.1f us\n",
             runu(gemm_u8x2_kernel<0>), runu(gemm_u8x2_kernel<2>),
             runu(gemm_u8x2_kernel<8>), runu(gemm_u8x2_kernel<16>), runu(gemm_u8x2_kernel<18>), runu(gemm_u8x2_kernel<10>), runu(gemm_u8x2_kernel<26>));
  }
  unsigned long long* dbg; hipMalloc(&dbg, 256 * 16); g.dbg = dbg;
  auto clock_of = [&](auto runner) {
    runner();
    std::vector<unsigned long long> h(512); hipMemcpy(h.data(), dbg, 512 * 8, hipMemcpyDeviceToHost);
    double cyc = 0, rt = 0; for (int i = 0; i < 256; ++i) { cyc += (double)h[2 * i]; rt += (double)h[2 * i + 1]; }
    return cyc / rt * 100.0;   // MHz: s_memrealtime ticks at 100 MHz
  };
  printf("in-kernel clock MHz: full %.0f | mfma-only %.0f | loads-only(no compute) %.0f | loads+mfma %.0f\n",
         clock_of([&] { run<16>(g, 20); }), clock_of([&] { run<16 + 13>(g, 20); }), clock_of([&] { run<16 + 2>(g, 20); }),
         clock_of([&] { run<16 + 12>(g, 20); }));
  printf("2x MFMA per stage: mfma-only %.1f | loads+mfma %.1f us\n", run<32 + 13>(g, 30), run<32 + 12>(g, 30));
  printf("no barriers at all: mfma-only %.1f | loads-only %.1f | loads+mfma %.1f us\n", run<64 + 13>(g, 30), run<64 + 2 + 4>(g, 30), run<64 + 12>(g, 30));
  {
    hipFuncSetAttribute((const void*)gemm_f16x2_s16_kernel<0, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_LDS);
    hipFuncSetAttribute((const void*)gemm_f16x2_s16_kernel<5, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_LDS);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int rep = 0; rep < 2; ++rep) {
      float ms[2];
      for (int v = 0; v < 2; ++v) {
        for (int i = 0; i < 5; ++i) { if (v == 0) gemm_f16x2_s16_kernel<0, 1><<<gemm_grid(g.tiles_m * g.tiles_n), GEMM_THREADS, GEMM_LDS>>>(g); else gemm_f16x2_s16_kernel<5, 1><<<gemm_grid(g.tiles_m * g.tiles_n), GEMM_THREADS, GEMM_LDS>>>(g); }
        hipEventRecord(a);
        for (int i = 0; i < 50; ++i) { if (v == 0) gemm_f16x2_s16_kernel<0, 1><<<gemm_grid(g.tiles_m * g.tiles_n), GEMM_THREADS, GEMM_LDS>>>(g); else gemm_f16x2_s16_kernel<5, 1><<<gemm_grid(g.tiles_m * g.tiles_n), GEMM_THREADS, GEMM_LDS>>>(g); }
        hipEventRecord(b); hipEventSynchronize(b);
        hipEventElapsedTime(&ms[v], a, b);
      }
      printf("16x16x32 variant: full %.1f | no-loads,no-stores %.1f us\n", ms[0] / 50 * 1e3f, ms[1] / 50 * 1e3f);
    }
  }
  const int iters = 50;
  for (int rep = 0; rep < 2; ++rep) {
    printf("full %.1f | no-loads %.1f | no-compute %.1f | no-stores %.1f | no-loads,no-stores %.1f | mfma-only %.1f | stores-only %.1f | loads+mfma(no frag reads, no stores) %.1f us\n",
           run<0>(g, iters), run<1>(g, iters), run<2>(g, iters), run<4>(g, iters), run<5>(g, iters), run<13>(g, iters), run<3>(g, iters), run<12>(g, iters));
  }
  return 0;
}
