// Stand-alone check + timing harness for the int8 contraction kernel (kernel tuning only; not part of the product).
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off tools/i8_gemm_bench.hip -o tools/i8_gemm_bench
// Checks the kernel on random operands against a double-precision host sum (this also pins the A/B lane maps of
// v_mfma_i32_32x32x32_i8: any k-permutation shared by both operands cancels, a wrong row/column map does not), then times
// the headline shape with the diagnostic variants.
#include <stdarg.h>
#include <stdio.h>
#include <random>
#include <vector>
#include "../llm-qat-on-gpt2_amd/csrc/spq_i8_kernel.h"
namespace spq {
void set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fprintf(stderr, "\n"); }
int check_launch(const char* what) { hipError_t e = hipGetLastError(); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", what, hipGetErrorString(e)); return -2; } return 0; }
}
using namespace spq;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <typename T> T* upload(const std::vector<T>& h) { T* d; hipMalloc(&d, h.size() * sizeof(T)); hipMemcpy(d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice); return d; }

static unsigned grid_for(int ntiles) {
  int dev = 0, n = 0; hipGetDevice(&dev); hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
  int cus = n >= 8 ? n / 8 * 8 : 256;
  return (unsigned)(ntiles < cus ? ntiles : cus);
}

template <int NL, int V>
float run(const GemmI8Args& g, int iters) {
  auto k = gemm_i8_kernel<NL, V, 0>;
  hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, I8_LDS);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  const unsigned grid = grid_for(g.tiles_m * g.tiles_n);
  for (int i = 0; i < 5; ++i) k<<<grid, I8_THREADS, I8_LDS>>>(g);
  hipEventRecord(a);
  for (int i = 0; i < iters; ++i) k<<<grid, I8_THREADS, I8_LDS>>>(g);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  if (hipGetLastError() != hipSuccess) printf("launch error\n");
  return ms / iters * 1e3f;
}

struct Problem {
  int M, N, K, R, Mp, Np, Kp, Rp, NL;
  std::vector<signed char> q, w;
  std::vector<_Float16> thi, tlo, bhi, blo;
  std::vector<float> rowinv, rowscale, bscale, bias;
  GemmI8Args g;
  float* y;
};

static int pad(int v, int a) { return (v + a - 1) / a * a; }

static void make(Problem& P, int M, int N, int K, int R, int NL, int qmax, unsigned seed) {
  std::mt19937 rng(seed);
  P.M = M; P.N = N; P.K = K; P.R = R; P.NL = NL;
  P.Mp = pad(M, I8_GM); P.Np = pad(N, I8_GN); P.Kp = pad(K, I8_GK); P.Rp = R ? pad(R, 64) : 0;
  std::uniform_int_distribution<int> lv(-qmax, qmax), dg(-128, 127), top(-64, 64);
  std::normal_distribution<float> nd(0.f, 3000.f);
  P.q.assign((size_t)P.Mp * P.Kp, 0); P.w.assign((size_t)NL * P.Np * P.Kp, 0);
  for (int m = 0; m < M; ++m) for (int k = 0; k < K; ++k) P.q[(size_t)m * P.Kp + k] = (signed char)lv(rng);
  for (int p = 0; p < NL; ++p)
    for (int n = 0; n < N; ++n) for (int k = 0; k < K; ++k)
      P.w[((size_t)p * P.Np + n) * P.Kp + k] = (signed char)((p == NL - 1 && NL == 3) ? top(rng) : dg(rng));
  auto f16v = [&](size_t rows, size_t rows_valid, int cols, int cols_valid) {
    std::vector<_Float16> v(rows * (size_t)cols, (_Float16)0.f);
    for (size_t r = 0; r < rows_valid; ++r) for (int c = 0; c < cols_valid; ++c) v[r * cols + c] = (_Float16)nd(rng);
    return v;
  };
  P.thi = f16v(P.Mp, M, P.Rp ? P.Rp : 1, R); P.tlo = f16v(P.Mp, M, P.Rp ? P.Rp : 1, R);
  P.bhi = f16v(P.Np, N, P.Rp ? P.Rp : 1, R); P.blo = f16v(P.Np, N, P.Rp ? P.Rp : 1, R);
  std::uniform_real_distribution<float> ur(0.5f, 2.f);
  P.rowinv.assign(P.Mp, 1.f); P.rowscale.assign(P.Np, 1.f); P.bscale.assign(P.Np, 1.f); P.bias.assign(N, 0.f);
  for (auto& v : P.rowinv) v = ldexpf(1.f, -10 + (int)(rng() % 4));
  for (auto& v : P.rowscale) v = ur(rng) * 1e-7f;
  for (auto& v : P.bscale) v = ldexpf(1.f, -14 + (int)(rng() % 4));
  for (auto& v : P.bias) v = ur(rng) - 1.f;
  GemmI8Args& g = P.g;
  g.qx = upload(P.q); g.W = upload(P.w); g.plane_stride = (int64_t)P.Np * P.Kp;
  g.thi = upload(P.thi); g.tlo = upload(P.tlo); g.Bhi = upload(P.bhi); g.Blo = upload(P.blo);
  g.rowinv = upload(P.rowinv); g.rowscale = upload(P.rowscale); g.bscale = upload(P.bscale); g.bias = upload(P.bias);
  hipMalloc(&P.y, (size_t)M * N * 4); hipMemset(P.y, 0xff, (size_t)M * N * 4);
  g.y = P.y; g.M = M; g.N = N; g.Kp = P.Kp; g.Rp = P.Rp; g.tiles_m = P.Mp / I8_GM; g.tiles_n = P.Np / I8_GN; g.epilogue = 0;
}

static double ref_at(const Problem& P, int m, int n, double* mag) {
  double base = 0;
  for (int p = 0; p < P.NL; ++p) {
    long long s = 0;
    for (int k = 0; k < P.K; ++k) s += (long long)P.q[(size_t)m * P.Kp + k] * (long long)P.w[((size_t)p * P.Np + n) * P.Kp + k];
    base += (double)s * (p == 0 ? 1.0 : (p == 1 ? 256.0 : 65536.0));
  }
  double u = 0;
  for (int j = 0; j < P.R; ++j) {
    const double th = (double)(float)P.thi[(size_t)m * P.Rp + j], tl = (double)(float)P.tlo[(size_t)m * P.Rp + j];
    const double bh = (double)(float)P.bhi[(size_t)n * P.Rp + j], bl = (double)(float)P.blo[(size_t)n * P.Rp + j];
    u += th * bh + th * bl + tl * bh;
  }
  double umag = 0;
  for (int j = 0; j < P.R; ++j) {
    const double th = fabs((double)(float)P.thi[(size_t)m * P.Rp + j]), tl = fabs((double)(float)P.tlo[(size_t)m * P.Rp + j]);
    const double bh = fabs((double)(float)P.bhi[(size_t)n * P.Rp + j]), bl = fabs((double)(float)P.blo[(size_t)n * P.Rp + j]);
    umag += th * bh + th * bl + tl * bh;
  }
  *mag = fabs(base * P.rowscale[n]) + umag * P.rowinv[m] * P.bscale[n] + fabs(P.bias[n]);   // the fp32 LoRA sum cancels internally
  return base * P.rowscale[n] + u * P.rowinv[m] * P.bscale[n] + P.bias[n];
}

template <int NL, int KIND = 0>   // KIND 0: 256x128 ring kernel; 1 / 2: 128x128 kernel with 1 / 2 stage buffers
static int check(int M, int N, int K, int R, int qmax, unsigned seed) {
  Problem P; make(P, M, N, K, R, NL, qmax, seed);
  if (KIND == 0) {
    auto k = gemm_i8_kernel<NL, 0, 0>;
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, I8_LDS);
    k<<<grid_for(P.g.tiles_m * P.g.tiles_n), I8_THREADS, I8_LDS>>>(P.g);
  } else if (KIND == 3) {
    auto k = gemm_i8_k128_kernel<NL, 0>;
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, i8k_lds(NL));
    const int nt = 2 * P.g.tiles_m * P.g.tiles_n; const unsigned c2 = 2 * grid_for(1 << 30);
    k<<<(unsigned)nt < c2 ? (unsigned)nt : c2, 256, i8k_lds(NL)>>>(P.g);
  } else {
    auto k = gemm_i8_t128_kernel<NL, KIND == 3 ? 1 : KIND, 0>;
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, i8t_lds(KIND));
    const int nt = 2 * P.g.tiles_m * P.g.tiles_n; const unsigned c2 = 2 * grid_for(1 << 30);
    k<<<(unsigned)nt < c2 ? (unsigned)nt : c2, 256, i8t_lds(KIND)>>>(P.g);
  }
  if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed: %s\n", hipGetErrorString(hipGetLastError())); return 1; }
  std::vector<float> y((size_t)M * N); hipMemcpy(y.data(), P.y, y.size() * 4, hipMemcpyDeviceToHost);
  std::mt19937 rng(seed + 99);
  double worst = 0; int bad = 0;
  const int samples = 4000;
  for (int s = 0; s < samples; ++s) {
    int m = rng() % M, n = rng() % N;
    if (s < 64) { m = (s & 1) ? M - 1 - (s >> 1) % M : (s >> 1) % M; n = (s & 2) ? N - 1 - (s >> 2) % N : (s >> 2) % N; }
    double mag;
    const double r = ref_at(P, m, n, &mag), got = y[(size_t)m * N + n];
    const double err = fabs(got - r) / (mag + 1e-6);     // relative to the terms' magnitudes (the terms cancel in the test data)
    if (!(err <= 5e-7)) { if (bad < 5) printf("  mismatch at (%d,%d): got %.9g want %.9g\n", m, n, got, r); ++bad; }
    if (err > worst) worst = err;
  }
  printf("check kind=%d NL=%d M=%d N=%d K=%d R=%d qmax=%d: %s (worst rel err %.2e over %d samples, %d bad)\n", KIND, NL, M, N, K, R, qmax,
         bad ? "FAIL" : "ok", worst, samples, bad);
  return bad ? 1 : 0;
}

int main(int argc, char** argv) {
  int fails = 0;
  fails += check<3>(512, 256, 128, 64, 7, 1);
  fails += check<3>(300, 200, 192, 100, 127, 2);     // ragged M, N; rank 100 (two 64-wide blocks, zero padded); 8-bit levels
  fails += check<3>(256, 128, 64, 0, 7, 3);          // no LoRA
  fails += check<1>(512, 384, 256, 64, 127, 4);
  fails += check<1>(257, 130, 64, 0, 7, 5);
  fails += check<3>(2048, 1024, 768, 64, 7, 6);      // several tiles per workgroup
  fails += check<3, 1>(512, 256, 128, 64, 7, 1);
  fails += check<3, 1>(300, 200, 192, 100, 127, 2);
  fails += check<1, 1>(257, 130, 64, 0, 7, 5);
  fails += check<3, 1>(8192, 3072, 768, 64, 7, 6);
  fails += check<3, 2>(512, 256, 128, 64, 7, 1);
  fails += check<3, 2>(300, 200, 192, 100, 127, 2);
  fails += check<1, 2>(512, 384, 256, 64, 127, 4);
  fails += check<3, 2>(8192, 3072, 768, 64, 7, 6);
  fails += check<3, 3>(512, 256, 128, 64, 7, 1);
  fails += check<3, 3>(300, 200, 256, 100, 127, 2);
  fails += check<1, 3>(512, 384, 256, 64, 127, 4);
  fails += check<1, 3>(257, 130, 128, 0, 7, 5);
  fails += check<3, 3>(8192, 3072, 768, 64, 7, 6);
  if (fails) printf("CHECK FAILED (%d cases)\n", fails);
  const int M = 8192, N = 3072, K = 768, R = 64;
  Problem P; make(P, M, N, K, R, 3, 7, 11);
  Problem Q; make(Q, M, N, K, R, 1, 7, 12);
  for (int rep = 0; rep < 2; ++rep) {
    printf("i8x3: full %.1f | no-copies %.1f | no-compute %.1f | no-stores %.1f | no-copies,no-stores %.1f | stores-only %.1f | skeleton %.1f us\n",
           run<3, 0>(P.g, 50), run<3, 2>(P.g, 50), run<3, 8>(P.g, 50), run<3, 16>(P.g, 50), run<3, 18>(P.g, 50), run<3, 10>(P.g, 50),
           run<3, 26>(P.g, 50));
    printf("i8x1: full %.1f | no-copies %.1f | no-compute %.1f | no-stores %.1f | no-copies,no-stores %.1f us\n",
           run<1, 0>(Q.g, 50), run<1, 2>(Q.g, 50), run<1, 8>(Q.g, 50), run<1, 16>(Q.g, 50), run<1, 18>(Q.g, 50));
  }
  { GemmI8Args g = P.g; g.Rp = 0; printf("i8x3 without LoRA-up: %.1f us\n", run<3, 0>(g, 50)); }
  auto runt = [&](auto k, int lds, const GemmI8Args& g) {
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    const int nt = 2 * g.tiles_m * g.tiles_n; const unsigned c2 = 2 * grid_for(1 << 30);
    const unsigned grid = (unsigned)nt < c2 ? (unsigned)nt : c2;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 5; ++i) k<<<grid, 256, lds>>>(g);
    hipEventRecord(a);
    for (int i = 0; i < 50; ++i) k<<<grid, 256, lds>>>(g);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); return ms / 50 * 1e3f;
  };
  for (int rep = 0; rep < 3; ++rep) {
    GemmI8Args g0 = P.g; g0.Rp = 0;
    printf("128x128 k128 kernel: i8x3 %.1f | i8x3 no LoRA %.1f | i8x1 %.1f us\n", runt(gemm_i8_k128_kernel<3, 0>, i8k_lds(3), P.g),
           runt(gemm_i8_k128_kernel<3, 0>, i8k_lds(3), g0), runt(gemm_i8_k128_kernel<1, 0>, i8k_lds(1), Q.g));
  }
  for (int rep = 0; rep < 2; ++rep)
    printf("128x128 kernel: i8x3 1-buffer %.1f | 2-buffer %.1f | i8x1 1-buffer %.1f | 2-buffer %.1f us\n",
           runt(gemm_i8_t128_kernel<3, 1, 0>, i8t_lds(1), P.g), runt(gemm_i8_t128_kernel<3, 2, 0>, i8t_lds(2), P.g),
           runt(gemm_i8_t128_kernel<1, 1, 0>, i8t_lds(1), Q.g), runt(gemm_i8_t128_kernel<1, 2, 0>, i8t_lds(2), Q.g));
  return 0;
}
