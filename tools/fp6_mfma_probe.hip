// Feasibility probe for the FP6 limb format of DESIGN.md section 6 (not part of the library):
//   (1) is v_mfma_scale_f32_16x16x128_f8f6f4 with e2m3 operands EXACT for small-integer activations times 4-significant-bit weight limbs
//       (every product a 7-bit integer times a power of two, 128 products summed inside one instruction, fp32 C operand up to 2^17)?
//   (2) what does one instruction cost next to v_mfma_f32_16x16x32_f16 from the same wave(s)?
// build: hipcc -O3 --offload-arch=gfx950 tools/fp6_mfma_probe.hip -o tools/fp6_mfma_probe     run: tools/fp6_mfma_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <random>
#include <vector>

typedef int v8i __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

// e2m3: sign(1) exp(2) mant(3), bias 1; exp 0 = subnormal (mant / 8)
static double e2m3_value(int code) {
  const int s = (code >> 5) & 1, e = (code >> 3) & 3, m = code & 7;
  const double v = e == 0 ? m / 8.0 : ldexp(1.0 + m / 8.0, e - 1);
  return s ? -v : v;
}
static int e2m3_of_int(int q) {             // |q| <= 7, exact
  static const int code[8] = {0x00, 0x08, 0x10, 0x14, 0x18, 0x1A, 0x1C, 0x1E};
  return (q < 0 ? 0x20 : 0) | code[abs(q)];
}

// one wave: D = A (16 x 128) . B^T (16 x 128) + C.  a_codes / b_codes: [16][128] 6-bit codes, one byte each.
// Assumed operand layout (as the other 16x16 MFMAs): lane l holds row l & 15, k = 32 (l >> 4) .. + 31, 6 bits each, packed LSB first.
__global__ void fp6_once(const unsigned char* a_codes, const unsigned char* b_codes, const float* c_in, float* d_out, int kperm) {
  const int l = threadIdx.x, row = l & 15, kb = l >> 4;
  unsigned a[8] = {0, 0, 0, 0, 0, 0, 0, 0}, b[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int i = 0; i < 32; ++i) {
    const int k = kperm == 0 ? 32 * kb + i : (i / 8) * 32 + kb * 8 + (i % 8);      // alternative k order, for the layout check
    const unsigned ca = a_codes[row * 128 + k], cb = b_codes[row * 128 + k];
    const int bit = 6 * i;
    a[bit >> 5] |= ca << (bit & 31); if ((bit & 31) > 26) a[(bit >> 5) + 1] |= ca >> (32 - (bit & 31));
    b[bit >> 5] |= cb << (bit & 31); if ((bit & 31) > 26) b[(bit >> 5) + 1] |= cb >> (32 - (bit & 31));
  }
  v8i va, vb;
  for (int i = 0; i < 8; ++i) { va[i] = (int)a[i]; vb[i] = (int)b[i]; }
  f32x4 c;
  for (int e = 0; e < 4; ++e) c[e] = c_in[(4 * kb + e) * 16 + row];              // C/D: col = lane & 15, row = 4 (lane >> 4) + e
  const f32x4 d = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(va, vb, c, 2, 2, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
  for (int e = 0; e < 4; ++e) d_out[(4 * kb + e) * 16 + row] = d[e];
}

// rate: every wave streams `iters` rounds of 16 independent MFMAs
template <int KIND>   // 0: fp6 16x16x128 scaled, 1: f16 16x16x32
__global__ __launch_bounds__(512) void rate_kernel(float* sink, unsigned long long* cycles, int iters) {
  f32x4 acc[16];
  for (int i = 0; i < 16; ++i) for (int e = 0; e < 4; ++e) acc[i][e] = 0.f;
  v8i va, vb; f16x8 ha, hb;
  for (int i = 0; i < 8; ++i) { va[i] = 0x08208208 + threadIdx.x; vb[i] = 0x10410410 + i; ha[i] = (_Float16)1.f; hb[i] = (_Float16)2.f; }
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      if (KIND == 0) acc[i] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(va, vb, acc[i], 2, 2, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
      else acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ha, hb, acc[i], 0, 0, 0);
    }
  }
  float s = 0.f;
  for (int i = 0; i < 16; ++i) s += acc[i][0];
  const unsigned long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0 && blockIdx.x == 0) cycles[0] = t1 - t0;
  if (s == 12345.f) sink[0] = s;
}

int main() {
  std::mt19937 rng(7);
  std::vector<unsigned char> a(16 * 128), b(16 * 128);
  std::vector<float> c(256), d(256);
  unsigned char *da, *db; float *dc, *dd;
  CK(hipMalloc(&da, a.size())); CK(hipMalloc(&db, b.size())); CK(hipMalloc(&dc, 1024)); CK(hipMalloc(&dd, 1024));
  int worst_perm = -1;
  for (int trial = 0; trial < 6; ++trial) {
    // trial 0-1: integer activations x arbitrary e2m3 limbs, C = 0; 2-3: C large (2^17 + fraction); 4-5: both operands arbitrary codes
    for (auto& v : a) v = trial < 4 ? (unsigned char)e2m3_of_int((int)(rng() % 15) - 7) : (unsigned char)(rng() & 63);
    for (auto& v : b) v = (unsigned char)(rng() & 63);
    for (auto& v : c) v = (trial == 2 || trial == 3) ? ldexpf(1.f, 17) + (float)(rng() % 64) / 64.f : 0.f;
    CK(hipMemcpy(da, a.data(), a.size(), hipMemcpyHostToDevice)); CK(hipMemcpy(db, b.data(), b.size(), hipMemcpyHostToDevice));
    CK(hipMemcpy(dc, c.data(), 1024, hipMemcpyHostToDevice));
    for (int perm = 0; perm < 2; ++perm) {
      fp6_once<<<1, 64>>>(da, db, dc, dd, perm);
      CK(hipMemcpy(d.data(), dd, 1024, hipMemcpyDeviceToHost));
      int bad = 0; double maxerr = 0;
      for (int i = 0; i < 16; ++i)
        for (int j = 0; j < 16; ++j) {
          double s = c[i * 16 + j];
          for (int k = 0; k < 128; ++k) s += e2m3_value(a[i * 128 + k]) * e2m3_value(b[j * 128 + k]);
          const double err = fabs((double)d[i * 16 + j] - s);
          if (err != 0) ++bad;
          maxerr = fmax(maxerr, err);
        }
      printf("trial %d layout %d: %d of 256 outputs differ from the exact sum, max |err| %.3g\n", trial, perm, bad, maxerr);
      if (bad == 0) worst_perm = perm;
    }
  }
  printf("layout that reproduces the exact sums: %d (0 = lane holds k = 32 (l >> 4) .. + 31)\n", worst_perm);

  unsigned long long* dcyc; float* sink;
  CK(hipMalloc(&dcyc, 8)); CK(hipMalloc(&sink, 4));
  const int iters = 2000;
  for (int waves_per_simd = 1; waves_per_simd <= 2; ++waves_per_simd) {
    const int threads = 256 * waves_per_simd;
    unsigned long long cyc[2];
    for (int kind = 0; kind < 2; ++kind) {
      for (int rep = 0; rep < 2; ++rep) {
        if (kind == 0) rate_kernel<0><<<256, threads>>>(sink, dcyc, iters); else rate_kernel<1><<<256, threads>>>(sink, dcyc, iters);
        CK(hipDeviceSynchronize());
      }
      CK(hipMemcpy(&cyc[kind], dcyc, 8, hipMemcpyDeviceToHost));
    }
    printf("%d wave(s) per SIMD, every CU busy: fp6 16x16x128 %.1f counter cycles per MFMA per wave (K = 128), f16 16x16x32 %.1f (K = 32)"
           " -> fp6 delivers %.2fx the k-depth per cycle\n", waves_per_simd, (double)cyc[0] / (16.0 * iters), (double)cyc[1] / (16.0 * iters),
           4.0 * (double)cyc[1] / (double)cyc[0]);
  }
  return 0;
}
