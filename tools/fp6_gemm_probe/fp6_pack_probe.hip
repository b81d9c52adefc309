// Building blocks for the FP6 operand format of DESIGN.md section 6 (round-4 plan; NOT part of the library): the two packing kernels the
// production path will need, checked bit for bit against the host packing that tools/fp6_gemm_probe/fp6_gemm_probe.hip feeds its kernels.
//   fp6_levels_kernel : activation levels q = clamp(rint(x / sx[k]), -7, 7) -> e2m3 codes, packed in MFMA-operand-tile order
//   fp6_digits_kernel : W'[n,k] (fp32) -> per-row power of two 2^E[n], w = round(W' 2^E) as a 25-bit integer, five balanced radix-32 digits
//                       (|d| <= 16) as e2m3 codes of d / 8, one packed plane per digit; rowscale[n] = 2^-E[n]
// Layout (both): rows in pairs of 32, k in blocks of 128; one (pair, k block) = 3072 B = [rec0 16-B parts 1 KB | rec1 16-B parts 1 KB |
// rec0 8-B parts 512 B | rec1 8-B parts 512 B], rec = 16 rows; lane l = (row & 15) + 16 (k / 32 % 4) owns a 24-B field = its row's 32 codes of
// k = 32 (l >> 4) .. + 31, 6 bits each, LSB first: bytes 0..15 in the 16-B part, bytes 16..23 in the 8-B part.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off tools/fp6_gemm_probe/fp6_pack_probe.hip -o tools/fp6_gemm_probe/fp6_pack_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
#include <math.h>
#include <vector>
#include <random>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)
constexpr int F6_PAIR = 3072;
constexpr int NL = 5;

__device__ __host__ inline int e2m3_of_eighths(int d) {          // d / 8, |d| <= 16
  const int a = d < 0 ? -d : d;
  const int code = a < 8 ? a : (a < 16 ? (0x08 | (a - 8)) : 0x10);
  return (d < 0 ? 0x20 : 0) | code;
}
__device__ __host__ inline int e2m3_of_int(int q) {              // |q| <= 7: 0x1E1C1A1814100800 holds the eight codes, one per byte
  const int a = q < 0 ? -q : q;
  const int code = (int)((0x1E1C1A1814100800ull >> (8 * a)) & 0x3F);
  return (q < 0 ? 0x20 : 0) | code;
}

// One workgroup (256 threads) per (pair of 32 rows, 128-deep k block).  `codes` (LDS): [planes][32 rows][128 k] bytes.  Thread t < 128 owns the
// field of row (t & 15) + 16 (t >> 6), k quarter (t >> 4) & 3 -- 16 consecutive threads write 256 contiguous bytes of a record's 16-B parts.
template <int PLANES>
__device__ __forceinline__ void write_fields(const unsigned char (*codes)[32][128], unsigned char* out, int64_t plane_stride, int tid) {
  if (tid >= 128) return;
  const int l15 = tid & 15, kq = (tid >> 4) & 3, rec = tid >> 6, row = rec * 16 + l15, lane = l15 + 16 * kq;
#pragma unroll
  for (int pl = 0; pl < PLANES; ++pl) {
    unsigned w[6] = {0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < 32; ++i) {
      const unsigned c = codes[pl][row][32 * kq + i];
      const int bit = 6 * i;
      w[bit >> 5] |= c << (bit & 31);
      if ((bit & 31) > 26) w[(bit >> 5) + 1] |= c >> (32 - (bit & 31));
    }
    unsigned char* base = out + (int64_t)pl * plane_stride;
    *reinterpret_cast<uint4*>(base + rec * 1024 + lane * 16) = make_uint4(w[0], w[1], w[2], w[3]);
    *reinterpret_cast<uint2*>(base + 2048 + rec * 512 + lane * 8) = make_uint2(w[4], w[5]);
  }
}

// activation levels: x [M, K] fp32 (M % 32 == 0, K % 128 == 0), sx [K]; out [M/32][K/128][3072]
__global__ __launch_bounds__(256) void fp6_levels_kernel(const float* x, const float* sx, unsigned char* out, int M, int K) {
  __shared__ unsigned char codes[1][32][128];
  const int KB = K / 128, pair = blockIdx.x / KB, kb = blockIdx.x % KB, tid = threadIdx.x;
  // 32 rows x 128 k = 4096 elements: thread t takes row t >> 3, 16 consecutive k (four float4)
  const int row = tid >> 3, k0 = (tid & 7) * 16;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float4 v = *reinterpret_cast<const float4*>(x + (int64_t)(pair * 32 + row) * K + kb * 128 + k0 + 4 * j);
    const float4 s = *reinterpret_cast<const float4*>(sx + kb * 128 + k0 + 4 * j);
    const float vv[4] = {v.x, v.y, v.z, v.w}, ss[4] = {s.x, s.y, s.z, s.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float q = fminf(fmaxf(rintf(vv[e] / ss[e]), -7.f), 7.f);       // quantization_methods.py:14-15 (IEEE division, ties to even)
      codes[0][row][k0 + 4 * j + e] = (unsigned char)e2m3_of_int((int)q);
    }
  }
  __syncthreads();
  write_fields<1>(codes, out + ((int64_t)pair * KB + kb) * F6_PAIR, 0, tid);
}

// weight digits: Wp [N, K] fp32 (N % 32 == 0, K % 128 == 0); rowexp[n] = E[n] computed by fp6_rowexp_kernel; out [NL][N/32][K/128][3072]
__global__ __launch_bounds__(256) void fp6_rowexp_kernel(const float* Wp, int* rowexp, float* rowscale, int N, int K) {   // one wave per row
  const int n = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (n >= N) return;
  float mx = 0.f;
  for (int k = lane * 4; k < K; k += 256) { const float4 v = *reinterpret_cast<const float4*>(Wp + (int64_t)n * K + k); mx = fmaxf(fmaxf(fmaxf(mx, fabsf(v.x)), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))); }
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
  int ex = 0; (void)frexpf(mx, &ex);
  const int E = 5 * NL - 1 - ex;                                   // row maximum in [2^23, 2^24)
  if (lane == 0) { rowexp[n] = E; rowscale[n] = ldexpf(1.f, -E); }
}
__global__ __launch_bounds__(256) void fp6_digits_kernel(const float* Wp, const int* rowexp, unsigned char* out, int N, int K) {
  __shared__ unsigned char codes[NL][32][128];
  const int KB = K / 128, pair = blockIdx.x / KB, kb = blockIdx.x % KB, tid = threadIdx.x;
  const int row = tid >> 3, k0 = (tid & 7) * 16;
  const int E = rowexp[pair * 32 + row];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float4 v = *reinterpret_cast<const float4*>(Wp + (int64_t)(pair * 32 + row) * K + kb * 128 + k0 + 4 * j);
    const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      int wi = (int)rintf(ldexpf(vv[e], E));                       // exact scaling; |w| <= 2^24: the rounding is to the nearest integer (ties to even)
      const int sgn = wi < 0 ? -1 : 1;
      wi = wi < 0 ? -wi : wi;
#pragma unroll
      for (int pl = 0; pl < NL; ++pl) {
        int d = wi & 31; if (d > 16) d -= 32;
        wi = (wi - d) >> 5;
        codes[pl][row][k0 + 4 * j + e] = (unsigned char)e2m3_of_eighths(sgn * d);
      }
    }
  }
  __syncthreads();
  write_fields<NL>(codes, out + ((int64_t)pair * KB + kb) * F6_PAIR, (int64_t)(N / 32) * KB * F6_PAIR, tid);
}

// ---- host reference: the packing of fp6_gemm_probe.hip
static std::vector<unsigned char> pack6(const std::vector<unsigned char>& codes, int rows, int K) {
  const int KB = K / 128, pairs = rows / 32;
  std::vector<unsigned char> out((size_t)pairs * KB * F6_PAIR, 0);
  for (int pr = 0; pr < pairs; ++pr)
    for (int kb = 0; kb < KB; ++kb) {
      unsigned char* base = out.data() + ((size_t)pr * KB + kb) * F6_PAIR;
      for (int rec = 0; rec < 2; ++rec)
        for (int l = 0; l < 64; ++l) {
          const int row = pr * 32 + rec * 16 + (l & 15), k0 = kb * 128 + 32 * (l >> 4);
          unsigned char bytes[24] = {0};
          for (int i = 0; i < 32; ++i) {
            const unsigned c = codes[(size_t)row * K + k0 + i];
            const int bit = 6 * i;
            bytes[bit >> 3] |= (unsigned char)(c << (bit & 7));
            if ((bit & 7) > 2) bytes[(bit >> 3) + 1] |= (unsigned char)(c >> (8 - (bit & 7)));
          }
          memcpy(base + rec * 1024 + l * 16, bytes, 16);
          memcpy(base + 2048 + rec * 512 + l * 8, bytes + 16, 8);
        }
    }
  return out;
}

int main() {
  const int M = 8192, N = 3072, K = 768, KB = K / 128;
  std::mt19937 rng(3);
  std::normal_distribution<float> nd(0.f, 1.f);
  std::vector<float> x((size_t)M * K), sx(K), Wp((size_t)N * K);
  for (auto& v : sx) v = 0.2f + 0.3f * fabsf(nd(rng));
  for (auto& v : x) v = nd(rng) * 1.3f;
  for (size_t i = 0; i < x.size(); i += 97) x[i] = (float)((int)(rng() % 15) - 7 + 0.5f) * sx[i % K];   // rounding ties
  for (auto& v : Wp) v = nd(rng) * 0.02f * (1.f + (float)(rng() % 7));
  // ---- host
  std::vector<unsigned char> qc((size_t)M * K);
  for (int m = 0; m < M; ++m) for (int k = 0; k < K; ++k) {
    const float q = fminf(fmaxf(rintf(x[(size_t)m * K + k] / sx[k]), -7.f), 7.f);
    qc[(size_t)m * K + k] = (unsigned char)e2m3_of_int((int)q);
  }
  const auto refA = pack6(qc, M, K);
  std::vector<std::vector<unsigned char>> dc(NL, std::vector<unsigned char>((size_t)N * K));
  std::vector<float> rs_ref(N);
  for (int n = 0; n < N; ++n) {
    float mx = 0; for (int k = 0; k < K; ++k) mx = fmaxf(mx, fabsf(Wp[(size_t)n * K + k]));
    int ex; frexpf(mx, &ex); const int E = 5 * NL - 1 - ex; rs_ref[n] = ldexpf(1.f, -E);
    for (int k = 0; k < K; ++k) {
      long long wi = llrint((double)ldexpf(Wp[(size_t)n * K + k], E));
      const int sgn = wi < 0 ? -1 : 1; wi = llabs(wi);
      for (int pl = 0; pl < NL; ++pl) { int d = (int)(wi % 32); if (d > 16) d -= 32; wi = (wi - d) / 32; dc[pl][(size_t)n * K + k] = (unsigned char)e2m3_of_eighths(sgn * d); }
    }
  }
  // ---- device
  float *dx, *dsx, *dW, *drs; unsigned char *dA, *dD; int* dE;
  const size_t a_bytes = (size_t)(M / 32) * KB * F6_PAIR, d_bytes = (size_t)NL * (N / 32) * KB * F6_PAIR;
  CK(hipMalloc(&dx, x.size() * 4)); CK(hipMalloc(&dsx, K * 4)); CK(hipMalloc(&dW, Wp.size() * 4)); CK(hipMalloc(&drs, N * 4)); CK(hipMalloc(&dE, N * 4));
  CK(hipMalloc(&dA, a_bytes)); CK(hipMalloc(&dD, d_bytes));
  CK(hipMemcpy(dx, x.data(), x.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dsx, sx.data(), K * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dW, Wp.data(), Wp.size() * 4, hipMemcpyHostToDevice));
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 2; ++rep) {
    float ms_a, ms_d;
    hipEventRecord(e0); for (int i = 0; i < 20; ++i) fp6_levels_kernel<<<(M / 32) * KB, 256>>>(dx, dsx, dA, M, K); hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms_a, e0, e1);
    hipEventRecord(e0);
    for (int i = 0; i < 20; ++i) { fp6_rowexp_kernel<<<(N + 3) / 4, 256>>>(dW, dE, drs, N, K); fp6_digits_kernel<<<(N / 32) * KB, 256>>>(dW, dE, dD, N, K); }
    hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms_d, e0, e1);
    if (rep) printf("fp6_levels_kernel %d x %d: %.1f us | fp6_rowexp + fp6_digits %d x %d (5 planes): %.1f us   (stand-alone, unfused: the production pass would do this inside the activation pass / the weight-row workgroups)\n",
                    M, K, ms_a * 50.f, N, K, ms_d * 50.f);
  }
  CK(hipDeviceSynchronize());
  std::vector<unsigned char> gotA(a_bytes), gotD(d_bytes); std::vector<float> rs(N);
  CK(hipMemcpy(gotA.data(), dA, a_bytes, hipMemcpyDeviceToHost)); CK(hipMemcpy(gotD.data(), dD, d_bytes, hipMemcpyDeviceToHost)); CK(hipMemcpy(rs.data(), drs, N * 4, hipMemcpyDeviceToHost));
  size_t badA = 0, badD = 0, badS = 0;
  for (size_t i = 0; i < a_bytes; ++i) badA += gotA[i] != refA[i];
  for (int pl = 0; pl < NL; ++pl) { const auto ref = pack6(dc[pl], N, K); const unsigned char* g = gotD.data() + (size_t)pl * (N / 32) * KB * F6_PAIR; for (size_t i = 0; i < ref.size(); ++i) badD += g[i] != ref[i]; }
  for (int n = 0; n < N; ++n) badS += rs[n] != rs_ref[n];
  printf("packed activation levels: %zu of %zu bytes differ from the host packing; digit planes: %zu of %zu; row scales: %zu of %d\n", badA, a_bytes, badD, d_bytes, badS, N);
  return (badA || badD || badS) ? 1 : 0;
}
