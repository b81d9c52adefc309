// Dense contraction on the fp32-input matrix cores (v_mfma_f32_32x32x2_f32): the always-valid operand path.
//
//   C[m,n] = sum_k A[m,k] B[n,k] + bias[n] + alpha2 * sum_j A2[m,j] B2[n,j]          ("NT": both K-contiguous)
//
// Numerics: an f32 MFMA is bit-for-bit a k-ordered fmaf chain (cdna guide §3), i.e. the same class of
// result as the CPU sgemm the reference calls; only the summation order differs.
//
// Tiling: 128x128 output tile per 256-thread workgroup (4 wavefronts, 64x64 each = 2x2 MFMA tiles of
// 32x32, 64 accumulator VGPRs), BK = 32.  Operand tiles go global -> registers (prefetched one k-step
// ahead, under the MFMAs of the current step) -> LDS [128][36] (row stride 144 B keeps the 16-byte
// fragment reads conflict-free: slot = 9*row + const (mod 16), 9 is odd).  The MFMA k index is permuted
// so that each lane reads 16 CONTIGUOUS k of its row (lane half h owns k = 16h..16h+15): four
// ds_read_b128 per 32x32 fragment instead of sixteen ds_read_b32.  2 workgroups per CU co-reside (37 KB
// LDS, <=128 VGPRs), so one computes while the other waits at its barrier.
#include "spq_common.h"

namespace spq {

constexpr int BM = 128, BN = 128, BK = 32, LDS_LD = BK + 4;
typedef float f32x16 __attribute__((ext_vector_type(16)));

struct GemmF32Args {
  const float* A; const float* B; const float* A2; const float* B2; const float* bias; float* C;
  int64_t lda, ldb, lda2, ldb2, ldc;
  int M, N, K, K2;
  float alpha2;
  int tiles_m, tiles_n;
};

// one operand tile: 128 rows x 32 k = 1024 float4, 4 per thread
struct TileRegs { float4 v[4]; };

template <bool FAST>
__device__ __forceinline__ void load_tile(TileRegs& t, const float* __restrict__ P, int64_t ld, int row0,
                                          int nrows, int k0, int K, int tid) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int idx = tid + 256 * i;
    const int r = idx >> 3, c = (idx & 7) << 2;
    const int gr = row0 + r, gk = k0 + c;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (gr < nrows) {
      const float* p = P + (int64_t)gr * ld + gk;
      if (FAST) {
        if (gk + 3 < K) v = *reinterpret_cast<const float4*>(p);
        else {
          if (gk + 0 < K) v.x = p[0];
          if (gk + 1 < K) v.y = p[1];
          if (gk + 2 < K) v.z = p[2];
        }
      } else {
        if (gk + 0 < K) v.x = p[0];
        if (gk + 1 < K) v.y = p[1];
        if (gk + 2 < K) v.z = p[2];
        if (gk + 3 < K) v.w = p[3];
      }
    }
    t.v[i] = v;
  }
}

__device__ __forceinline__ void store_tile(const TileRegs& t, float* __restrict__ S, int tid) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int idx = tid + 256 * i;
    const int r = idx >> 3, c = (idx & 7) << 2;
    *reinterpret_cast<float4*>(S + r * LDS_LD + c) = t.v[i];
  }
}

// acc += A_tile(64 rows of this wave) x B_tile(64 cols of this wave) over one BK=32 step
__device__ __forceinline__ void mfma_step(f32x16 (&acc)[2][2], const float* __restrict__ As,
                                          const float* __restrict__ Bs, int wm, int wn, int lane) {
  const int l31 = lane & 31, h = lane >> 5;
  float a[2][16], b[2][16];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const float* pa = As + (wm * 64 + t * 32 + l31) * LDS_LD + 16 * h;
    const float* pb = Bs + (wn * 64 + t * 32 + l31) * LDS_LD + 16 * h;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float4 va = *reinterpret_cast<const float4*>(pa + 4 * q);
      float4 vb = *reinterpret_cast<const float4*>(pb + 4 * q);
      a[t][4 * q + 0] = va.x; a[t][4 * q + 1] = va.y; a[t][4 * q + 2] = va.z; a[t][4 * q + 3] = va.w;
      b[t][4 * q + 0] = vb.x; b[t][4 * q + 1] = vb.y; b[t][4 * q + 2] = vb.z; b[t][4 * q + 3] = vb.w;
    }
  }
#pragma unroll
  for (int s = 0; s < 16; ++s) {
#pragma unroll
    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
      for (int tn = 0; tn < 2; ++tn)
        acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[tm][s], b[tn][s], acc[tm][tn], 0, 0, 0);
  }
}

template <bool FAST>
__device__ __forceinline__ void run_segment(f32x16 (&acc)[2][2], const float* __restrict__ A, int64_t lda,
                                            const float* __restrict__ B, int64_t ldb, int K, int M, int N,
                                            int bm, int bn, float* As, float* Bs, int tid, int wm, int wn,
                                            int lane) {
  TileRegs ra, rb;
  load_tile<FAST>(ra, A, lda, bm, M, 0, K, tid);
  load_tile<FAST>(rb, B, ldb, bn, N, 0, K, tid);
  for (int k0 = 0; k0 < K; k0 += BK) {
    __syncthreads();                       // previous step's fragment reads are done
    store_tile(ra, As, tid);
    store_tile(rb, Bs, tid);
    __syncthreads();
    if (k0 + BK < K) {                     // prefetch the next tile under this step's MFMAs
      load_tile<FAST>(ra, A, lda, bm, M, k0 + BK, K, tid);
      load_tile<FAST>(rb, B, ldb, bn, N, k0 + BK, K, tid);
    }
    mfma_step(acc, As, Bs, wm, wn, lane);
  }
}

template <bool FAST>
__global__ __launch_bounds__(256, 2) void gemm_f32_nt_kernel(GemmF32Args g) {
  __shared__ __attribute__((aligned(16))) float smem[2 * BM * LDS_LD];
  float* As = smem;
  float* Bs = smem + BM * LDS_LD;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int wm = w >> 1, wn = w & 1;

  // XCD-aware tile order: workgroups that share an XCD (blockIdx % 8, observed round-robin placement --
  // speed only) walk a contiguous run of tiles, N fastest, so they share A row panels in that XCD's L2.
  const int nwg = g.tiles_m * g.tiles_n;
  const int orig = blockIdx.x;
  const int q = nwg >> 3, rr = nwg & 7, xcd = orig & 7;
  const int wgid = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + (orig >> 3);
  const int bm = (wgid / g.tiles_n) * BM;
  const int bn = (wgid % g.tiles_n) * BN;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  if (g.K2 > 0) {  // the low-rank segment first, so its scaling applies to it alone (lora.py:52-53)
    run_segment<FAST>(acc, g.A2, g.lda2, g.B2, g.ldb2, g.K2, g.M, g.N, bm, bn, As, Bs, tid, wm, wn, lane);
    if (g.alpha2 != 1.0f) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int e = 0; e < 16; ++e) acc[i][j][e] *= g.alpha2;
    }
  }
  run_segment<FAST>(acc, g.A, g.lda, g.B, g.ldb, g.K, g.M, g.N, bm, bn, As, Bs, tid, wm, wn, lane);

  // epilogue: C/D map of the 32x32 MFMA: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
  const int l31 = lane & 31, h = lane >> 5;
#pragma unroll
  for (int tn = 0; tn < 2; ++tn) {
    const int n = bn + wn * 64 + tn * 32 + l31;
    if (n >= g.N) continue;
    const float bv = g.bias ? g.bias[n] : 0.f;
#pragma unroll
    for (int tm = 0; tm < 2; ++tm) {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int m = bm + wm * 64 + tm * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
        if (m < g.M) g.C[(int64_t)m * g.ldc + n] = acc[tm][tn][e] + bv;
      }
    }
  }
}

int launch_gemm_f32_nt(const float* A, int64_t lda, const float* B, int64_t ldb, int64_t K, const float* A2,
                       int64_t lda2, const float* B2, int64_t ldb2, int64_t K2, float alpha2,
                       const float* bias, float* C, int64_t ldc, int64_t M, int64_t N, hipStream_t st) {
  GemmF32Args g;
  g.A = A; g.B = B; g.A2 = A2; g.B2 = B2; g.bias = bias; g.C = C;
  g.lda = lda; g.ldb = ldb; g.lda2 = lda2; g.ldb2 = ldb2; g.ldc = ldc;
  g.M = (int)M; g.N = (int)N; g.K = (int)K; g.K2 = (int)K2; g.alpha2 = alpha2;
  g.tiles_m = (int)((M + BM - 1) / BM); g.tiles_n = (int)((N + BN - 1) / BN);
  bool fast = aligned16(A) && aligned16(B) && (lda % 4 == 0) && (ldb % 4 == 0);
  if (K2 > 0) fast = fast && aligned16(A2) && aligned16(B2) && (lda2 % 4 == 0) && (ldb2 % 4 == 0);
  const unsigned grid = (unsigned)(g.tiles_m * g.tiles_n);
  if (fast) gemm_f32_nt_kernel<true><<<grid, 256, 0, st>>>(g);
  else gemm_f32_nt_kernel<false><<<grid, 256, 0, st>>>(g);
  return check_launch("spq_gemm_f32_nt");
}

}  // namespace spq

extern "C" int spq_gemm_f32_nt(const float* A, int64_t lda, const float* B, int64_t ldb, int64_t K,
                               const float* A2, int64_t lda2, const float* B2, int64_t ldb2, int64_t K2,
                               float alpha2, const float* bias, float* C, int64_t ldc, int64_t M, int64_t N,
                               spq_stream_t stream) {
  using namespace spq;
  SPQ_REQUIRE(A && B && C, "spq_gemm_f32_nt: null pointer");
  SPQ_REQUIRE(M > 0 && N > 0 && K > 0, "spq_gemm_f32_nt: empty problem (%lld,%lld,%lld)", (long long)M,
              (long long)N, (long long)K);
  SPQ_REQUIRE(M < (1 << 30) && N < (1 << 30) && K < (1 << 30), "spq_gemm_f32_nt: dimension too large");
  SPQ_REQUIRE(lda >= K && ldb >= K && ldc >= N, "spq_gemm_f32_nt: leading dimension smaller than the row");
  SPQ_REQUIRE(K2 >= 0 && (K2 == 0 || (A2 && B2 && lda2 >= K2 && ldb2 >= K2)),
              "spq_gemm_f32_nt: bad second segment");
  return launch_gemm_f32_nt(A, lda, B, ldb, K, A2, lda2, B2, ldb2, K2, alpha2, bias, C, ldc, M, N,
                            (hipStream_t)stream);
}
