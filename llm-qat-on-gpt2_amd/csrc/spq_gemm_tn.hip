// Token-contraction ("TN") GEMM of the backward pass:  out[i, j] = alpha * sum_m P[m, i] * Q[m, j]
//   d/dA = s * x^T . (g . FQ(B)^T)   (P = x [M,K],  Q = g.FQ(B)^T [M,r])      lora.py:51 backward
//   d/dB = s * (x . FQ(A))^T . g     (P = x.FQ(A) [M,r],  Q = g [M,N])        lora.py:52 backward
// One side is LoRA-rank thin, M is long: the work is reading P and Q once (HBM-bound), so the contraction is split over M
// (grid.z slices x 4 waves) and reduced in a fixed order (deterministic, unlike atomics).
// v_mfma_f32_32x32x2_f32 wants A[i][k] with lane = i + 32 k and B[k][j] with lane = j + 32 k: for k = token that is one
// coalesced 128-B row segment of P (resp. Q) per half-wave -- fragments come straight from global memory, no LDS staging.
#include "spq_common.h"

namespace spq {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int TN_T = 64;   // block tile 64 x 64, every wave computes the whole tile over its own tokens

struct GemmTnArgs {
  const float* P; const float* Q;
  float* part;          // [S][I][J] partial sums (S > 1) or the output itself (S == 1)
  int64_t ldp, ldq, ldo;
  int M, I, J, rows_per_slice, S;
  float alpha;
};

__global__ __launch_bounds__(256) void gemm_tn_kernel(GemmTnArgs a) {
  __shared__ float red[4][32][TN_T];   // 32 KB
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, l31 = lane & 31, h = lane >> 5;
  const int i0 = blockIdx.x * TN_T, j0 = blockIdx.y * TN_T, s = blockIdx.z;
  const int m_begin = s * a.rows_per_slice;
  const int m_end = min(a.M, m_begin + a.rows_per_slice);
  const bool i_ok0 = i0 + l31 < a.I, i_ok1 = i0 + 32 + l31 < a.I;
  const bool j_ok0 = j0 + l31 < a.J, j_ok1 = j0 + 32 + l31 < a.J;
  const float* p = a.P + i0 + l31;
  const float* q = a.Q + j0 + l31;
  f32x16 acc00 = {0}, acc01 = {0}, acc10 = {0}, acc11 = {0};
  // tokens of this wave: m_begin + 8 u + 2 w + h, u = 0, 1, ...   (the 4 waves of a block read 8 adjacent rows)
  constexpr int U = 4;
  for (int mb = m_begin + 2 * w; mb < m_end; mb += 8 * U) {
    float a0[U], a1[U], b0[U], b1[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int m = mb + 8 * u + h;
      const bool ok = m < m_end;
      const float* pr = p + (int64_t)m * a.ldp;
      const float* qr = q + (int64_t)m * a.ldq;
      a0[u] = (ok && i_ok0) ? pr[0] : 0.f;
      a1[u] = (ok && i_ok1) ? pr[32] : 0.f;
      b0[u] = (ok && j_ok0) ? qr[0] : 0.f;
      b1[u] = (ok && j_ok1) ? qr[32] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      acc00 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[u], b0[u], acc00, 0, 0, 0);
      acc01 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[u], b1[u], acc01, 0, 0, 0);
      acc10 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[u], b0[u], acc10, 0, 0, 0);
      acc11 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[u], b1[u], acc11, 0, 0, 0);
    }
  }
  // C/D map of the 32x32 MFMA: column = lane & 31, row = (e & 3) + 8 (e >> 2) + 4 (lane >> 5).
  // The four per-wave partials meet in LDS, 32 tile rows at a time (32 KB), and are summed in a fixed order.
  float* dst = a.part + (a.S > 1 ? (int64_t)s * a.I * a.ldo : 0);
  const float scale = a.S > 1 ? 1.f : a.alpha;
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    if (half) __syncthreads();
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int row = (e & 3) + 8 * (e >> 2) + 4 * h;
      red[w][row][l31] = half ? acc10[e] : acc00[e];
      red[w][row][32 + l31] = half ? acc11[e] : acc01[e];
    }
    __syncthreads();
    for (int idx = tid; idx < 32 * TN_T; idx += 256) {
      const int r = idx >> 6, c = idx & 63, i = i0 + 32 * half + r;
      if (i < a.I && j0 + c < a.J) {
        const float v = (red[0][r][c] + red[1][r][c]) + (red[2][r][c] + red[3][r][c]);     // fixed order
        dst[(int64_t)i * a.ldo + j0 + c] = v * scale;
      }
    }
  }
}

// out = alpha * sum_s part[s]   (fixed order)
__global__ __launch_bounds__(256) void gemm_tn_reduce_kernel(const float* __restrict__ part, int S, int64_t n, float alpha,
                                                             float* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float v = part[i];
  for (int s = 1; s < S; ++s) v += part[(int64_t)s * n + i];
  out[i] = v * alpha;
}

static int tn_slices(int64_t M, int64_t I, int64_t J) {
  const int64_t tiles = ((I + TN_T - 1) / TN_T) * ((J + TN_T - 1) / TN_T);
  int64_t S = (1024 + tiles - 1) / tiles;                 // ~4 workgroups per CU
  const int64_t max_s = (M + 63) / 64;                    // at least 64 tokens per slice
  if (S > max_s) S = max_s;
  if (S > 256) S = 256;
  return (int)(S < 1 ? 1 : S);
}

}  // namespace spq

using namespace spq;

extern "C" size_t spq_gemm_f32_tn_workspace_bytes(int64_t M, int64_t I, int64_t J) {
  if (M <= 0 || I <= 0 || J <= 0) return 0;
  const int S = tn_slices(M, I, J);
  return (S > 1 ? (size_t)S * (size_t)I * (size_t)J * sizeof(float) : 0) + 256;
}

extern "C" int spq_gemm_f32_tn(const float* P, int64_t ldp, const float* Q, int64_t ldq, int64_t M, int64_t I, int64_t J,
                               float alpha, float* out, void* workspace, size_t workspace_bytes, spq_stream_t stream) {
  SPQ_REQUIRE(P && Q && out, "spq_gemm_f32_tn: null pointer");
  SPQ_REQUIRE(M > 0 && I > 0 && J > 0 && ldp >= I && ldq >= J, "spq_gemm_f32_tn: bad shape M=%lld I=%lld J=%lld",
              (long long)M, (long long)I, (long long)J);
  SPQ_REQUIRE(M < (1 << 30) && I < (1 << 30) && J < (1 << 30), "spq_gemm_f32_tn: dimension too large");
  const int S = tn_slices(M, I, J);
  if (S > 1 && (!workspace || workspace_bytes < spq_gemm_f32_tn_workspace_bytes(M, I, J) || !aligned16(workspace))) {
    set_error("spq_gemm_f32_tn: workspace %zu B < required %zu B", workspace_bytes, spq_gemm_f32_tn_workspace_bytes(M, I, J));
    return SPQ_ERR_WORKSPACE;
  }
  hipStream_t st = (hipStream_t)stream;
  GemmTnArgs a;
  a.P = P; a.Q = Q; a.ldp = ldp; a.ldq = ldq; a.ldo = J;
  a.M = (int)M; a.I = (int)I; a.J = (int)J; a.S = S; a.alpha = alpha;
  a.rows_per_slice = (int)(((M + S - 1) / S + 7) / 8 * 8);
  a.part = S > 1 ? (float*)workspace : out;
  dim3 grid((unsigned)((I + TN_T - 1) / TN_T), (unsigned)((J + TN_T - 1) / TN_T), (unsigned)S);
  gemm_tn_kernel<<<grid, 256, 0, st>>>(a);
  int rc = check_launch("spq_gemm_f32_tn");
  if (rc || S == 1) return rc;
  const int64_t n = I * J;
  gemm_tn_reduce_kernel<<<(unsigned)((n + 255) / 256), 256, 0, st>>>((const float*)workspace, S, n, alpha, out);
  return check_launch("spq_gemm_f32_tn(reduce)");
}
