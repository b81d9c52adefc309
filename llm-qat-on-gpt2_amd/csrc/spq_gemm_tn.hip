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

constexpr int TN_I = 64, TN_J = 128;   // block tile; every wave computes the whole tile over its own tokens
constexpr int TN_WAVES = 8;

struct GemmTnArgs {
  const float* P; const float* Q;
  float* part;          // [S][I][J] partial sums (S > 1) or the output itself (S == 1)
  int64_t ldp, ldq, ldo;
  int M, I, J, rows_per_slice, S;
  float alpha;
};

// Fragment trick: lane (l, h) of a wave loads P[m + h][i0 + 2l .. +1] (8 B) and Q[m + h][j0 + 4l .. +3] (16 B); component c
// of the load is the MFMA operand of the tile whose rows / columns are {2l + c} resp. {4l + c} -- a fixed permutation of
// the block tile, undone when the accumulators are written out.  2 x 4 MFMAs per pair of tokens and 24 B per lane.
__global__ __launch_bounds__(64 * TN_WAVES) void gemm_tn_kernel(GemmTnArgs a) {
  __shared__ float red[TN_WAVES][8][TN_J];          // 32 KB: 8 tile rows of every wave's partial at a time
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, l31 = lane & 31, h = lane >> 5;
  const int i0 = blockIdx.x * TN_I, j0 = blockIdx.y * TN_J, s = blockIdx.z;
  const int m_begin = s * a.rows_per_slice;
  const int m_end = min(a.M, m_begin + a.rows_per_slice);
  const bool vec_p = (a.ldp & 1) == 0 && aligned16(a.P) && i0 + TN_I <= a.I;
  const bool vec_q = (a.ldq & 3) == 0 && aligned16(a.Q) && j0 + TN_J <= a.J;
  f32x16 acc[2][4];
#pragma unroll
  for (int ci = 0; ci < 2; ++ci)
#pragma unroll
    for (int cj = 0; cj < 4; ++cj)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[ci][cj][e] = 0.f;

  // tokens of this wave: m_begin + 2 w + 16 u + h.  Two register sets: the loads of batch b + 1 are issued before the
  // MFMAs of batch b (2 x 4 x U MFMAs of 64 cycles cover the HBM latency of the next batch).
  constexpr int U = 4;
  float2 pv[2][U]; float4 qv[2][U];
  auto load_batch = [&](int mb, float2 (&pb)[U], float4 (&qb)[U]) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int m = mb + 2 * TN_WAVES * u + h;
      pb[u] = make_float2(0.f, 0.f); qb[u] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (m < m_end) {
        const float* pr = a.P + (int64_t)m * a.ldp + i0 + 2 * l31;
        const float* qr = a.Q + (int64_t)m * a.ldq + j0 + 4 * l31;
        if (vec_p) pb[u] = *reinterpret_cast<const float2*>(pr);
        else {
          if (i0 + 2 * l31 < a.I) pb[u].x = pr[0];
          if (i0 + 2 * l31 + 1 < a.I) pb[u].y = pr[1];
        }
        if (vec_q) qb[u] = *reinterpret_cast<const float4*>(qr);
        else {
          if (j0 + 4 * l31 + 0 < a.J) qb[u].x = qr[0];
          if (j0 + 4 * l31 + 1 < a.J) qb[u].y = qr[1];
          if (j0 + 4 * l31 + 2 < a.J) qb[u].z = qr[2];
          if (j0 + 4 * l31 + 3 < a.J) qb[u].w = qr[3];
        }
      }
    }
  };
  auto mfma_batch = [&](const float2 (&pb)[U], const float4 (&qb)[U]) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(pb[u].x, qb[u].x, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(pb[u].x, qb[u].y, acc[0][1], 0, 0, 0);
      acc[0][2] = __builtin_amdgcn_mfma_f32_32x32x2f32(pb[u].x, qb[u].z, acc[0][2], 0, 0, 0);
      acc[0][3] = __builtin_amdgcn_mfma_f32_32x32x2f32(pb[u].x, qb[u].w, acc[0][3], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(pb[u].y, qb[u].x, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(pb[u].y, qb[u].y, acc[1][1], 0, 0, 0);
      acc[1][2] = __builtin_amdgcn_mfma_f32_32x32x2f32(pb[u].y, qb[u].z, acc[1][2], 0, 0, 0);
      acc[1][3] = __builtin_amdgcn_mfma_f32_32x32x2f32(pb[u].y, qb[u].w, acc[1][3], 0, 0, 0);
    }
  };
  constexpr int STEP = 2 * TN_WAVES * U;
  int mb = m_begin + 2 * w;
  load_batch(mb, pv[0], qv[0]);
  while (true) {
    load_batch(mb + STEP, pv[1], qv[1]);          // past m_end: zeros, no loads
    mfma_batch(pv[0], qv[0]);
    mb += STEP;
    if (mb >= m_end) break;
    load_batch(mb + STEP, pv[0], qv[0]);
    mfma_batch(pv[1], qv[1]);
    mb += STEP;
    if (mb >= m_end) break;
  }
  // C/D map of the 32x32 MFMA: column = lane & 31, row = (e & 3) + 8 (e >> 2) + 4 (lane >> 5).  Accumulator (ci, cj), MFMA
  // row rr, column cc is output (i0 + 2 rr + ci, j0 + 4 cc + cj).  Round (ci, g): the 8 MFMA rows 8 g' ... that share
  // e >> 2 == g, i.e. rr = 8 g + 4 h + (e & 3); the TN_WAVES partials meet in LDS and are summed in a fixed order.
  float* dst = a.part + (a.S > 1 ? (int64_t)s * a.I * a.ldo : 0);
  const float scale = a.S > 1 ? 1.f : a.alpha;
#pragma unroll
  for (int ci = 0; ci < 2; ++ci)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      __syncthreads();
#pragma unroll
      for (int e4 = 0; e4 < 4; ++e4) {
        const int lr = 4 * h + e4;                     // local row 0..7 <-> MFMA row rr = 8 g + lr
#pragma unroll
        for (int cj = 0; cj < 4; ++cj) red[w][lr][4 * l31 + cj] = acc[ci][cj][4 * g + e4];
      }
      __syncthreads();
      for (int idx = tid; idx < 8 * TN_J; idx += 64 * TN_WAVES) {
        const int lr = idx >> 7, c = idx & (TN_J - 1);
        const int i = i0 + 2 * (8 * g + lr) + ci, j = j0 + c;
        if (i < a.I && j < a.J) {
          float v = red[0][lr][c];
#pragma unroll
          for (int q = 1; q < TN_WAVES; ++q) v += red[q][lr][c];          // fixed order
          dst[(int64_t)i * a.ldo + j] = v * scale;
        }
      }
    }
}

// out = alpha * sum_s part[s]   (fixed order; four independent loads in flight per thread)
__global__ __launch_bounds__(256) void gemm_tn_reduce_kernel(const float* __restrict__ part, int S, int64_t n, float alpha,
                                                             float* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float v = part[i];
  int s = 1;
  for (; s + 3 < S; s += 4) {
    const float v0 = part[(int64_t)s * n + i], v1 = part[(int64_t)(s + 1) * n + i];
    const float v2 = part[(int64_t)(s + 2) * n + i], v3 = part[(int64_t)(s + 3) * n + i];
    v = (((v + v0) + v1) + v2) + v3;
  }
  for (; s < S; ++s) v += part[(int64_t)s * n + i];
  out[i] = v * alpha;
}

static int tn_slices(int64_t M, int64_t I, int64_t J) {
  const int64_t tiles = ((I + TN_I - 1) / TN_I) * ((J + TN_J - 1) / TN_J);
  int64_t S = 512 / tiles;                                // <= 2 full rounds of one 8-wave workgroup per CU
  const int64_t max_s = (M + 16 * TN_WAVES - 1) / (16 * TN_WAVES);   // at least 16 tokens per wave
  if (S > max_s) S = max_s;
  if (S > 128) S = 128;
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
  a.rows_per_slice = (int)(((M + S - 1) / S + 15) / 16 * 16);
  a.part = S > 1 ? (float*)workspace : out;
  dim3 grid((unsigned)((I + TN_I - 1) / TN_I), (unsigned)((J + TN_J - 1) / TN_J), (unsigned)S);
  gemm_tn_kernel<<<grid, 64 * TN_WAVES, 0, st>>>(a);
  int rc = check_launch("spq_gemm_f32_tn");
  if (rc || S == 1) return rc;
  const int64_t n = I * J;
  gemm_tn_reduce_kernel<<<(unsigned)((n + 255) / 256), 256, 0, st>>>((const float*)workspace, S, n, alpha, out);
  return check_launch("spq_gemm_f32_tn(reduce)");
}
