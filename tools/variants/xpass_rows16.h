// The 16-row panel form of the activation pass (with the in-pass weight-row variant, SPQ_PREP_INPASS).
// Moved out of the product library in round 3 (measured slower than the default kernels); include AFTER
// llm-qat-on-gpt2_amd/csrc/spq_f16x2.hip (tools/gemm_bench.hip does).  Not built into libspq.so, not reachable from the C ABI.
#pragma once
namespace spq {
template <int PREP>   // 0: no weight rows; 1: fp16 limb rows; 2: int8 level rows (SPQ_PATH_I8)
__global__ __launch_bounds__(256, SPQ_XP16R_CH <= 4 ? 3 : 2) void xpass_rows16_kernel(XPassArgs a, PrepArgs pa, int prep_rows) {
  extern __shared__ __attribute__((aligned(16))) char xsm[];
  constexpr int CH = XP16R_CH;
  char* xs = xsm;
  char* as = xsm + XP16R_XS;
  float* sxs = reinterpret_cast<float*>(xsm + XP16R_XS + XP_NAS * XP_AS);
  float* lnst = reinterpret_cast<float*>(xsm + XP16R_XS + XP_NAS * XP_AS + 2 * XP16R_CH * 64 * 4);
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m0 = blockIdx.x * XR16;
  if (a.ln_w) ln_panel_stats(a, m0, XR16, lnst);            // visible to every thread after the first panel's barrier
  const float qhi = (float)((1 << (a.bits - 1)) - 1), qlo = -qhi;
  const float pscale = a.limbs ? a.xscale[0] : 1.f;
  const bool with_lora = a.r > 0;
  const int l15 = lane & 15, q4 = lane >> 4;
  __shared__ float s_qn[256];
  const float* qn_lut = (a.limbs || a.lora_fq) ? fill_log_qn_lut(s_qn, a.bits, a.qtype, a.symmetric) : nullptr;   // (uniform branch)

  f32x4 acc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int e = 0; e < 4; ++e) acc[i][e] = 0.f;

  // x copy: a 64-column chunk of the 16-row panel is 4 pieces of 1 KB (4 rows x 256 B); wave w issues piece w
  const int prow = w * 4 + (lane >> 4), ppos = lane & 15;
  const float* x_src = a.x + (int64_t)min(m0 + prow, a.M - 1) * a.K + ((ppos ^ (prow & 15)) << 2);
  const int q_row = tid >> 4, q_pos = tid & 15;
  const int q_kof = (q_pos ^ (q_row & 15)) << 2;
  const int64_t q_dst = (int64_t)min(m0 + q_row, a.M - 1) * a.Kp + q_kof;

  // FQ(A)^T chunk [64 j x 64 k] fp32: thread -> rows a_r, a_r+16, a_r+32, a_r+48, 16-B source chunk a_c; register staged
  const int total_chunks = a.K / 64;
  float4 ra[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) ra[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  const int a_r = tid >> 4, a_c = tid & 15;
  const float* a_src = a.aT + (int64_t)a_r * a.K + (a_c << 2);
  const int64_t a_step = (int64_t)16 * a.K;
  const int a_dst = a_r * 256 + ((a_c ^ (a_r & 15)) << 4);       // (a_r + 16 i) & 15 == a_r & 15
#define SPQ_LOAD_A(k0)                                                                                       \
  do {                                                                                                       \
    ra[0] = *reinterpret_cast<const float4*>(a_src + (k0));                                                  \
    ra[1] = *reinterpret_cast<const float4*>(a_src + a_step + (k0));                                         \
    ra[2] = *reinterpret_cast<const float4*>(a_src + 2 * a_step + (k0));                                     \
    ra[3] = *reinterpret_cast<const float4*>(a_src + 3 * a_step + (k0));                                     \
  } while (0)
#define SPQ_STORE_A(buf)                                                                                     \
  do {                                                                                                       \
    char* d_ = as + (buf) * XP_AS + a_dst;                                                                   \
    *reinterpret_cast<float4*>(d_) = ra[0];                                                                  \
    *reinterpret_cast<float4*>(d_ + 16 * 256) = ra[1];                                                       \
    *reinterpret_cast<float4*>(d_ + 32 * 256) = ra[2];                                                       \
    *reinterpret_cast<float4*>(d_ + 48 * 256) = ra[3];                                                       \
  } while (0)

  int gc = 0;
  if (with_lora) SPQ_LOAD_A(0);
  for (int p0 = 0; p0 < a.K; p0 += CH * 64) {
    const int nch = min(CH, (a.K - p0) / 64);
    for (int c = 0; c < nch; ++c)
      if (!(SPQ_XP_DIAG & 8) || (p0 == 0 && c == 0)) glds16(x_src + p0 + c * 64, xs + c * (XR16 * 256) + w * 1024);
    if (PREP && p0 == 0) {
      // weight rows of this workgroup, while the first panel's copies are in flight (their latency and the rows' load latency
      // overlap).  The rows' LoRA-B columns B[j][n0 .. n0+nrows) are short contiguous runs: staged [row][j] in LDS (the second
      // FQ(A)^T buffer is not in use yet) instead of one strided scalar read per (row, j) from every wave.
      const int n0 = (int)blockIdx.x * prep_rows;
      const int np = (pa.N + GN - 1) / GN * GN;
      const int nrows = min(prep_rows, np - n0);
      const float* sbw = nullptr;
      if (pa.B && nrows > 0) {
        float* sB = reinterpret_cast<float*>(as + XP_AS);
        const int nbv = min(nrows, pa.N - n0);
        for (int e = threadIdx.x; e < pa.r * nbv; e += 256) {
          const int j = e / nbv, i = e - j * nbv;
          sB[i * pa.Rp + j] = pa.B[(int64_t)j * pa.N + n0 + i];
        }
        __syncthreads();
        sbw = sB;
      }
      for (int i = (int)(threadIdx.x >> 6); i < nrows; i += 4)
        prep_row_wave<(PREP == 1 ? 0 : 1), 4>(pa, n0 + i, threadIdx.x & 63, (sbw && n0 + i < pa.N) ? sbw + i * pa.Rp : nullptr);
    }
    for (int k = tid; k < nch * 64; k += 256) {
      sxs[k] = a.x_pc ? a.sx[p0 + k] : a.sx[0];
      sxs[CH * 64 + k] = (a.limbs || a.lora_fq) ? (a.x_pc ? a.zx[p0 + k] : a.zx[0]) : 0.f;
    }
    if (with_lora && p0 == 0) SPQ_STORE_A(0);
    __syncthreads();                                       // vmcnt(0): the panel landed; FQ(A)^T chunk gc is in LDS
    if (a.ln_w) { ln_panel_apply(a, xs, XR16 * 256, nch, p0, q_row, q_pos, q_kof, lnst); __syncthreads(); }
    for (int c = 0; c < nch; ++c, ++gc) {
      const int k0 = p0 + c * 64;
      const bool next_a = with_lora && gc + 1 < total_chunks && !(SPQ_XP_DIAG & 1);
      if (next_a) SPQ_LOAD_A((gc + 1) * 64);
      if (!(SPQ_XP_DIAG & 4) || a.M == 12345) {
        const float4 v = *reinterpret_cast<const float4*>(xs + c * (XR16 * 256) + q_row * 256 + q_pos * 16);
        const float4 sc = *reinterpret_cast<const float4*>(sxs + c * 64 + q_kof);
        const float4 zp = *reinterpret_cast<const float4*>(sxs + CH * 64 + c * 64 + q_kof);
        store_act4(a, q_dst + k0, v, sc, zp, qlo, qhi, pscale, qn_lut);
      }
      if (with_lora) {
        const int pa = 4 * w + q4;                         // 16-B source chunk of this lane: k = 16 w + 4 q4 .. + 3
        float4 av = *reinterpret_cast<const float4*>(xs + c * (XR16 * 256) + l15 * 256 + ((pa ^ l15) << 4));
        if (a.lora_fq)
          av = fq_act4(a, av, *reinterpret_cast<const float4*>(sxs + c * 64 + 4 * pa),
                       *reinterpret_cast<const float4*>(sxs + CH * 64 + c * 64 + 4 * pa), qn_lut);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const int rb = t * 16 + l15;
          const float4 bv = *reinterpret_cast<const float4*>(as + (gc & 1) * XP_AS + rb * 256 + ((pa ^ (rb & 15)) << 4));
          if (SPQ_XP_DIAG & 2) { acc[t][0] += av.x + bv.x; continue; }
          acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, bv.x, acc[t], 0, 0, 0);
          acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, bv.y, acc[t], 0, 0, 0);
          acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.z, bv.z, acc[t], 0, 0, 0);
          acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.w, bv.w, acc[t], 0, 0, 0);
        }
        if (next_a) SPQ_STORE_A((gc + 1) & 1);
      }
      if (!(SPQ_XP_DIAG & 16)) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
      }
    }
    __syncthreads();
  }
#undef SPQ_LOAD_A
#undef SPQ_STORE_A
  if (!with_lora) return;
  xpass16_finish(a, acc, reinterpret_cast<float*>(xsm), m0, tid);
}

}  // namespace spq
