// The activation pass with the LoRA-down product on the f16 matrix pipe (SPQ_LORA_DOWN_F16).
// Moved out of the product library in round 3 (measured slower than the default kernels); include AFTER
// llm-qat-on-gpt2_amd/csrc/spq_f16x2.hip (tools/gemm_bench.hip does).  Not built into libspq.so, not reachable from the C ABI.
#pragma once
namespace spq {
// -------------------------------------------------------------------------------------------------------------------
// The same pass with the LoRA-down product on the f16 matrix pipe (fp32-input MFMA runs at 1/16 of its rate and was half of
// this kernel's time at K = 3072).  x has no calibrated bound, so each 32-row panel gets per-row powers of two 2^g[m] from
// its own row maxima (one LDS sweep), x * 2^g and FQ(A)^T * 2^S are split into two fp16 limbs on the fly (S from the LoRA-A
// quantizer's range, a.ascale), three v_mfma_f32_32x32x16_f16 per k-step of 16 (hi.hi, hi.lo, lo.hi), and the panel's
// partial product is folded into an fp32 running sum with 2^-g[m] 2^-S before the next panel re-scales.  Waves 0..3 own the
// four k-steps of a 64-column chunk; all 8 waves do the level pass.
// -------------------------------------------------------------------------------------------------------------------
constexpr int XP16_LDS = XP_LDS + 256;                     // + per-row scales of the current panel
__global__ __launch_bounds__(512) void xpass_panel16_kernel(XPassArgs a) {
  extern __shared__ __attribute__((aligned(16))) char xsm[];
  char* xs = xsm;
  char* as = xsm + XP_XS;
  float* sxs = reinterpret_cast<float*>(xsm + XP_XS + XP_NAS * XP_AS);
  float* rs = reinterpret_cast<float*>(xsm + XP_LDS);      // [32] 2^g[m], [32] 2^-g[m] * 2^-S
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m0 = blockIdx.x * XR;
  const float qhi = (float)((1 << (a.bits - 1)) - 1), qlo = -qhi;
  const float pscale = a.limbs ? a.xscale[0] : 1.f;
  const bool with_lora = a.r > 0;
  const int l31 = lane & 31, h = lane >> 5;
  const float a_mul = a.ascale[0], a_inv = a.ascale[1];

  f32x16 acc[2], tsum[2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) { acc[i][e] = 0.f; tsum[i][e] = 0.f; }

  const int prow = w * 4 + (lane >> 4), ppos = lane & 15;
  const float* x_src = a.x + (int64_t)min(m0 + prow, a.M - 1) * a.K + ((ppos ^ (prow & 15)) << 2);
  const int q_row = tid >> 4, q_pos = tid & 15;
  const int q_kof = (q_pos ^ (q_row & 15)) << 2;
  const int64_t q_dst = (int64_t)min(m0 + q_row, a.M - 1) * a.Kp + q_kof;

  // FQ(A)^T chunk g lives in buffer g & 1 as two fp16 planes [64 j][64 k] (hi at +0, lo at +8 KB); the 16-B piece c of row
  // j sits at position c ^ ((j >> 1) & 7).  Staged through registers as in xpass_panel_kernel, converted on the way.
  const int total_chunks = a.K / 64;
  // Two register sets: chunk n travels in set n & 1, loaded two iterations before it is written to LDS, so the L2 latency
  // of the FQ(A)^T rows (~1 us, longer than one chunk's work) never sits on the per-chunk critical path.
  // (named registers and a chunk loop unrolled by two: a run-time set index makes hipcc copy the sets around and wait for
  // loads it has just issued)
  float4 ra0_0, ra1_0, ra0_1, ra1_1;
  ra0_0 = ra1_0 = ra0_1 = ra1_1 = make_float4(0.f, 0.f, 0.f, 0.f);
  const int a_r = tid >> 4, a_c = tid & 15;                // row (and row + 32), 4 k at 4 a_c
  const float* a_src = a.aT + (int64_t)a_r * a.K + (a_c << 2);
  const int64_t a_step = (int64_t)32 * a.K;
  const int a_dst0 = a_r * 128 + ((((a_c >> 1) ^ ((a_r >> 1) & 7)) << 4) | ((a_c & 1) << 3));
  const int a_dst1 = (a_r + 32) * 128 + ((((a_c >> 1) ^ (((a_r + 32) >> 1) & 7)) << 4) | ((a_c & 1) << 3));
#define SPQ_STORE_A(buf, R0, R1)                                                                          \
  do {                                                                                                    \
    char* d_ = as + (buf) * XP_AS;                                                                        \
    union { _Float16 hh[4]; uint2 u; } hi_, lo_;                                                          \
    split2((R0).x * a_mul, hi_.hh[0], lo_.hh[0]); split2((R0).y * a_mul, hi_.hh[1], lo_.hh[1]);           \
    split2((R0).z * a_mul, hi_.hh[2], lo_.hh[2]); split2((R0).w * a_mul, hi_.hh[3], lo_.hh[3]);           \
    *reinterpret_cast<uint2*>(d_ + a_dst0) = hi_.u; *reinterpret_cast<uint2*>(d_ + 8192 + a_dst0) = lo_.u; \
    split2((R1).x * a_mul, hi_.hh[0], lo_.hh[0]); split2((R1).y * a_mul, hi_.hh[1], lo_.hh[1]);           \
    split2((R1).z * a_mul, hi_.hh[2], lo_.hh[2]); split2((R1).w * a_mul, hi_.hh[3], lo_.hh[3]);           \
    *reinterpret_cast<uint2*>(d_ + a_dst1) = hi_.u; *reinterpret_cast<uint2*>(d_ + 8192 + a_dst1) = lo_.u; \
  } while (0)
#define SPQ_LOAD_A(n, R0, R1)                                                          \
  do {                                                                                 \
    if ((n) < total_chunks) {                                                          \
      (R0) = *reinterpret_cast<const float4*>(a_src + (n) * 64);                       \
      (R1) = *reinterpret_cast<const float4*>(a_src + a_step + (n) * 64);              \
    }                                                                                  \
  } while (0)

  int gc = 0;
  if (with_lora) {
    SPQ_LOAD_A(0, ra0_0, ra1_0);
    SPQ_LOAD_A(1, ra0_1, ra1_1);
  }
  for (int p0 = 0; p0 < a.K; p0 += XP_CHUNKS * 64) {
    const int nch = min(XP_CHUNKS, (a.K - p0) / 64);
    for (int c = 0; c < nch; ++c) glds16(x_src + p0 + c * 64, xs + c * (XR * 256) + w * 1024);
    for (int k = tid; k < nch * 64; k += 512) {
      sxs[k] = a.x_pc ? a.sx[p0 + k] : a.sx[0];
      sxs[XP_CHUNKS * 64 + k] = (a.limbs || a.lora_fq) ? (a.x_pc ? a.zx[p0 + k] : a.zx[0]) : 0.f;
    }
    if (with_lora && p0 == 0) {
      SPQ_STORE_A(0, ra0_0, ra1_0);
      SPQ_LOAD_A(2, ra0_0, ra1_0);
    }
    __syncthreads();                                       // the panel landed; FQ(A)^T chunk gc is in LDS
    if (with_lora) {                                       // per-row scale of this panel: 16 lanes per row sweep its chunks
      float mx = 0.f;
      for (int c = 0; c < nch; ++c) {
        const float4 v = *reinterpret_cast<const float4*>(xs + c * (XR * 256) + q_row * 256 + q_pos * 16);
        mx = fmaxf(fmaxf(mx, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
      }
      mx = fmaxf(mx, __shfl_xor(mx, 1, 64)); mx = fmaxf(mx, __shfl_xor(mx, 2, 64));
      mx = fmaxf(mx, __shfl_xor(mx, 4, 64)); mx = fmaxf(mx, __shfl_xor(mx, 8, 64));
      if (q_pos == 0) { const float p = pow2_scale_for(mx); rs[q_row] = p; rs[32 + q_row] = (1.0f / p) * a_inv; }
      __syncthreads();
    }
    // chunk gc reads FQ(A)^T from LDS buffer gc & 1; gc is even at every panel start (XP_CHUNKS is even)
    auto chunk = [&](int c, auto par_tag) {
      constexpr int PAR = decltype(par_tag)::value;

      const int k0 = p0 + c * 64;
      const bool next_a = with_lora && gc + 1 < total_chunks;
      {
        const float4 v = *reinterpret_cast<const float4*>(xs + c * (XR * 256) + q_row * 256 + q_pos * 16);
        const float4 sc = *reinterpret_cast<const float4*>(sxs + c * 64 + q_kof);
        const float4 zp = *reinterpret_cast<const float4*>(sxs + XP_CHUNKS * 64 + c * 64 + q_kof);
        store_act4(a, q_dst + k0, v, sc, zp, qlo, qhi, pscale);
      }
      if (with_lora && w < 4) {                            // k-step w of this chunk: k = 16 w + 8 h .. + 7
        const int s0 = 4 * w + 2 * h;
        const char* xrow = xs + c * (XR * 256) + l31 * 256;
        float4 x0 = *reinterpret_cast<const float4*>(xrow + ((s0 ^ (l31 & 15)) << 4));
        float4 x1 = *reinterpret_cast<const float4*>(xrow + (((s0 + 1) ^ (l31 & 15)) << 4));
        if (a.lora_fq) {
          const float* scp = sxs + c * 64 + 16 * w + 8 * h;
          x0 = fq_act4(a, x0, *reinterpret_cast<const float4*>(scp), *reinterpret_cast<const float4*>(scp + XP_CHUNKS * 64));
          x1 = fq_act4(a, x1, *reinterpret_cast<const float4*>(scp + 4), *reinterpret_cast<const float4*>(scp + XP_CHUNKS * 64 + 4));
        }
        const float rsc = rs[l31];
        union { _Float16 hh[8]; f16x8 v; } xh, xl;
        split2(x0.x * rsc, xh.hh[0], xl.hh[0]); split2(x0.y * rsc, xh.hh[1], xl.hh[1]);
        split2(x0.z * rsc, xh.hh[2], xl.hh[2]); split2(x0.w * rsc, xh.hh[3], xl.hh[3]);
        split2(x1.x * rsc, xh.hh[4], xl.hh[4]); split2(x1.y * rsc, xh.hh[5], xl.hh[5]);
        split2(x1.z * rsc, xh.hh[6], xl.hh[6]); split2(x1.w * rsc, xh.hh[7], xl.hh[7]);
        const char* ab = as + PAR * XP_AS;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const int j = t * 32 + l31;
          const int off = j * 128 + (((2 * w + h) ^ ((j >> 1) & 7)) << 4);
          const f16x8 bh = *reinterpret_cast<const f16x8*>(ab + off);
          const f16x8 bl = *reinterpret_cast<const f16x8*>(ab + 8192 + off);
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xh.v, bh, acc[t], 0, 0, 0);
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xh.v, bl, acc[t], 0, 0, 0);
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xl.v, bh, acc[t], 0, 0, 0);
        }
      }
      if (next_a) {                                        // the other LDS buffer: every wave left it at the last barrier
        if (PAR == 0) { SPQ_STORE_A(1, ra0_1, ra1_1); SPQ_LOAD_A(gc + 3, ra0_1, ra1_1); }
        else { SPQ_STORE_A(0, ra0_0, ra1_0); SPQ_LOAD_A(gc + 3, ra0_0, ra1_0); }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
          ++gc;
    };
    {
      int c = 0;
      for (; c + 1 < nch; c += 2) { chunk(c, std::integral_constant<int, 0>{}); chunk(c + 1, std::integral_constant<int, 1>{}); }
      if (c < nch) chunk(c, std::integral_constant<int, 0>{});
    }
    if (with_lora) {                                       // fold this panel's product into the fp32 sum: * 2^-g[m] 2^-S
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          tsum[t][e] += acc[t][e] * rs[32 + (e & 3) + 8 * (e >> 2) + 4 * h];
          acc[t][e] = 0.f;
        }
    }
    __syncthreads();                                       // panel images and scales are free for the next panel
  }
#undef SPQ_LOAD_A
#undef SPQ_STORE_A
  if (!with_lora) return;
  xpass_finish<2, 8>(a, tsum, reinterpret_cast<float*>(xsm), m0, tid);
}

}  // namespace spq
