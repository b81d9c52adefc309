// The contraction on 256 x 128 tiles, two workgroups per CU (included by spq_f16x2.hip after the 128 x 128 kernel).
//
// Why.  In-kernel stamps of gemm_f16x2_t128_kernel (tools/t128_bench, T128_DIAG) show what bounds it at the headline shape: the
// CU's vector-memory path.  A 128 x 128 tile pulls 48 KB through global -> LDS copies per 64-deep stage and pushes 64 KB of
// output: 4.3 MB per CU and launch at the ~45 B/clk a CU sustains for these copies = 100 k of the 126 k cycles a wave lives, and
// the epilogue's stores run at the same ~40 B/clk per CU (batching its LDS transposes changed nothing).  The matrix pipe needs
// 83 k.  So the lever is bytes per MFMA: a 256 x 128 tile shares each weight stage (hi + lo limb, 32 KB) between two 128-row
// halves: 64 KB per stage for twice the MFMAs, 2/3 of the copy bytes per output.
//
// Shape.  4 waves (2 x 2), each 128 x 64 outputs (TM = 8 blocks of 16 rows) -- 128 accumulator registers, so two waves per SIMD
// and two workgroups per CU (64 KB of LDS each: ONE stage buffer, as in the 128 x 128 kernel; the epilogue's transpose slices live
// inside it).  A stage is 128 MFMAs per wave (2048 cycles) against 1024 there, its copies 16 pieces per wave against 12: the
// matrix phase of one workgroup now covers the copy phase of the other.
//
// Work list.  2 workgroups x 256 CUs = 512 slots; M = 8192, N = 3072 is 768 such tiles = 1.5 rounds.  So the last row bands
// are cut into 128 x 128 HALF tiles (the same code with TM = 4: the waves are 64 x 64) and every workgroup gets whole tiles first,
// half tiles last, balanced by a static rule (t256_plan on the host picks the number of bands to cut; the kernel derives its own
// items from {n_full, n_half}).  At the headline shape: 504 whole + 528 half tiles = 3 units of 128 x 128 for each of 512 workgroups.
#pragma once

namespace spq {

constexpr int T256_STAGE_A = 256 * GK * 2;                 // 32 KB (a half tile uses the first 16 KB)
constexpr int T256_STAGE = T256_STAGE_A + 2 * STAGE_B;     // 64 KB
constexpr int T256_LDS = T256_STAGE;
#ifndef T256_DIAG       // tools/t256_bench only: 1 = no copies after a tile's first stage, 2 = no MFMAs, 4 = no stores, 8 = stamps
#define T256_DIAG 0
#endif
#ifndef T256_PRIO_PERIOD   // the CU's two workgroups swap issue priority every T256_PRIO_PERIOD stages (0: leave the arbiter alone)
#define T256_PRIO_PERIOD 7
#endif
#ifndef T256_RING
#define T256_RING 1
#endif
#ifndef T256_GROUP_M    // whole-tile rows (256 rows each) per L2 band of the tile order
#define T256_GROUP_M 4
#endif

struct T256Plan {
  int n_full;           // whole tiles: the first full_bands * tiles_n (256-row bands from the top)
  int n_half;           // half tiles: the remaining rows, 128 x 128 each
  int full_bands;
  int grid;
};

template <int TM, int AL, int EPI>
__device__ __forceinline__ void t256_epilogue(const GemmF16Args& g, char* smem, f32x4 (&acc)[TM][4], const int bm, const int bn, const int w,
                                              const int lane, const float out_scale) {
  constexpr int WROWS = TM * 16;
  const int wm = w >> 1, wn = w & 1;
  const int l15 = lane & 15, q4 = lane >> 4;
  // epilogue: per-wave transpose slices inside the (now free) stage buffer, whole 128-B lines per store
  float4 ep_rs[2], ep_bv[2];
#pragma unroll
  for (int tn = 0; tn < 2; ++tn) {
    const int n = bn + wn * 64 + tn * 32 + (lane & 7) * 4;
    ep_rs[tn] = make_float4(0.f, 0.f, 0.f, 0.f); ep_bv[tn] = ep_rs[tn];
    if (n < g.N) {
      ep_rs[tn] = *reinterpret_cast<const float4*>(g.rowscale + n);
      if (AL == 2) { ep_rs[tn].x *= out_scale; ep_rs[tn].y *= out_scale; ep_rs[tn].z *= out_scale; ep_rs[tn].w *= out_scale; }
      if (g.bias) ep_bv[tn] = *reinterpret_cast<const float4*>(g.bias + n);
    }
  }
  {
    char* eb = smem + w * (T256_STAGE / 4);                  // 16 KB per wave: four 16 x 32 blocks per round
    const int c4 = (lane & 7) * 4;
    const bool interior = (bm + 32 * TM <= g.M) && (bn + GN <= g.N);
#pragma unroll
    for (int tn = 0; tn < 2; ++tn) {
      const int n = bn + wn * 64 + tn * 32 + c4;
      const bool n_ok = n < g.N;
      const float4 rs = ep_rs[tn], bv = ep_bv[tn];
#pragma unroll
      for (int t4 = 0; t4 < TM; t4 += 4) {
#pragma unroll
        for (int tq = 0; tq < 4; ++tq)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            *reinterpret_cast<float*>(eb + tq * EPI_WAVE + (4 * q4 + e) * 144 + l15 * 4) = acc[t4 + tq][2 * tn][e];
            *reinterpret_cast<float*>(eb + tq * EPI_WAVE + (4 * q4 + e) * 144 + (16 + l15) * 4) = acc[t4 + tq][2 * tn + 1][e];
          }
        float4 v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = *reinterpret_cast<const float4*>(eb + (j >> 1) * EPI_WAVE + ((j & 1) * 8 + (lane >> 3)) * 144 + c4 * 4);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int m = bm + wm * WROWS + (t4 + (j >> 1)) * 16 + (j & 1) * 8 + (lane >> 3);
          float4 o;
          o.x = v[j].x * rs.x + bv.x; o.y = v[j].y * rs.y + bv.y; o.z = v[j].z * rs.z + bv.z; o.w = v[j].w * rs.w + bv.w;
          if (EPI == 1) { o.x = gelu_erf(o.x); o.y = gelu_erf(o.y); o.z = gelu_erf(o.z); o.w = gelu_erf(o.w); }
          float* dst = g.y + (int64_t)m * g.N + n;
          if (T256_DIAG & 4) { if (o.x == 12345.f) *reinterpret_cast<float4*>(dst) = o; }
          else if (interior) *reinterpret_cast<float4*>(dst) = o;
          else if (n_ok && m < g.M) *reinterpret_cast<float4*>(dst) = o;
        }
      }
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");             // the slices are done with: the next tile's copies may land on them
  __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory");
}

// one tile: rows [bm, bm + 32 * TM) x columns [bn, bn + 128); self-contained (first copies ... epilogue, trailing barrier)
template <int TM, int AL, int EPI>
__device__ __forceinline__ void t256_tile(const GemmF16Args& g, char* smem, const int bm, const int bn, const int w, const int lane,
                                          unsigned long long* st_sum, int& stage_ctr) {
  constexpr int WROWS = TM * 16;                            // rows per wave
  const int wm = w >> 1, wn = w & 1;
  const int l15 = lane & 15, q4 = lane >> 4;
  const int nl = (g.Rp / GK) * 2;
  const int T = nl + AL * (g.Kp / GK);
  const float lora_to_base = (AL == 2) ? g.xscale[0] : 1.f;
  const float out_scale = (AL == 2) ? g.xscale[1] : 1.f;

  // copy pieces (1 KB = 8 rows x 128 B each): A has 4 * TM pieces, wave w owns TM of them; each B limb 16, wave w owns 4
  const int prow = lane >> 3, pchunk = lane & 7;
  auto issue = [&](int t) {
    if ((T256_DIAG & 1) && t != 0) return;
    const _Float16 *A, *Bh, *Bl; int lda, ldb, k0; bool two;
    if (t < nl) {
      const int which = t & 1;
      A = which ? g.tlo : g.thi; lda = g.Rp; Bh = g.Bhi; Bl = g.Blo; ldb = g.Rp; k0 = (t >> 1) * GK; two = !which;
    } else if (AL == 1) {
      A = g.qx; lda = g.Kp; Bh = g.Whi; Bl = g.Wlo; ldb = g.Kp; k0 = (t - nl) * GK; two = true;
    } else {
      const int tb = t - nl, which = tb & 1;
      A = which ? g.xl : g.qx; lda = g.Kp; Bh = g.Whi; Bl = g.Wlo; ldb = g.Kp; k0 = (tb >> 1) * GK; two = !which;
    }
    // addresses = wave-uniform piece base (scalar registers) + one of TWO 32-bit lane offsets: a piece is 8 rows x 128 B, the
    // swizzle term (row >> 1) & 7 of its rows is (prow >> 1) for even pieces and 4 + (prow >> 1) for odd ones, and both
    // operands of a stage have the same row pitch -- 16 per-piece 64-bit lane addresses would not fit the register budget
    const int ld = lda;                                      // == ldb
    const unsigned ve = (unsigned)(prow * ld + ((pchunk ^ (prow >> 1)) * 8)) * 2u;
    const unsigned vo = (unsigned)(prow * ld + ((pchunk ^ (4 + (prow >> 1))) * 8)) * 2u;
    const char* Ab = reinterpret_cast<const char*>(A + (int64_t)bm * lda + k0) + (int64_t)(TM * w) * 16 * ld;
    const char* Bhb = reinterpret_cast<const char*>(Bh + (int64_t)bn * ldb + k0) + (int64_t)(4 * w) * 16 * ld;
    const char* Blb = reinterpret_cast<const char*>(Bl + (int64_t)bn * ldb + k0) + (int64_t)(4 * w) * 16 * ld;
#pragma unroll
    for (int i = 0; i < TM; ++i)
      glds16(Ab + (int64_t)i * 16 * ld + ((i & 1) ? vo : ve), smem + (TM * w + i) * 1024);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      glds16(Bhb + (int64_t)i * 16 * ld + ((i & 1) ? vo : ve), smem + T256_STAGE_A + (4 * w + i) * 1024);
      if (two) glds16(Blb + (int64_t)i * 16 * ld + ((i & 1) ? vo : ve), smem + T256_STAGE_A + STAGE_B + (4 * w + i) * 1024);
    }
  };

  const int sx7 = (l15 >> 1) & 7;
  const int fa_row = (wm * WROWS + l15) * 128;
  const int fb_row = T256_STAGE_A + (wn * 64 + l15) * 128;
  int koff[2];
#pragma unroll
  for (int s = 0; s < 2; ++s) koff[s] = ((4 * s + q4) ^ sx7) * 16;
  f32x4 acc[TM][4];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int jj = 0; jj < 4; ++jj)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[i][jj][e] = 0.f;

  // a stage: weight fragments of a k half first (8 reads), then one activation fragment per 8 MFMAs
  auto stage = [&](bool two, bool have_next, int nt) {
#if T256_DIAG & 8
    const unsigned long long s0 = __builtin_readcyclecounter();
#endif
#if T256_PRIO_PERIOD > 0
    if (stage_ctr % T256_PRIO_PERIOD == 0) {
      if ((stage_ctr / T256_PRIO_PERIOD) & 1) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0);
    }
    ++stage_ctr;
#endif
    __syncthreads();                                         // vmcnt(0) + barrier: the stage has landed
#if T256_DIAG & 8
    const unsigned long long s1 = __builtin_readcyclecounter();
#endif
    if (!(T256_DIAG & 2)) {
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        f16x8 bh[4], bl[4];
#pragma unroll
        for (int tn = 0; tn < 4; ++tn) {
          bh[tn] = *reinterpret_cast<const f16x8*>(smem + fb_row + tn * 2048 + koff[s]);
          if (two) bl[tn] = *reinterpret_cast<const f16x8*>(smem + fb_row + STAGE_B + tn * 2048 + koff[s]);
        }
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) {
          const f16x8 a = *reinterpret_cast<const f16x8*>(smem + fa_row + tm * 2048 + koff[s]);
#pragma unroll
          for (int tn = 0; tn < 4; ++tn) {
            acc[tm][tn] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, bh[tn], acc[tm][tn], 0, 0, 0);
            if (two) acc[tm][tn] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, bl[tn], acc[tm][tn], 0, 0, 0);
          }
        }
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");           // my fragment reads are complete (and may not sink below)
#if T256_DIAG & 8
    asm volatile("s_nop 0" :: "v"(acc[TM - 1][3][0]), "v"(acc[0][0][0]) : "memory");
    const unsigned long long s2 = __builtin_readcyclecounter();
#endif
    __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory");   // every wave has read its fragments
#if T256_DIAG & 8
    const unsigned long long s3 = __builtin_readcyclecounter();
#endif
    if (have_next) issue(nt);
#if T256_DIAG & 8
    const unsigned long long s4 = __builtin_readcyclecounter();
    st_sum[0] += s1 - s0; st_sum[1] += s2 - s1; st_sum[2] += s3 - s2; st_sum[3] += s4 - s3;
#endif
  };

  issue(0);
  for (int t = 0; t < nl; t += 2) {
    stage(true, true, t + 1);
    stage(false, t + 2 < T, t + 2);
  }
  if (nl > 0) {                                              // LoRA partial sums -> units of the base sum: * 2^-g[m]
    f32x4 riv[TM];
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) riv[tm] = *reinterpret_cast<const f32x4*>(g.rowinv + bm + wm * WROWS + tm * 16 + 4 * q4);
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float ri = riv[tm][e];
        if (AL == 2) ri *= lora_to_base;
#pragma unroll
        for (int tn = 0; tn < 4; ++tn) acc[tm][tn][e] *= ri;
      }
  }
  if (AL == 1) {
    for (int t = nl; t < T; ++t) stage(true, t + 1 < T, t + 1);
  } else {
    for (int t = nl; t < T; t += 2) {
      stage(true, true, t + 1);
      stage(false, t + 2 < T, t + 2);
    }
  }
#if T256_DIAG & 8
  const unsigned long long e0 = __builtin_readcyclecounter();
#endif
  t256_epilogue<TM, AL, EPI>(g, smem, acc, bm, bn, w, lane, out_scale);
#if T256_DIAG & 8
  st_sum[4] += __builtin_readcyclecounter() - e0;
#endif
}


// ---- ring variant (T256_RING): the 64-KB buffer is TWO 32-deep half stages ([A 256 x 64 B][B-hi 128 x 64 B][B-lo]: 32 KB each).
// Per half stage: wait for its copies + barrier (which also says that every wave has read the other slot), issue the NEXT half
// stage's copies into the other slot, then this one's fragment reads and MFMAs -- the workgroup's own matrix phase covers its own
// copy latency; one barrier per half stage.  Rows are 64 B = four 16-B chunks; chunk c of row r sits at position
// c ^ g[(r >> 2) & 3], g = {0, 3, 2, 1}: the 16 lanes of every ds_read_b128 group (rows l15, chunk q4) hit 16 distinct slots.
constexpr int T256_HALF_A = 256 * 32 * 2;                  // 16 KB
constexpr int T256_HALF_B = 128 * 32 * 2;                  // 8 KB per limb
constexpr int T256_HALF = T256_HALF_A + 2 * T256_HALF_B;   // 32 KB
__device__ __forceinline__ int ring_g(int j) { return (4 - j) & 3; }

template <int TM, int AL, int EPI>
__device__ __forceinline__ void t256_tile_ring(const GemmF16Args& g, char* smem, const int bm, const int bn, const int w, const int lane,
                                               unsigned long long* st_sum, int& stage_ctr) {
  constexpr int WROWS = TM * 16;
  const int wm = w >> 1, wn = w & 1;
  const int l15 = lane & 15, q4 = lane >> 4;
  const int nl = (g.Rp / GK) * 2;
  const int T = nl + AL * (g.Kp / GK);                      // 64-deep stages; half stages u = 2 * t + h
  const int U = 2 * T;
  const float lora_to_base = (AL == 2) ? g.xscale[0] : 1.f;
  const float out_scale = (AL == 2) ? g.xscale[1] : 1.f;

  // a copy piece = 1 KB = 16 rows x 64 B: lane L -> row L >> 2, chunk position L & 3, source chunk (L & 3) ^ g[L >> 4]
  // A: 2 * TM pieces (wave w owns TM / 2 ... of them), each B limb 8 pieces (2 per wave)
  const int prow = lane >> 2;
  const int pcol = ((lane & 3) ^ ring_g(lane >> 4)) * 8;    // elements
  auto issue = [&](int u) {
    if ((T256_DIAG & 1) && u > 1) return;
    const int t = u >> 1, h = u & 1;
    char* slot = smem + (u & 1) * T256_HALF;
    const _Float16 *A, *Bh, *Bl; int ld, k0; bool two;
    if (t < nl) {
      const int which = t & 1;
      A = which ? g.tlo : g.thi; ld = g.Rp; Bh = g.Bhi; Bl = g.Blo; k0 = (t >> 1) * GK; two = !which;
    } else if (AL == 1) {
      A = g.qx; ld = g.Kp; Bh = g.Whi; Bl = g.Wlo; k0 = (t - nl) * GK; two = true;
    } else {
      const int tb = t - nl, which = tb & 1;
      A = which ? g.xl : g.qx; ld = g.Kp; Bh = g.Whi; Bl = g.Wlo; k0 = (tb >> 1) * GK; two = !which;
    }
    k0 += 32 * h;
    const unsigned v = (unsigned)(prow * ld + pcol) * 2u;
    constexpr int AP = TM / 2;                               // A pieces per wave (16 rows each)
    const char* Ab = reinterpret_cast<const char*>(A + (int64_t)bm * ld + k0) + (int64_t)(AP * w) * 32 * ld;
    const char* Bhb = reinterpret_cast<const char*>(Bh + (int64_t)bn * ld + k0) + (int64_t)(2 * w) * 32 * ld;
    const char* Blb = reinterpret_cast<const char*>(Bl + (int64_t)bn * ld + k0) + (int64_t)(2 * w) * 32 * ld;
#pragma unroll
    for (int i = 0; i < AP; ++i) glds16(Ab + (int64_t)i * 32 * ld + v, slot + (AP * w + i) * 1024);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      glds16(Bhb + (int64_t)i * 32 * ld + v, slot + T256_HALF_A + (2 * w + i) * 1024);
      if (two) glds16(Blb + (int64_t)i * 32 * ld + v, slot + T256_HALF_A + T256_HALF_B + (2 * w + i) * 1024);
    }
  };

  const int fpos = (q4 ^ ring_g((l15 >> 2) & 3)) * 16;
  const int fa_off = (wm * WROWS + l15) * 64 + fpos;
  const int fb_off = T256_HALF_A + (wn * 64 + l15) * 64 + fpos;
  f32x4 acc[TM][4];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int jj = 0; jj < 4; ++jj)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[i][jj][e] = 0.f;

  auto half_stage = [&](int u, bool two) {
#if T256_PRIO_PERIOD > 0
    if (stage_ctr % T256_PRIO_PERIOD == 0) {
      if ((stage_ctr / T256_PRIO_PERIOD) & 1) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0);
    }
    ++stage_ctr;
#endif
#if T256_DIAG & 8
    const unsigned long long s0 = __builtin_readcyclecounter();
#endif
    __syncthreads();                                         // my copies of u have landed, my reads of the other slot are done; barrier
#if T256_DIAG & 8
    const unsigned long long s1 = __builtin_readcyclecounter();
#endif
    if (u + 1 < U) issue(u + 1);
#if T256_DIAG & 8
    const unsigned long long s2 = __builtin_readcyclecounter();
#endif
    if (!(T256_DIAG & 2)) {
      const char* slot = smem + (u & 1) * T256_HALF;
      f16x8 bh[4], bl[4];
#pragma unroll
      for (int tn = 0; tn < 4; ++tn) {
        bh[tn] = *reinterpret_cast<const f16x8*>(slot + fb_off + tn * 1024);
        if (two) bl[tn] = *reinterpret_cast<const f16x8*>(slot + fb_off + T256_HALF_B + tn * 1024);
      }
#pragma unroll
      for (int tm = 0; tm < TM; ++tm) {
        const f16x8 a = *reinterpret_cast<const f16x8*>(slot + fa_off + tm * 1024);
#pragma unroll
        for (int tn = 0; tn < 4; ++tn) {
          acc[tm][tn] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, bh[tn], acc[tm][tn], 0, 0, 0);
          if (two) acc[tm][tn] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, bl[tn], acc[tm][tn], 0, 0, 0);
        }
      }
    }
#if T256_DIAG & 8
    asm volatile("s_nop 0" :: "v"(acc[TM - 1][3][0]), "v"(acc[0][0][0]) : "memory");
    const unsigned long long s3 = __builtin_readcyclecounter();
    st_sum[0] += s1 - s0; st_sum[1] += s3 - s2; st_sum[3] += s2 - s1;
#endif
  };

  issue(0);
  for (int t = 0; t < nl; ++t) { const bool two = !(t & 1); half_stage(2 * t, two); half_stage(2 * t + 1, two); }
  if (nl > 0) {                                              // LoRA partial sums -> units of the base sum: * 2^-g[m]
    f32x4 riv[TM];
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) riv[tm] = *reinterpret_cast<const f32x4*>(g.rowinv + bm + wm * WROWS + tm * 16 + 4 * q4);
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float ri = riv[tm][e];
        if (AL == 2) ri *= lora_to_base;
#pragma unroll
        for (int tn = 0; tn < 4; ++tn) acc[tm][tn][e] *= ri;
      }
  }
  if (AL == 1) {
    for (int t = nl; t < T; ++t) { half_stage(2 * t, true); half_stage(2 * t + 1, true); }
  } else {
    for (int t = nl; t < T; ++t) { const bool two = !((t - nl) & 1); half_stage(2 * t, two); half_stage(2 * t + 1, two); }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory");   // every wave has read its last fragments: the buffer is free
#if T256_DIAG & 8
  const unsigned long long e0 = __builtin_readcyclecounter();
#endif
  t256_epilogue<TM, AL, EPI>(g, smem, acc, bm, bn, w, lane, out_scale);
#if T256_DIAG & 8
  st_sum[4] += __builtin_readcyclecounter() - e0;
#endif
}

template <int AL, int EPI>
__global__ __launch_bounds__(256, 2) void gemm_f16x2_t256_kernel(GemmF16Args g, T256Plan pl) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int G = (int)gridDim.x, b = (int)blockIdx.x;
  unsigned long long st_sum[5] = {0, 0, 0, 0, 0};
  int stage_ctr = (b >= G / 2) ? T256_PRIO_PERIOD : 0;       // the CU's second workgroup starts on the other priority
#if T256_DIAG & 8
  const unsigned long long t_kernel = __builtin_readcyclecounter();
  const unsigned long long t_real = __builtin_amdgcn_s_memrealtime();
#endif
  // whole tiles b, b + G, ...: XCD-aware order (workgroup b runs on XCD b % 8), T256_GROUP_M tile rows per L2 band
  for (int f = b; f < pl.n_full; f += G) {
    const int nwg = pl.n_full;
    const int q8 = nwg >> 3, r8 = nwg & 7, xcd = f & 7;
    const int wgid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (f >> 3);
    constexpr int GROUP_M = T256_GROUP_M;
    const int band = wgid / (GROUP_M * g.tiles_n);
    const int band_rows = min(GROUP_M, pl.full_bands - band * GROUP_M);
    const int in_band = wgid - band * GROUP_M * g.tiles_n;
    const int bm = (band * GROUP_M + in_band % band_rows) * 256;
    const int bn = (in_band / band_rows) * GN;
    if (T256_RING) t256_tile_ring<8, AL, EPI>(g, smem, bm, bn, w, lane, st_sum, stage_ctr); else t256_tile<8, AL, EPI>(g, smem, bm, bn, w, lane, st_sum, stage_ctr);
  }
  // half tiles: workgroups that got one whole tile fewer than the others take two halves first, the rest is dealt round robin
  {
    const int rem = pl.n_full % G;
    const int deficit = rem ? G - rem : 0;                   // workgroups rem .. G-1
    const int h0 = min(2 * deficit, pl.n_half);
    const int hrows = 2 * (g.tiles_m - pl.full_bands);       // 128-row blocks below the whole tiles
    auto half = [&](int h) {
      const int bm = pl.full_bands * 256 + (h % hrows) * 128;
      const int bn = (h / hrows) * GN;
      if (T256_RING) t256_tile_ring<4, AL, EPI>(g, smem, bm, bn, w, lane, st_sum, stage_ctr); else t256_tile<4, AL, EPI>(g, smem, bm, bn, w, lane, st_sum, stage_ctr);
    };
    if (rem && b >= rem) {
#pragma unroll 1
      for (int i = 0; i < 2; ++i) { const int h = 2 * (b - rem) + i; if (h < h0) half(h); }
    }
#pragma unroll 1
    for (int h = h0 + b; h < pl.n_half; h += G) half(h);
  }
#if T256_DIAG & 8
  if (g.dbg && lane == 0) {
    unsigned long long* o = g.dbg + ((int64_t)blockIdx.x * 4 + w) * 8;
    for (int i = 0; i < 5; ++i) o[i] = st_sum[i];
    o[5] = __builtin_readcyclecounter() - t_kernel;
    o[6] = t_real; o[7] = __builtin_amdgcn_s_memrealtime();
  }
#endif
}

// host: how many 256-row bands (from the bottom) to cut into half tiles so that the static assignment balances
static T256Plan t256_plan(int tiles_m /* 256-row bands */, int tiles_n, int grid) {
  T256Plan best{}; long best_span = -1;
  for (int hb = 0; hb <= tiles_m && hb <= 16; ++hb) {
    const int F = (tiles_m - hb) * tiles_n, H = 2 * hb * tiles_n;
    const int rem = F % grid, deficit = rem ? grid - rem : 0, h0 = std::min(2 * deficit, H);
    long span = 0;
    for (int b = 0; b < grid; ++b) {
      long u = 2L * ((F - b + grid - 1) / grid > 0 ? (F - b + grid - 1) / grid : 0);
      if (rem && b >= rem) { for (int i = 0; i < 2; ++i) if (2 * (b - rem) + i < h0) ++u; }
      if (H - h0 - b > 0) u += (H - h0 - b + grid - 1) / grid;
      span = std::max(span, u);
    }
    if (best_span < 0 || span < best_span) { best_span = span; best = T256Plan{F, H, tiles_m - hb, grid}; }
  }
  return best;
}

}  // namespace spq
