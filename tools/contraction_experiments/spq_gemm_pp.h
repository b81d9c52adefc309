// The contraction as a barrier-staggered ("ping-pong") workgroup: gemm_f16x2_pp_kernel.  Included by spq_f16x2.hip (namespace spq).
//
// Why: the 128 x 128 kernel needs 47 B/clk/CU of global->LDS copies at full matrix rate against
// the ~45 B/clk a CU takes in, keeps ONE stage buffer per workgroup and relies on three independent workgroups falling out of
// step.  This kernel
//   * walks 256 x (32 NT) tiles (NT = 6: 256 x 192, two tiles per CU at the headline shape; 26.7 B/clk of copies at full rate),
//   * splits its 8 waves into two groups of four (one wave of each group per SIMD) that run the SAME stage sequence one barrier
//     apart: while group 0 streams the MFMAs of stage s, group 1 reads its stage-s fragments and issues its share of the copies
//     of stage s+2, and vice versa -- the SIMD's matrix pipe and its memory issue are busy by construction,
//   * keeps 32-deep stages in a ring of THREE LDS slots: a copy has a whole phase (two barrier intervals) to land and is waited
//     for with a counted vmcnt one phase after it was issued (never a drain in front of a read).
// Same operands (GemmF16Args), same k -> MFMA assignment and the same accumulation order per output as gemm_f16x2_t128_kernel:
// the two kernels produce bit-identical outputs (tests/test_gpu_pp.py, tools/pp_bench.hip).
//
// LDS slot (32-deep stage): [A 256 rows x 64 B][B-hi BN x 64 B][B-lo BN x 64 B], 64-byte rows of four 16-B chunks; chunk c of
// row r sits at chunk position c ^ G((r >> 2) & 3), G = {0, 3, 2, 1}: the 16 lanes of every ds_read_b128 service group (lanes
// {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, ... of a fragment read: row = lane & 15, chunk = lane >> 4) then hit 16 distinct
// 16-byte slots of the 256-byte bank row.  Copies are LDS-DMA (16 B per lane, 1 KB = 16 rows per wave instruction) with the
// swizzle applied to the per-lane SOURCE address.
//
// Barrier intervals (B = s_barrier of all 8 waves; group 1 executes one extra barrier first, group 0 one extra at the end):
//     group 0:  LOAD_s | MFMA_s | LOAD_s+1 | MFMA_s+1 | ...
//     group 1:  MFMA_s-1 | LOAD_s | MFMA_s | LOAD_s+1 | ...
//   LOAD_s : fragment reads of stage s (slot s % 3); wait for MY copies of stage s+1 (issued in LOAD_s-1); issue my copies of
//            stage s+2 into slot (s+2) % 3 = the slot of stage s-1, whose last reader (group 1's LOAD_s-1) finished before the
//            barrier that opened this interval; lgkmcnt(0)
//   MFMA_s : the stage's MFMAs (48 per wave for NT = 6, two limbs)
// RAW: every wave has waited for its stage-(s+2) copies by the start of its LOAD_s+1, and at least one barrier lies between that
// and the first read of the slot (group 0's LOAD_s+2).
#pragma once

#ifndef PP_DIAG        // tools/pp_bench only (the library builds 0): 1 = copies of a tile's first two stages only, 2 = no MFMAs /
#define PP_DIAG 0      // fragment reads, 4 = no epilogue stores, 8 = in-kernel stamps, 16 = no epilogue at all
#endif
#ifndef PP_WAIT_LATE   // 1: wait for the previous phase's copies AFTER this phase's fragment reads have been issued (default),
#define PP_WAIT_LATE 1 // 0: before them
#endif
#ifndef PP_PRIO        // issue priority: 0 = none, 1 = the MFMA segment at priority 1, 2 = the LOAD segment at priority 1
#define PP_PRIO 1
#endif
#ifndef PP_GROUP_M
#define PP_GROUP_M 8
#endif
#if PP_DIAG & 32       // coarse stamps (a handful per wave): kernel start, loop start, per tile: end of the stage loop, end of the epilogue
#define PP_CSTAMP(i) do { if (cs_n + (i) < 12) cs[cs_n + (i)] = __builtin_readcyclecounter(); } while (0)
#else
#define PP_CSTAMP(i) do { } while (0)
#endif
#if PP_DIAG & 32
#define PP_LSTAMP(v) const unsigned long long v = __builtin_readcyclecounter()
#else
#define PP_LSTAMP(v) const unsigned long long v = 0
#endif
#if PP_DIAG & 8
#define PP_STAMP(v) const unsigned long long v = __builtin_readcyclecounter()
#else
#define PP_STAMP(v) const unsigned long long v = 0
#endif

constexpr int PP_BM = 256;

template <int NT>
struct PPCfg {
  static constexpr int BN = 32 * NT;                     // 2 waves across N, NT fragments of 16 columns each
  static constexpr int A_BYTES = PP_BM * 64;             // 16 KB
  static constexpr int B_BYTES = BN * 64;
  static constexpr int STAGE = A_BYTES + 2 * B_BYTES;
  static constexpr int NB = BN / 16;                     // 1-KB pieces per B limb
  static constexpr int PIECES = 16 + 2 * NB;
  static constexpr int P = PIECES / 8;                   // pieces per wave and stage
  static constexpr int EPI_OFF = 3 * STAGE;
  static constexpr int LDS = EPI_OFF + 8 * EPI_WAVE;
  static_assert(PIECES % 8 == 0, "pieces must divide over 8 waves");
};

__device__ __forceinline__ int pp_g(int j) { return j ^ ((j & 1) << 1); }     // {0, 3, 2, 1}

// number of stages of one tile, and whether stage t has both B limbs
template <int AL>
__device__ __forceinline__ bool pp_two(int t, int nls) {
  if (t < nls) return !(t & 2);
  if (AL == 1) return true;
  return !((t - nls) & 2);
}

// Which piece of a stage wave w copies as its i-th: group 0 (waves 0..3) the 16 A pieces and the first 4 P - 16 B-hi pieces, group 1
// the other B-hi pieces and every B-lo piece.  (A slot's B-lo block is then written by group 1 only, one interval after group 1's
// own -- later -- reads of it: the B-lo fragments may be read inside the MFMA segment.)  kind 0 = A, 1 = B-hi, 2 = B-lo.
template <int NT>
__device__ __forceinline__ int pp_piece_kind(int w, int i) {
  constexpr int P = PPCfg<NT>::P, NB = PPCfg<NT>::NB, H0 = 4 * P - 16;      // H0 B-hi pieces go to group 0
  if (w < 4) return i < 4 ? 0 : 1;
  const int q = (w - 4) * P + i;                                             // 0 .. 4P-1 over group 1
  return q < NB - H0 ? 1 : 2;
}
template <int NT>
__device__ __forceinline__ int pp_piece_index(int w, int i) {
  constexpr int P = PPCfg<NT>::P, NB = PPCfg<NT>::NB, H0 = 4 * P - 16;
  if (w < 4) return i < 4 ? 4 * w + i : (i - 4) * 4 + w;
  const int q = (w - 4) * P + i;
  return q < NB - H0 ? H0 + q : q - (NB - H0);
}

// LDS-DMA, 16 B per lane, source = wave-uniform base + per-lane BYTE offset, destination = wave-uniform LDS byte address + 16 lane.
// Inline asm: the compiler then neither drains vmcnt in front of LDS reads it cannot prove disjoint nor does 64-bit vector address
// arithmetic per piece.  Its own counted waits do not see these loads, so NO ordinary VGPR-destination load may be in flight
// across one of them: the kernel drains (s_waitcnt vmcnt(0)) after every ordinary load it issues beside the ring.
__device__ __forceinline__ void pp_glds16(const void* sbase, unsigned voff, unsigned lds_addr) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(voff), "s"(sbase), "s"(lds_addr) : "memory");
}

template <int NT, int AL, int EPI>
__global__ __launch_bounds__(512, 2) void gemm_f16x2_pp_kernel(GemmF16Args g) {
  using C = PPCfg<NT>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = w >> 1, wn = w & 1;                        // 4 (M) x 2 (N) waves, 64 x 16 NT outputs each
  const int grp = w >> 2;                                   // waves w and w + 4 share a SIMD
  const int l15 = lane & 15, q4 = lane >> 4;
  const int tiles_m = g.tiles_m;                            // 256-row tiles (Mp is a multiple of 256)
  const int tiles_n = (g.tiles_n * GN) / C::BN;             // the host takes this kernel only when BN divides Np
  const int nwg = tiles_m * tiles_n;
  const int nls = (g.Rp / 64) * 4;                          // LoRA stages: per 64-wide block (thi k0, thi k32 | tlo k0, tlo k32)
  const int T = nls + (AL == 1 ? g.Kp / 32 : (g.Kp / 64) * 4);
  const int gstride = (int)gridDim.x;
  auto tile_of = [&](int p, int& bm, int& bn) {             // XCD-aware band order: PP_GROUP_M tile rows per band
    const int q8 = nwg >> 3, r8 = nwg & 7, xcd = p & 7;
    const int wgid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (p >> 3);
    constexpr int GROUP_M = PP_GROUP_M;
    const int band = wgid / (GROUP_M * tiles_n);
    const int band_rows = min(GROUP_M, tiles_m - band * GROUP_M);
    const int in_band = wgid - band * GROUP_M * tiles_n;
    bm = (band * GROUP_M + in_band % band_rows) * PP_BM;
    bn = (in_band / band_rows) * C::BN;
  };
  const float lora_to_base = (AL == 2) ? g.xscale[0] : 1.f;
  const float out_scale = (AL == 2) ? g.xscale[1] : 1.f;
  int p = blockIdx.x;
  if (p >= nwg) return;
  int bm, bn;
  tile_of(p, bm, bn);

  // ---- copies: piece j = 8 i + w of a stage; pieces 0..15 = A, 16..16+NB-1 = B-hi, then B-lo ---------------------------
  const int prow = lane >> 2;
  const int pcol = ((lane & 3) ^ pp_g((prow >> 2) & 3)) * 8;   // source chunk (elements) that lands at chunk position lane & 3
  // ONE per-lane byte offset per row pitch (row prow of a piece + swizzled chunk); a piece's first row goes into its scalar base
  const unsigned voffK = (unsigned)(prow * g.Kp + pcol) * 2u, voffR = (unsigned)(prow * g.Rp + pcol) * 2u;
  // the operand pointers stay in SGPRs (otherwise the compiler re-reads the selected one from the kernel arguments every stage and
  // waits lgkmcnt(0) for it -- i.e. for the fragment reads just issued -- in front of the copies)
  uint64_t p_qx = (uint64_t)g.qx, p_xl = (uint64_t)g.xl, p_thi = (uint64_t)g.thi, p_tlo = (uint64_t)g.tlo;
  uint64_t p_whi = (uint64_t)g.Whi, p_wlo = (uint64_t)g.Wlo, p_bhi = (uint64_t)g.Bhi, p_blo = (uint64_t)g.Blo;
  int ldK = g.Kp, ldR = g.Rp;
  asm volatile("" : "+s"(p_qx), "+s"(p_xl), "+s"(p_thi), "+s"(p_tlo), "+s"(p_whi), "+s"(p_wlo), "+s"(p_bhi), "+s"(p_blo), "+s"(ldK), "+s"(ldR));
  // issue cursor: stage `it` of tile `ip` (coordinates ibm, ibn) goes into slot `islot`.  src[i] = source of this wave's piece i
  // for that stage (wave-uniform, first row of the piece folded in); from one stage to the next of the same operand it advances by
  // 64 bytes, and is recomputed where the operand changes (tile start, thi -> tlo, LoRA -> base, the (hi, hi | lo, lo) quads of a
  // two-limb activation).  Scalar work per stage is what the LOAD segment's length is made of: ~100 scalar instructions and 20
  // branches per stage (everything recomputed every stage) took 1300 cycles against 780 for the partner's MFMAs.
  int ip = p, it = 0, ibm = bm, ibn = bn, islot = 0;
  uint64_t src[C::P];
  unsigned pdst[C::P];                                       // byte offset of piece i inside a slot
#pragma unroll
  for (int i = 0; i < C::P; ++i) {
    const int kind = pp_piece_kind<NT>(w, i), idx = pp_piece_index<NT>(w, i);
    pdst[i] = (kind == 0 ? 0u : (kind == 1 ? (unsigned)C::A_BYTES : (unsigned)(C::A_BYTES + C::B_BYTES))) + (unsigned)idx * 1024u;
  }
  bool cur_lora = false, cur_lo = false;
  auto setup_src = [&]() {                                   // sources of stage `it` of tile (ibm, ibn)
    const bool lora = it < nls;
    const int tt = lora ? it : it - nls;
    const bool quad = lora || AL == 2;                        // four stages per 64-wide block: (hi k0, hi k32 | lo k0, lo k32)
    const int k0 = quad ? (tt >> 2) * 64 + (tt & 1) * 32 : tt * 32;
    const bool lo = quad && (tt & 2);
    // (bit masks, not selects: the compiler turns a select between pointers into a lookup table in scratch memory)
    const uint64_t mL = (uint64_t)0 - (uint64_t)lora, mO = (uint64_t)0 - (uint64_t)lo;
    const uint64_t Ap = (((p_tlo & mO) | (p_thi & ~mO)) & mL) | (((AL == 2 ? (p_xl & mO) : 0) | (p_qx & ~mO)) & ~mL);
    const uint64_t Bhp = (p_bhi & mL) | (p_whi & ~mL);
    const uint64_t Blp = (p_blo & mL) | (p_wlo & ~mL);
    const int ld = (ldR & (int)mL) | (ldK & ~(int)mL);
    const uint64_t A = Ap + (uint64_t)(((int64_t)ibm * ld + k0) * 2);
    const uint64_t Bh = Bhp + (uint64_t)(((int64_t)ibn * ld + k0) * 2);
    const uint64_t Bl = Blp + (uint64_t)(((int64_t)ibn * ld + k0) * 2);
#pragma unroll
    for (int i = 0; i < C::P; ++i) {
      const int kind = pp_piece_kind<NT>(w, i), idx = pp_piece_index<NT>(w, i);   // wave-uniform
      const uint64_t mA = (uint64_t)0 - (uint64_t)(kind == 0), mH = (uint64_t)0 - (uint64_t)(kind == 1);
      src[i] = ((A & mA) | (((Bh & mH) | (Bl & ~mH)) & ~mA)) + (uint64_t)((int64_t)(16 * idx) * ld * 2);
    }
    cur_lora = lora; cur_lo = lo;
  };
  setup_src();
  // piece i (compile-time after unrolling) of the cursor's stage.  Past the workgroup's last stage the cursor stays where it is
  // and the piece is copied once more into the ring's free slot, which nothing reads: no branch in the steady state.
  auto issue_piece = [&](int i) {
    if ((PP_DIAG & 1) && it >= 2) return;
    pp_glds16((const void*)src[i], cur_lora ? voffR : voffK, lds0 + (unsigned)islot * C::STAGE + pdst[i]);
  };
  auto issue_advance = [&]() {                               // the cursor moves to the next stage
    if (ip >= nwg) return;
    islot = islot == 2 ? 0 : islot + 1;
    ++it;
    if (it == T) {
      it = 0; ip += gstride;
      if (ip >= nwg) { it = T - 1; islot = islot == 0 ? 2 : islot - 1; return; }   // (the cursor stays on a valid stage and slot)
      tile_of(ip, ibm, ibn);
      setup_src();
    } else {
      const int tt = it < nls ? it : it - nls;
      const bool quad = it < nls || AL == 2;
      if (it == nls || (quad && !(tt & 1))) setup_src();      // a new operand or k block; otherwise + 64 bytes is right
      else {
#pragma unroll
        for (int i = 0; i < C::P; ++i) src[i] += 64;
      }
    }
  };
  auto issue_next = [&]() {
#pragma unroll
    for (int i = 0; i < C::P; ++i) issue_piece(i);
    issue_advance();
  };

  // ---- fragments ---------------------------------------------------------------------------------------------------
  const int csw = (q4 ^ pp_g((l15 >> 2) & 3)) * 16;
  const int a_rd = (wm * 64 + l15) * 64 + csw;
  const int b_rd = C::A_BYTES + (wn * 16 * NT + l15) * 64 + csw;
  f16x8 fa[4], fbh[NT], fbl[NT];
  f32x4 acc[4][NT];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int jj = 0; jj < NT; ++jj)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[i][jj][e] = 0.f;

  unsigned long long st_sum[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long cs[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; int cs_n = 0;
  (void)cs; (void)cs_n;
  PP_CSTAMP(0); cs_n = 1;
  PP_STAMP(t_kernel);
#if PP_DIAG & 8
  const unsigned long long t_real = __builtin_amdgcn_s_memrealtime();
#endif

  // ---- prologue: the first two stages --------------------------------------------------------------------------------
  issue_next();
  issue_next();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory");
  if (grp) { __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); }     // group 1 runs one barrier behind

  PP_CSTAMP(0); cs_n = 2;
  int slot = 0, ct = 0;                // slot and in-tile index of the stage being computed
  int stores_pending = 0;              // the previous tile's epilogue stores are still counted in vmcnt (interior tiles)
#pragma clang loop unroll(disable)
  while (true) {
    const bool two = pp_two<AL>(ct, nls);
    PP_STAMP(s0);
    // ---------------- LOAD segment ----------------
    const char* sb = smem + slot * C::STAGE;
    f32x4 riv[4];
    const bool rescale = (ct == nls) && nls > 0;             // LoRA partial sums -> units of the base sum: * 2^-g[m]
    if (rescale) {
#pragma unroll
      for (int tm = 0; tm < 4; ++tm) riv[tm] = *reinterpret_cast<const f32x4*>(g.rowinv + bm + wm * 64 + tm * 16 + 4 * q4);
    }
    // (the copies issued one phase ago are waited for at the END of this segment, behind this phase's own: a counted vmcnt.  The
    // compiler's waits for riv, which it counts without the LDS-DMA in flight, can only wait for more than they need: in order.)
    PP_STAMP(l0);
    if (rescale) {                                           // (before the copies: the compiler's own waits for riv must find nothing in flight)
#pragma unroll
      for (int tm = 0; tm < 4; ++tm)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float ri = riv[tm][e];
          if (AL == 2) ri *= lora_to_base;
#pragma unroll
          for (int tn = 0; tn < NT; ++tn) acc[tm][tn][e] *= ri;
        }
    }
    PP_STAMP(l1);
    // Fragment reads (LDS-bound: 4 waves x (4 + 2 NT) KB at 256 B/clk) and the copies of the stage two ahead (bound by the CU's
    // ~45 B/clk global->LDS path: ~100 cycles of issue per piece) use different units: issued alternately they overlap -- one
    // after the other they took 300 + 480 cycles of a 780-cycle MFMA interval.
    {
      constexpr int NR = 4 + NT;                             // reads of this segment: A, B-hi
      auto rd = [&](int r) {
        if (PP_DIAG & 2) return;
        if (r < 4) fa[r] = *reinterpret_cast<const f16x8*>(sb + a_rd + r * 1024);
        else fbh[r - 4] = *reinterpret_cast<const f16x8*>(sb + b_rd + (r - 4) * 1024);
      };
#pragma unroll
      for (int i = 0; i < C::P; ++i) {
        issue_piece(i);
#pragma unroll
        for (int r = i * NR / C::P; r < (i + 1) * NR / C::P; ++r) rd(r);
      }
    }
    issue_advance();
    // my copies of the NEXT stage (issued one phase ago) have landed: all but this phase's P pieces -- and, right after an
    // interior tile's epilogue, its 4 NT stores, which were issued between the two
    if (stores_pending) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(C::P + 4 * NT) : "memory");
    else asm volatile("s_waitcnt vmcnt(%0)" :: "n"(C::P) : "memory");
    stores_pending = 0;
    PP_STAMP(l2);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    PP_STAMP(s1);
    __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    PP_STAMP(s2);
    // ---------------- MFMA segment ----------------
    if (!(PP_DIAG & 2)) {
      if (PP_PRIO == 1) __builtin_amdgcn_s_setprio(1);
      if (PP_PRIO == 2) __builtin_amdgcn_s_setprio(0);
      if (two) {                                             // B-lo fragments: they land under the hi products
#pragma unroll
        for (int t = 0; t < NT; ++t) fbl[t] = *reinterpret_cast<const f16x8*>(sb + b_rd + C::B_BYTES + t * 1024);
      }
      // all hi products, then (two-limb stages) all lo products: per output the order hi, lo of the 128 x 128 kernel.  (One
      // if / else around both forms made the compiler keep two accumulator sets.)
#pragma unroll
      for (int tm = 0; tm < 4; ++tm)
#pragma unroll
        for (int tn = 0; tn < NT; ++tn)
          acc[tm][tn] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[tm], fbh[tn], acc[tm][tn], 0, 0, 0);
      if (two) {
#pragma unroll
        for (int tm = 0; tm < 4; ++tm)
#pragma unroll
          for (int tn = 0; tn < NT; ++tn)
            acc[tm][tn] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[tm], fbl[tn], acc[tm][tn], 0, 0, 0);
      }
      if (PP_PRIO == 1) __builtin_amdgcn_s_setprio(0);
      if (PP_PRIO == 2) __builtin_amdgcn_s_setprio(1);
    }
    __builtin_amdgcn_sched_barrier(0);
    if (PP_DIAG & 8) asm volatile("s_nop 0" :: "v"(acc[3][NT - 1][0]), "v"(acc[0][0][0]) : "memory");
    PP_STAMP(s3);
    // After a tile's last stage group 1 stores its outputs BEFORE this barrier and group 0 after it: both epilogues then fall
    // into the same barrier interval (group 0: epilogue + LOAD_0 of the next tile; group 1: its last MFMAs + epilogue) instead of
    // one after the other.
    const bool last = ct + 1 == T;
    if (!(last && grp)) { __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); }
    PP_STAMP(s4);
    if (PP_DIAG & 8) { st_sum[0] += s1 - s0; st_sum[1] += s2 - s1; st_sum[2] += s3 - s2; st_sum[3] += s4 - s3;
                       st_sum[4] += l0 - s0; st_sum[5] += l1 - l0; st_sum[6] += l2 - l1; st_sum[7] += s1 - l2; }
    slot = slot == 2 ? 0 : slot + 1;
    if (!last) { ++ct; continue; }

    // ---------------- epilogue: scale, bias, (GELU), whole 128-byte lines ----------------
    PP_CSTAMP(0);
    PP_STAMP(e0);
    if (!(PP_DIAG & 16)) {
      constexpr int NC = NT / 2;                              // 32-column chunks per wave
      float4 ep_rs[NC], ep_bv[NC];
#pragma unroll
      for (int tc = 0; tc < NC; ++tc) {
        const int n = bn + wn * 16 * NT + tc * 32 + (lane & 7) * 4;
        ep_rs[tc] = make_float4(0.f, 0.f, 0.f, 0.f); ep_bv[tc] = ep_rs[tc];
        if (n < g.N) {
          ep_rs[tc] = *reinterpret_cast<const float4*>(g.rowscale + n);
          if (AL == 2) { ep_rs[tc].x *= out_scale; ep_rs[tc].y *= out_scale; ep_rs[tc].z *= out_scale; ep_rs[tc].w *= out_scale; }
          if (g.bias) ep_bv[tc] = *reinterpret_cast<const float4*>(g.bias + n);
        }
      }
      if (PP_DIAG & 8) asm volatile("s_nop 0" :: "v"(ep_rs[0].x), "v"(ep_bv[NC - 1].w) : "memory");
      PP_STAMP(e1);
      if (PP_DIAG & 8) st_sum[8] += e1 - e0;
      char* eb = smem + C::EPI_OFF + w * EPI_WAVE;
      const int c4 = (lane & 7) * 4;
      const bool interior = (bm + PP_BM <= g.M) && (bn + C::BN <= g.N);
#pragma unroll
      for (int tc = 0; tc < NC; ++tc) {
        const int n = bn + wn * 16 * NT + tc * 32 + c4;
        const bool n_ok = n < g.N;
        const float4 rs = ep_rs[tc], bv = ep_bv[tc];
#pragma unroll
        for (int tm = 0; tm < 4; ++tm) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            *reinterpret_cast<float*>(eb + (4 * q4 + e) * 144 + l15 * 4) = acc[tm][2 * tc][e];
            *reinterpret_cast<float*>(eb + (4 * q4 + e) * 144 + (16 + l15) * 4) = acc[tm][2 * tc + 1][e];
          }
#pragma unroll
          for (int it2 = 0; it2 < 2; ++it2) {
            const int r16 = it2 * 8 + (lane >> 3);
            const float4 v = *reinterpret_cast<const float4*>(eb + r16 * 144 + c4 * 4);
            const int m = bm + wm * 64 + tm * 16 + r16;
            float4 o;
            o.x = v.x * rs.x + bv.x; o.y = v.y * rs.y + bv.y; o.z = v.z * rs.z + bv.z; o.w = v.w * rs.w + bv.w;
            if (EPI == 1) { o.x = gelu_erf(o.x); o.y = gelu_erf(o.y); o.z = gelu_erf(o.z); o.w = gelu_erf(o.w); }
            float* dst = g.y + (int64_t)m * g.N + n;
            if (PP_DIAG & 4) { if (o.x == 12345.f) *reinterpret_cast<float4*>(dst) = o; }
            else if (interior) *reinterpret_cast<float4*>(dst) = o;
            else if (n_ok && m < g.M) *reinterpret_cast<float4*>(dst) = o;
          }
        }
      }
      stores_pending = (interior && !(PP_DIAG & 4)) ? 1 : 0;
    }
    PP_STAMP(e2);
    if (grp) { __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); }      // group 1's deferred barrier
    PP_STAMP(e3);
    PP_CSTAMP(1); cs_n += 2;
    if (PP_DIAG & 8) { st_sum[9] += e2 - e0; st_sum[10] += e3 - e2; }
    p += gstride;
    if (p >= nwg) break;
    tile_of(p, bm, bn);
    ct = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int jj = 0; jj < NT; ++jj)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[i][jj][e] = 0.f;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // no LDS-DMA may be in flight when the workgroup's LDS is released
  if (!grp) { __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); }      // matches group 1's extra barrier
#if PP_DIAG & 32
  if (g.dbg && lane == 0) {
    unsigned long long* o = g.dbg + ((int64_t)blockIdx.x * 8 + w) * 16;
    const unsigned long long t_exit = __builtin_readcyclecounter();
    for (int i = 0; i < 12; ++i) o[i] = cs[i];
    o[12] = t_exit; o[13] = __builtin_amdgcn_s_memrealtime();
  }
#endif
#if PP_DIAG & 8
  if (g.dbg && lane == 0) {
    unsigned long long* o = g.dbg + ((int64_t)blockIdx.x * 8 + w) * 16;
    for (int i = 0; i < 4; ++i) o[i] = st_sum[i];
    o[4] = __builtin_readcyclecounter() - t_kernel;
    const unsigned long long t_end = __builtin_amdgcn_s_memrealtime();
    o[5] = t_end - t_real; o[6] = t_real; o[7] = t_end;
    for (int i = 4; i < 12; ++i) o[4 + i] = st_sum[i];
  }
#endif
}

// =====================================================================================================================
// gemm_f16x2_lc_kernel: loader / compute wave roles (12 waves: 8 compute, 4 loaders; three per SIMD, <= 168 VGPRs).
//
// What the ping-pong kernel above measured (tools/pp_bench, headline shape, in-kernel stamps): (1) a wave that issues LDS-DMA is
// held by the CU's global->LDS path for ~110 cycles per 1-KB piece, so a LOAD segment with 5 pieces + 10..16 fragment reads takes
// 850..950 cycles against 770 for the partner's 48 MFMAs -- the loop runs at 71 % of the matrix rate; (2) every CU reaches its
// epilogue at the same time: 50 MB of stores per tile round, ~5 us at the chip's write rate, and since vmcnt counts loads and
// stores in issue order a wave cannot confirm its next copies before its own 24 stores have completed -- 10..20 k cycles per
// tile with the matrix pipe idle.  Hence the roles:
//   * waves 8..11 (one per SIMD) only copy: every half stage they issue P pieces of the ring's next free half and wait, with a
//     counted vmcnt that never sees a store, for the half issued one step earlier.  The copy path runs all the time.
//   * waves 0..7 compute (4 x 2, 64 x 16 NT outputs each) and never issue a vector-memory LOAD inside the stage loop (the LoRA row
//     scales come through LDS, staged by a loader): epilogue stores are fire-and-forget and drain under the next tile's MFMAs.
//     A wave reads its A fragments once per stage and walks the N fragments with a two-deep window of (B-hi, B-lo) pairs, the
//     reads of pair tn+1 in flight under the 8 MFMAs of pair tn: 96 accumulator + 16 + 16 fragment registers.
//   * the two compute groups (waves 0..3, 4..7: one wave of each per SIMD) run half a stage apart (two barriers per stage, all 12
//     waves), so one group's stage-start fragment reads sit under the other group's MFMAs.
// Barriers b_0 .. b_2S (S = stages of this workgroup).  Group 0 computes stage s in [b_2s, b_2s+2), group 1 in [b_2s+1, b_2s+3).
// The loaders cut a stage's pieces into halves H_2s (A, first B-hi pieces) and H_2s+1 (other B-hi, B-lo); after b_k they issue
// H_k+3 -- into the slot of stage (k+3)/2 - 3, which group 1 left at b_k at the latest -- and wait until everything up to H_k+2 has
// landed before they arrive at b_k+1: stage s is complete in LDS at b_2s.  Same arithmetic and order per output as the other
// f16 kernels: bit-identical results.
// =====================================================================================================================
#ifndef LC_WIN
#define LC_WIN 3
#endif
template <int NT> struct LCCfg {
  using C = PPCfg<NT>;
  static constexpr int RIV_OFF = C::EPI_OFF + 8 * EPI_WAVE;  // two 1-KB blocks (tile parity): rowinv of the tile's 256 rows
  static constexpr int FLAG_OFF = RIV_OFF + 2048;            // per ring slot: FULL counter (+1 per loader wave and fill) at +16 slot,
  static constexpr int LDS = FLAG_OFF + 128;                 // FREE counter (+1 per compute wave and stage read) at +64 + 16 slot
};

template <int NT, int AL, int EPI>
__global__ __launch_bounds__(768, 3) void gemm_f16x2_lc_kernel(GemmF16Args g) {
  using C = PPCfg<NT>;
  using L = LCCfg<NT>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tiles_m = g.tiles_m;
  const int tiles_n = (g.tiles_n * GN) / C::BN;             // the host takes this kernel only when BN divides Np
  const int nwg = tiles_m * tiles_n;
  const int nls = (g.Rp / 64) * 4;
  const int T = nls + (AL == 1 ? g.Kp / 32 : (g.Kp / 64) * 4);
  const int gstride = (int)gridDim.x;
  auto tile_of = [&](int p, int& bm, int& bn) {
    const int q8 = nwg >> 3, r8 = nwg & 7, xcd = p & 7;
    const int wgid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (p >> 3);
    constexpr int GROUP_M = PP_GROUP_M;
    const int band = wgid / (GROUP_M * tiles_n);
    const int band_rows = min(GROUP_M, tiles_m - band * GROUP_M);
    const int in_band = wgid - band * GROUP_M * tiles_n;
    bm = (band * GROUP_M + in_band % band_rows) * PP_BM;
    bn = (in_band / band_rows) * C::BN;
  };
  int p = blockIdx.x;
  if (p >= nwg) return;
  const int my_tiles = (nwg - p + gstride - 1) / gstride;
  const int S = my_tiles * T;                                // stages of this workgroup; barriers b_0 .. b_2S
  int bm, bn;
  tile_of(p, bm, bn);

  // Hand-off through monotonic counters in LDS instead of workgroup barriers (mfma_stream_probe.hip: a barrier per stage or half stage
  // costs two MFMA-streaming waves of a SIMD a third of their rate -- 1747 against 1311 cycles per stage -- because it puts them in
  // step): FULL[slot] += 1 by each loader wave once its pieces of a fill have landed (behind its counted vmcnt), FREE[slot] += 1 by
  // each compute wave behind its last fragment read of a stage (the LDS executes a wave's operations in order).
  if (tid < 32) *reinterpret_cast<unsigned*>(smem + L::FLAG_OFF + tid * 4) = 0u;
  __syncthreads();
  auto lds_poll = [&](unsigned addr, int target) {           // spin until the counter has reached target (wave-uniform)
    unsigned v;
    do {
      asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
      v = __builtin_amdgcn_readfirstlane(v);
      if ((int)(v - (unsigned)target) < 0) __builtin_amdgcn_s_sleep(1);
    } while ((int)(v - (unsigned)target) < 0);
  };
  auto lds_add1 = [&](unsigned addr) {                       // one lane adds 1, no return value, no wait
    if (lane == 0) asm volatile("ds_add_u32 %0, %1" :: "v"(addr), "v"(1u) : "memory");
  };
  const unsigned flags = lds0 + L::FLAG_OFF;
  if (w >= 8) {
    // ================================================ loader ================================================
    // Hot path per stage: 2 P x (LDS address -> m0, one LDS-DMA, source += 64 B).  Everything else -- which operand, which k block,
    // which tile -- changes only at SEGMENT boundaries (a tile's thi stages, its tlo stages, its base stages: three per tile at the
    // headline shape), where next_segment() recomputes the 2 P sources from scratch.  (A generic per-stage cursor cost ~850
    // scalar cycles per stage, and a wave's scalar stream advances slowly beside two MFMA-streaming waves on its SIMD.)
    const int lw = w - 8;
#ifdef LC_FAKE128      // timing probe only (wrong data): a piece reads 8 rows x 128 B instead of 16 rows x 64 B
    const int prow = lane >> 3;
    const int pcol = (lane & 7) * 8;
#else
    const int prow = lane >> 2;
    const int pcol = ((lane & 3) ^ pp_g((prow >> 2) & 3)) * 8;
#endif
    const unsigned voffK = (unsigned)(prow * g.Kp + pcol) * 2u, voffR = (unsigned)(prow * g.Rp + pcol) * 2u;
    constexpr int NP = 2 * C::P;
    uint64_t src[NP];
    unsigned dstc[NP];
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const int h = i / C::P, ii = i % C::P;
      const int kd = pp_piece_kind<NT>(4 * h + lw, ii), r0 = 16 * pp_piece_index<NT>(4 * h + lw, ii);
      dstc[i] = (kd == 0 ? 0u : (kd == 1 ? (unsigned)C::A_BYTES : (unsigned)(C::A_BYTES + C::B_BYTES))) + (unsigned)r0 * 64u;
    }
    uint64_t p_riv = (uint64_t)g.rowinv;
    asm volatile("" : "+s"(p_riv));
    int ip = p, ibm = bm, ibn = bn, itile = 0;
    int seg = 0, seg_left = 0;                               // segment index inside the tile, stages left in it
    unsigned voff = voffK;
    // segments of a tile: LoRA blocks (hi: 2 stages, lo: 2 stages) x Rp/64, then AL == 1: one base segment of Kp/32 stages,
    // AL == 2: per 64-wide block (hi: 2, lo: 2)
    const int n_lora_seg = (g.Rp / 64) * 2;
    const int n_seg = n_lora_seg + (AL == 1 ? 1 : (g.Kp / 64) * 2);
    auto next_segment = [&]() {
      const bool lora = seg < n_lora_seg;
      const int sb2 = lora ? seg : seg - n_lora_seg;
      const bool quad = lora || AL == 2;
      const bool lo = quad && (sb2 & 1);
      const int k0 = quad ? (sb2 >> 1) * 64 : 0;
      seg_left = quad ? 2 : g.Kp / 32;
      // (bit masks, not selects: the compiler turns a select between pointers into a lookup table in scratch memory)
      const uint64_t mL = (uint64_t)0 - (uint64_t)lora, mO = (uint64_t)0 - (uint64_t)lo;
      const uint64_t Ap = ((((uint64_t)g.tlo & mO) | ((uint64_t)g.thi & ~mO)) & mL) | ((((AL == 2 ? (uint64_t)g.xl : 0) & mO) | ((uint64_t)g.qx & ~mO)) & ~mL);
      const uint64_t Bhp = ((uint64_t)g.Bhi & mL) | ((uint64_t)g.Whi & ~mL);
      const uint64_t Blp = ((uint64_t)g.Blo & mL) | ((uint64_t)g.Wlo & ~mL);
      const int ld = (g.Rp & (int)mL) | (g.Kp & ~(int)mL);
      voff = lora ? voffR : voffK;
#pragma unroll
      for (int i = 0; i < NP; ++i) {
        const int h = i / C::P, ii = i % C::P;
        const int kd = pp_piece_kind<NT>(4 * h + lw, ii), r0 = 16 * pp_piece_index<NT>(4 * h + lw, ii);
        const uint64_t mA = (uint64_t)0 - (uint64_t)(kd == 0), mH = (uint64_t)0 - (uint64_t)(kd == 1);
        const uint64_t base = (Ap & mA) | (((Bhp & mH) | (Blp & ~mH)) & ~mA);
        const int r = ((ibm & (int)mA) | (ibn & ~(int)mA)) + r0;
        src[i] = base + (uint64_t)(((int64_t)r * ld + k0) * 2);
      }
    };
    next_segment();
    unsigned slot_base = lds0;
    int stages_issued = 0;
    // a stage is issued in two halves (pieces [0, P) and [P, 2P)) so that the landing of the PREVIOUS stage can be confirmed and
    // published between them: its copies then have this stage's wait for a free slot plus half an issue to land, and FULL is
    // published as early as a counted vmcnt allows
    auto issue_first_half = [&]() {
      if (seg == 0 && seg_left == (n_lora_seg ? 2 : (AL == 1 ? g.Kp / 32 : 2)) && lw == 0 && n_lora_seg > 0)   // first stage of a tile
        pp_glds16((const void*)(p_riv + (uint64_t)ibm * 4u), (unsigned)lane * 16u, lds0 + L::RIV_OFF + (unsigned)(itile & 1) * 1024u);
      if (!((PP_DIAG & 1) && stages_issued >= 2)) {
#pragma unroll
        for (int i = 0; i < C::P; ++i) pp_glds16((const void*)src[i], voff, slot_base + dstc[i]);
      }
    };
    auto issue_second_half = [&]() {
      if (!((PP_DIAG & 1) && stages_issued >= 2)) {
#pragma unroll
        for (int i = C::P; i < NP; ++i) pp_glds16((const void*)src[i], voff, slot_base + dstc[i]);
      }
#pragma unroll
      for (int i = 0; i < NP; ++i) src[i] += 64;
      slot_base = slot_base == lds0 + 2u * C::STAGE ? lds0 : slot_base + C::STAGE;
      ++stages_issued;
      if (--seg_left == 0) {
        if (++seg == n_seg) {
          seg = 0; ip += gstride; ++itile;
          if (ip < nwg) tile_of(ip, ibm, ibn);
        }
        if (ip < nwg) next_segment();
      }
    };
#ifdef LC_LPRIO
    __builtin_amdgcn_s_setprio(LC_LPRIO);
#endif
    unsigned long long l_bar = 0, l_iss = 0, l_wait = 0;
    (void)l_bar; (void)l_iss; (void)l_wait;
    int slot_i = 0, use_i = 0;                               // slot of the stage to issue, how often that slot has been filled
#pragma clang loop unroll(disable)
    for (int k = 0; k < S; ++k) {
      PP_LSTAMP(t0);
      if (use_i > 0 && !(PP_DIAG & (256 | 1024))) lds_poll(flags + 64u + 16u * (unsigned)slot_i, 8 * use_i);      // every compute wave has left the slot's last stage
      PP_LSTAMP(t1);
      issue_first_half();
      PP_LSTAMP(t2);
      if (k > 0) {                                           // stage k-1 has landed: all but the P pieces just issued
        if (PP_DIAG & 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" :: "n"(C::P) : "memory");
        lds_add1(flags + 16u * (unsigned)(slot_i == 0 ? 2 : slot_i - 1));
      }
      PP_LSTAMP(t3);
      issue_second_half();
      PP_LSTAMP(t4);
      if (PP_DIAG & 32) { l_bar += t1 - t0; l_iss += (t2 - t1) + (t4 - t3); l_wait += t3 - t2; }
      if (++slot_i == 3) { slot_i = 0; ++use_i; }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    lds_add1(flags + 16u * (unsigned)(slot_i == 0 ? 2 : slot_i - 1));
#if PP_DIAG & 32
    if (g.dbg && lane == 0) {
      unsigned long long* o = g.dbg + ((int64_t)gridDim.x * 8 + (int64_t)blockIdx.x * 4 + lw) * 16;
      o[0] = l_bar; o[1] = l_iss; o[2] = l_wait;
    }
#endif
    return;
  }

  // ================================================== compute ==================================================
  const int wm = w >> 1, wn = w & 1;                        // 4 (M) x 2 (N) waves, 64 x 16 NT outputs each
  const int l15 = lane & 15, q4 = lane >> 4;
  const float lora_to_base = (AL == 2) ? g.xscale[0] : 1.f;
  const float out_scale = (AL == 2) ? g.xscale[1] : 1.f;
  const int csw = (q4 ^ pp_g((l15 >> 2) & 3)) * 16;
  const int a_rd = (wm * 64 + l15) * 64 + csw;
  const int b_rd = C::A_BYTES + (wn * 16 * NT + l15) * 64 + csw;
  f32x4 acc[4][NT];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int jj = 0; jj < NT; ++jj)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[i][jj][e] = 0.f;

  unsigned long long cs[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; int cs_n = 0;
  (void)cs; (void)cs_n;
  PP_CSTAMP(0); cs_n = 1;

  auto epilogue = [&](int ebm, int ebn) {
    if (PP_DIAG & 16) return;
    constexpr int NC = NT / 2;
    float4 ep_rs[NC], ep_bv[NC];
#pragma unroll
    for (int tc = 0; tc < NC; ++tc) {
      const int n = ebn + wn * 16 * NT + tc * 32 + (lane & 7) * 4;
      ep_rs[tc] = make_float4(0.f, 0.f, 0.f, 0.f); ep_bv[tc] = ep_rs[tc];
      if (n < g.N) {
        ep_rs[tc] = *reinterpret_cast<const float4*>(g.rowscale + n);
        if (AL == 2) { ep_rs[tc].x *= out_scale; ep_rs[tc].y *= out_scale; ep_rs[tc].z *= out_scale; ep_rs[tc].w *= out_scale; }
        if (g.bias) ep_bv[tc] = *reinterpret_cast<const float4*>(g.bias + n);
      }
    }
    char* eb = smem + C::EPI_OFF + w * EPI_WAVE;
    const int c4 = (lane & 7) * 4;
    const bool interior = (ebm + PP_BM <= g.M) && (ebn + C::BN <= g.N);
#pragma unroll
    for (int tc = 0; tc < NC; ++tc) {
      const int n = ebn + wn * 16 * NT + tc * 32 + c4;
      const bool n_ok = n < g.N;
      const float4 rs = ep_rs[tc], bv = ep_bv[tc];
#pragma unroll
      for (int tm = 0; tm < 4; ++tm) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          *reinterpret_cast<float*>(eb + (4 * q4 + e) * 144 + l15 * 4) = acc[tm][2 * tc][e];
          *reinterpret_cast<float*>(eb + (4 * q4 + e) * 144 + (16 + l15) * 4) = acc[tm][2 * tc + 1][e];
        }
#pragma unroll
        for (int it2 = 0; it2 < 2; ++it2) {
          const int r16 = it2 * 8 + (lane >> 3);
          const float4 v = *reinterpret_cast<const float4*>(eb + r16 * 144 + c4 * 4);
          const int m = ebm + wm * 64 + tm * 16 + r16;
          float4 o;
          o.x = v.x * rs.x + bv.x; o.y = v.y * rs.y + bv.y; o.z = v.z * rs.z + bv.z; o.w = v.w * rs.w + bv.w;
          if (EPI == 1) { o.x = gelu_erf(o.x); o.y = gelu_erf(o.y); o.z = gelu_erf(o.z); o.w = gelu_erf(o.w); }
          float* dst = g.y + (int64_t)m * g.N + n;
          if (PP_DIAG & 4) { if (o.x == 12345.f) *reinterpret_cast<float4*>(dst) = o; }
          else if (interior) *reinterpret_cast<float4*>(dst) = o;
          else if (n_ok && m < g.M) *reinterpret_cast<float4*>(dst) = o;
        }
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int jj = 0; jj < NT; ++jj)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[i][jj][e] = 0.f;
  };

  int slot = 0, use = 0, ct = 0, tile_i = 0;
  unsigned full_ahead = 0;                                   // FULL counter of the next stage's slot as read during this stage
#ifdef LC_G1PRIO
  if (w >= 4) __builtin_amdgcn_s_setprio(LC_G1PRIO);         // the younger wave of each SIMD otherwise gets what the older one leaves
#endif
  PP_CSTAMP(0); cs_n = 2;
#pragma clang loop unroll(disable)
  while (true) {
    const bool two = pp_two<AL>(ct, nls);
    // every loader wave's pieces of this stage have landed.  The counter was read ahead, in the middle of the previous stage (the
    // read's latency and its s_waitcnt in front of the stage's first fragment reads cost ~12 % of the loop); if the fill had not
    // been published by then, poll.
    if (!(PP_DIAG & (256 | 512))) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if ((int)((unsigned)__builtin_amdgcn_readfirstlane(full_ahead) - (unsigned)(4 * (use + 1))) < 0) lds_poll(flags + 16u * (unsigned)slot, 4 * (use + 1));
    }
    if (ct == nls && nls > 0) {                              // LoRA partial sums -> units of the base sum: * 2^-g[m]
      const char* rb = smem + L::RIV_OFF + (tile_i & 1) * 1024 + (wm * 64 + 4 * q4) * 4;
#pragma unroll
      for (int tm = 0; tm < 4; ++tm) {
        const f32x4 riv = *reinterpret_cast<const f32x4*>(rb + tm * 64);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float ri = riv[e];
          if (AL == 2) ri *= lora_to_base;
#pragma unroll
          for (int tn = 0; tn < NT; ++tn) acc[tm][tn][e] *= ri;
        }
      }
    }
    // A fragments once per stage; the N fragments through a window of LC_WIN (B-hi, B-lo) pairs: the reads of pair tn + LC_WIN - 1
    // travel under the MFMAs of pairs tn .. tn + LC_WIN - 2.  The reads are inline asm with counted lgkmcnt waits: left to the
    // compiler, (a) with the limb count a run-time branch it counts one read per pair and waits for the pair issued ONE iteration
    // ago instead of two, (b) with both limbs read unconditionally it sinks the B-lo reads into the conditional block that uses
    // them, right in front of their MFMAs, (c) two instantiations of the body make it keep two accumulator sets.  Both limbs are
    // read in every stage (a one-limb stage reads a stale B-lo block and ignores it); only the lo MFMAs are conditional.
    if (!(PP_DIAG & 2)) {
      f16x8 fa[4], bh[LC_WIN], bl[LC_WIN];
      const unsigned aa = lds0 + (unsigned)slot * C::STAGE + (unsigned)a_rd, ba = lds0 + (unsigned)slot * C::STAGE + (unsigned)b_rd;
#define LC_DSREAD(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off) : "memory")
#pragma unroll
      for (int t = 0; t < 4; ++t) LC_DSREAD(fa[t], aa, t * 1024);
#pragma unroll
      for (int j = 0; j < LC_WIN - 1; ++j) { LC_DSREAD(bh[j], ba, j * 1024); LC_DSREAD(bl[j], ba, C::B_BYTES + j * 1024); }
#pragma unroll
      for (int tn = 0; tn < NT; ++tn) {
        constexpr int W1 = LC_WIN - 1;
        if (tn + W1 < NT) { LC_DSREAD(bh[(tn + W1) % LC_WIN], ba, (tn + W1) * 1024); LC_DSREAD(bl[(tn + W1) % LC_WIN], ba, C::B_BYTES + (tn + W1) * 1024); }
        if (tn + W1 == NT - 1 && !(PP_DIAG & (256 | 1024))) lds_add1(flags + 64u + 16u * (unsigned)slot);        // behind this wave's last read of the stage
        if (tn + W1 == NT - 1 && !(PP_DIAG & (256 | 512)))      // ... and the NEXT stage's FULL counter, read ahead
          asm volatile("ds_read_b32 %0, %1" : "=v"(full_ahead) : "v"(flags + 16u * (unsigned)(slot == 2 ? 0 : slot + 1)) : "memory");
        const int ahead = (tn + W1 < NT ? tn + W1 : NT - 1) - tn;      // pairs issued behind pair tn
        // (the ds_add behind the last pair counts in lgkmcnt too: one more operation in flight from there on)
        const int extra = (tn + W1 >= NT - 1 ? 1 : 0) * ((PP_DIAG & (256 | 1024) ? 0 : 1) + (PP_DIAG & (256 | 512) ? 0 : 1));
        if (2 * ahead + extra == 6) asm volatile("s_waitcnt lgkmcnt(6)" ::: "memory");
        else if (2 * ahead + extra == 5) asm volatile("s_waitcnt lgkmcnt(5)" ::: "memory");
        else if (2 * ahead + extra == 4) asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
        else if (2 * ahead + extra == 3) asm volatile("s_waitcnt lgkmcnt(3)" ::: "memory");
        else if (2 * ahead + extra == 2) asm volatile("s_waitcnt lgkmcnt(2)" ::: "memory");
        else if (2 * ahead + extra == 1) asm volatile("s_waitcnt lgkmcnt(1)" ::: "memory");
        else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int tm = 0; tm < 4; ++tm) acc[tm][tn] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[tm], bh[tn % LC_WIN], acc[tm][tn], 0, 0, 0);
        if (two) {
#pragma unroll
          for (int tm = 0; tm < 4; ++tm) acc[tm][tn] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[tm], bl[tn % LC_WIN], acc[tm][tn], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
#undef LC_DSREAD
    } else { lds_add1(flags + 64u + 16u * (unsigned)slot); full_ahead = 0; }
    if (++slot == 3) { slot = 0; ++use; }
    if (++ct < T) continue;
    // ---- the tile is complete: its stores are issued and left behind (no vector-memory load follows them in this wave) ----
    PP_CSTAMP(0);
    epilogue(bm, bn);
    PP_CSTAMP(1); cs_n += 2;
    p += gstride; ++tile_i; ct = 0;
    if (p >= nwg) break;
    tile_of(p, bm, bn);
  }
#if PP_DIAG & 32
  if (g.dbg && lane == 0) {
    unsigned long long* o = g.dbg + ((int64_t)blockIdx.x * 8 + w) * 16;
    const unsigned long long t_exit = __builtin_readcyclecounter();
    for (int i = 0; i < 12; ++i) o[i] = cs[i];
    o[12] = t_exit; o[13] = __builtin_amdgcn_s_memrealtime();
  }
#endif
}
