// The byte-level contraction (levels as bytes, three-slot ring): the former SPQ_PATH_U8X2.
// Moved out of the product library in round 3 (measured slower than the default kernels); include AFTER
// llm-qat-on-gpt2_amd/csrc/spq_f16x2.hip (tools/gemm_bench.hip does).  Not built into libspq.so, not reachable from the C ABI.
#pragma once
namespace spq {
// =================================================================================================
// The contraction, byte-level variant (input quantizer <= 8 bit): the level matrix is stored as bytes q + 128, which
// halves the A-side bytes of every stage (global -> LDS copies are what the matrix pipe waits for) and lets THREE
// 48-KB stage buffers fit in LDS, so every copy has two full stages to land (counted vmcnt, raw barriers).
// Bytes become exact fp16 in registers with two instructions per pair: v_perm_b32 builds fp16(1024 + u) (0x6400 | u),
// v_pk_add_f16 subtracts 1152.
//
// Stage kinds per tile (all use the slot layout [A 16 KB][B-hi 16 KB][B-lo 16 KB]):
//   LORA2 (32 deep)  A = thi block (fp16), B = {Bhi, Blo}      16 MFMAs per wave
//   LORA1 (32 deep)  A = tlo block (fp16), B = {Bhi}            8 MFMAs per wave
//   BASE  (64 deep)  A = level bytes,      B = {Whi, Wlo}      32 MFMAs per wave
// k assignment inside a BASE stage: lane half h owns bytes [32h, 32h+32) of its row; MFMA s (0..3) covers
// k = 32h + 8s .. +7 on both operands (any assignment is valid as long as A and B agree), so a lane reads its A operand
// for two MFMAs with ONE 16-byte LDS read.
// =================================================================================================
constexpr int U8_SLOT_A = GM * 64;                         // 16 KB (BASE: 256 x 64 B;  LORA: 256 x 32 fp16)
constexpr int U8_SLOT_B = GN * 128;                        // 16 KB per limb (BASE: 128 x 64 fp16; LORA: 128 x 32 fp16 = 8 KB used)
constexpr int U8_SLOT = U8_SLOT_A + 2 * U8_SLOT_B;         // 48 KB
constexpr int U8_EPI_WAVE = 8 * 144;                       // 8 rows x (32 floats + pad) per wave
constexpr int U8_LDS = 3 * U8_SLOT + 8 * U8_EPI_WAVE;      // 144 KB + 9 KB

typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

// 16 level bytes -> two fp16x8 MFMA operands (bytes 0..7, bytes 8..15)
__device__ __forceinline__ void unpack16(const uint4 raw, f16x8& lo8, f16x8& hi8) {
  const unsigned d[4] = {raw.x, raw.y, raw.z, raw.w};
  union { unsigned u[4]; f16x8 v; } o[2];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    union { unsigned u; f16x2 h; } a, b;
    a.u = __builtin_amdgcn_perm(0x64646464u, d[i], 0x04010400u);      // [u0, 0x64, u1, 0x64] = fp16(1024+u0), fp16(1024+u1)
    b.u = __builtin_amdgcn_perm(0x64646464u, d[i], 0x04030402u);      // u2, u3
    a.h = a.h - (f16x2)(_Float16)1152.f;                               // exact: q = u - 128
    b.h = b.h - (f16x2)(_Float16)1152.f;
    o[i >> 1].u[2 * (i & 1)] = a.u;
    o[i >> 1].u[2 * (i & 1) + 1] = b.u;
  }
  lo8 = o[0].v; hi8 = o[1].v;
}

struct GemmU8Args {
  const unsigned char* qx;                  // [Mp, Kp] bytes q + 128
  const _Float16 *thi, *tlo;                // [Mp, Rp]
  const _Float16 *Whi, *Wlo, *Bhi, *Blo;    // [Np,Kp] x2, [Np,Rp] x2
  const float *rowinv, *rowscale, *bias;
  float* y;
  int M, N, Kp, Rp;
  int tiles_m, tiles_n;
};

// V (tools/gemm_bench only; the library instantiates 0): 1 = no byte unpack, 2 = no copies after the prologue,
// 4 = extra barrier at stage start, 8 = no MFMAs / fragment reads, 16 = no epilogue stores
template <int V>
__global__ __launch_bounds__(GEMM_THREADS, 2) void gemm_u8x2_kernel(GemmU8Args g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = w >> 1, wn = w & 1;
  const int l31 = lane & 31, h = lane >> 5;

  const int nwg = g.tiles_m * g.tiles_n;
  const int nl = (g.Rp / 32) * 2;           // LoRA stages per tile (LORA2, LORA1 per 32-wide block of r)
  const int T = nl + g.Kp / GK;             // stages per tile
  const int gstride = (int)gridDim.x;

  auto tile_of = [&](int p, int& bm, int& bn) {          // same XCD-aware band order as gemm_f16x2_kernel
    const int q8 = nwg >> 3, r8 = nwg & 7, xcd = p & 7;
    const int wgid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (p >> 3);
    constexpr int GROUP_M = 8;
    const int band = wgid / (GROUP_M * g.tiles_n);
    const int band_rows = min(GROUP_M, g.tiles_m - band * GROUP_M);
    const int in_band = wgid - band * GROUP_M * g.tiles_n;
    bm = (band * GROUP_M + in_band % band_rows) * GM;
    bn = (in_band / band_rows) * GN;
  };

  // ---- copies (1 KB pieces, lane*16 linear in LDS, swizzle applied to the per-lane source address)
  // 64-B rows (A of every stage, B of LoRA stages): piece = 16 rows x 4 chunks, source chunk = pos ^ ((row>>2)&3)
  // 128-B rows (B of BASE stages):                   piece =  8 rows x 8 chunks, source chunk = pos ^ ((row>>1)&7)
  const int r64 = lane >> 2, p64 = lane & 3;
  const int r128 = lane >> 3, p128 = lane & 7;
  const int a_row0 = (2 * w) * 16 + r64, a_row1 = a_row0 + 16;          // A: 16 pieces, wave w owns 2w, 2w+1
  const int a_c0 = (p64 ^ ((a_row0 >> 2) & 3)) * 16, a_c1 = (p64 ^ ((a_row1 >> 2) & 3)) * 16;     // byte offsets in the row
  const int bb_row0 = (2 * w) * 8 + r128, bb_row1 = bb_row0 + 8;        // BASE B: 16 pieces per limb, wave w owns 2w, 2w+1
  const int bb_c0 = (p128 ^ ((bb_row0 >> 1) & 7)) * 8, bb_c1 = (p128 ^ ((bb_row1 >> 1) & 7)) * 8; // fp16 element offsets
  const int bl_row = w * 16 + r64;                                       // LoRA B: 8 pieces per limb, wave w owns piece w
  const int bl_c = (p64 ^ ((bl_row >> 2) & 3)) * 8;

  // The copies of a stage are planned once (per-lane source pointers, wave-uniform LDS destinations) and then issued
  // one instruction at a time BETWEEN groups of MFMAs of the running stage: an LDS-DMA instruction occupies its wave's
  // issue for tens of cycles, which the matrix pipe covers with the MFMAs already queued (and the SIMD's other wave).
  const char* cp_src[6];
  int cp_dst[6];
  int cp_n = 0;
  // per-lane source bases of the load cursor's tile (recomputed only when it enters a new tile)
  const char *tb_a0, *tb_a1, *tb_w0h, *tb_w0l, *tb_w1h, *tb_w1l, *tb_t0h, *tb_t1h, *tb_t0l, *tb_t1l, *tb_bh, *tb_bl;
  auto tile_bases = [&](int tbm, int tbn) {
    const unsigned char* A = g.qx + (int64_t)tbm * g.Kp;
    tb_a0 = reinterpret_cast<const char*>(A + (int64_t)a_row0 * g.Kp + a_c0);
    tb_a1 = reinterpret_cast<const char*>(A + (int64_t)a_row1 * g.Kp + a_c1);
    const int64_t b0 = (int64_t)tbn * g.Kp;
    tb_w0h = reinterpret_cast<const char*>(g.Whi + b0 + (int64_t)bb_row0 * g.Kp + bb_c0);
    tb_w0l = reinterpret_cast<const char*>(g.Wlo + b0 + (int64_t)bb_row0 * g.Kp + bb_c0);
    tb_w1h = reinterpret_cast<const char*>(g.Whi + b0 + (int64_t)bb_row1 * g.Kp + bb_c1);
    tb_w1l = reinterpret_cast<const char*>(g.Wlo + b0 + (int64_t)bb_row1 * g.Kp + bb_c1);
    tb_t0h = reinterpret_cast<const char*>(g.thi + (int64_t)(tbm + a_row0) * g.Rp) + a_c0;
    tb_t1h = reinterpret_cast<const char*>(g.thi + (int64_t)(tbm + a_row1) * g.Rp) + a_c1;
    tb_t0l = reinterpret_cast<const char*>(g.tlo + (int64_t)(tbm + a_row0) * g.Rp) + a_c0;
    tb_t1l = reinterpret_cast<const char*>(g.tlo + (int64_t)(tbm + a_row1) * g.Rp) + a_c1;
    const int64_t bo = (int64_t)(tbn + bl_row) * g.Rp + bl_c;
    tb_bh = reinterpret_cast<const char*>(g.Bhi + bo);
    tb_bl = reinterpret_cast<const char*>(g.Blo + bo);
  };
  const int d_a0 = (2 * w) * 1024, d_a1 = d_a0 + 1024;
  const int d_b0h = U8_SLOT_A + (2 * w) * 1024, d_b0l = d_b0h + U8_SLOT_B, d_b1h = d_b0h + 1024, d_b1l = d_b0l + 1024;
  const int d_lh = U8_SLOT_A + w * 1024, d_ll = d_lh + U8_SLOT_B;
  auto plan = [&](int t, int slot) {
    const int sb = slot * U8_SLOT;
    if (t < nl) {
      const int off = (t >> 1) * 64;                       // 32 fp16 per LoRA block
      cp_src[0] = ((t & 1) ? tb_t0l : tb_t0h) + off; cp_dst[0] = sb + d_a0;
      cp_src[1] = ((t & 1) ? tb_t1l : tb_t1h) + off; cp_dst[1] = sb + d_a1;
      cp_src[2] = tb_bh + off; cp_dst[2] = sb + d_lh;
      cp_src[3] = tb_bl + off; cp_dst[3] = sb + d_ll;      // LORA1 ignores it; keeps the count uniform
      cp_n = 4;
      return;
    }
    const int ka = (t - nl) * GK, kb = ka * 2;             // byte offsets: 64 level bytes / 64 fp16 per stage
    cp_src[0] = tb_a0 + ka; cp_dst[0] = sb + d_a0;
    cp_src[1] = tb_a1 + ka; cp_dst[1] = sb + d_a1;
    cp_src[2] = tb_w0h + kb; cp_dst[2] = sb + d_b0h;
    cp_src[3] = tb_w0l + kb; cp_dst[3] = sb + d_b0l;
    cp_src[4] = tb_w1h + kb; cp_dst[4] = sb + d_b1h;
    cp_src[5] = tb_w1l + kb; cp_dst[5] = sb + d_b1l;
    cp_n = 6;
  };
#define SPQ_PIECE(J) do { if ((J) < cp_n && !(V & 2)) glds16(cp_src[J], smem + cp_dst[J]); } while (0)

  // ---- fragment addressing
  const int s3 = (l31 >> 2) & 3, s7 = (l31 >> 1) & 7;     // swizzle keys of this lane's fragment rows
  const int fa = (wm * 64 + l31) * 64;                     // A row (64-B rows); + tm * 2048
  const int fb128 = U8_SLOT_A + (wn * 64 + l31) * 128;     // BASE B row; + tn * 4096 (+ U8_SLOT_B for lo)
  const int fb64 = U8_SLOT_A + (wn * 64 + l31) * 64;       // LoRA B row; + tn * 2048

  f32x16 acc[2][2];
  auto zero_acc = [&]() {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int jj = 0; jj < 2; ++jj)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][jj][e] = 0.f;
  };
  // four MFMAs on the (tm, tn) accumulators with one limb
  auto mfma4 = [&](const f16x8 (&a)[2], const f16x8 (&b)[2]) {
#pragma unroll
    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
      for (int tn = 0; tn < 2; ++tn)
        acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[tm], b[tn], acc[tm][tn], 0, 0, 0);
  };
  auto base_stage = [&](const char* sb) {
#pragma unroll
    for (int jh = 0; jh < 2; ++jh) {                       // two halves of 32 k (per lane half: 16 bytes of A)
      f16x8 a[2][2], bh[2][2], bl[2][2];                   // [s within the half][tile]
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        // read through the fp16 vector type: hipcc puts s_waitcnt vmcnt(0) (draining the copies in flight) in front of
        // an integer-typed LDS read that follows an LDS-DMA, but not in front of a half-typed one
        const f16x8 rawh = *reinterpret_cast<const f16x8*>(sb + fa + t * 2048 + (((2 * h + jh) ^ s3) << 4));
        if (V & 1) { a[0][t] = rawh; a[1][t] = rawh; } else unpack16(__builtin_bit_cast(uint4, rawh), a[0][t], a[1][t]);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const int c = ((4 * h + 2 * jh + i) ^ s7) << 4;
          bh[i][t] = *reinterpret_cast<const f16x8*>(sb + fb128 + t * 4096 + c);
          bl[i][t] = *reinterpret_cast<const f16x8*>(sb + fb128 + U8_SLOT_B + t * 4096 + c);
        }
      }
      mfma4(a[0], bh[0]); if (jh == 0) SPQ_PIECE(0); else SPQ_PIECE(4);
      mfma4(a[0], bl[0]); if (jh == 0) SPQ_PIECE(1); else SPQ_PIECE(5);
      mfma4(a[1], bh[1]); if (jh == 0) SPQ_PIECE(2);
      mfma4(a[1], bl[1]); if (jh == 0) SPQ_PIECE(3);
    }
  };
  auto lora_stage = [&](const char* sb, bool two) {
#pragma unroll
    for (int s = 0; s < 2; ++s) {                          // two k16 blocks; standard k = 16s + 8h
      f16x8 a[2], bh[2], bl[2];
      const int c = ((2 * s + h) ^ s3) << 4;
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        a[t] = *reinterpret_cast<const f16x8*>(sb + fa + t * 2048 + c);
        bh[t] = *reinterpret_cast<const f16x8*>(sb + fb64 + t * 2048 + c);
        if (two) bl[t] = *reinterpret_cast<const f16x8*>(sb + fb64 + U8_SLOT_B + t * 2048 + c);
      }
      mfma4(a, bh); if (s == 0) SPQ_PIECE(0); else SPQ_PIECE(3);
      if (s == 0) SPQ_PIECE(1);
      if (two) mfma4(a, bl);
      if (s == 0) SPQ_PIECE(2); else { SPQ_PIECE(4); SPQ_PIECE(5); }
    }
  };

  // cursors: compute (cp, ct, cbm, cbn); load (lp, lt, lbm, lbn) runs two stages ahead
  int cp = blockIdx.x, cbm, cbn;
  if (cp >= nwg) return;
  tile_of(cp, cbm, cbn);
  int lp = cp, lt = 0, lbm = cbm, lbn = cbn;
  bool lvalid = true;
  auto advance_load = [&]() {
    if (++lt == T) {
      lt = 0; lp += gstride;
      lvalid = lp < nwg;
      if (lvalid) { tile_of(lp, lbm, lbn); tile_bases(lbm, lbn); }
    }
  };
  tile_bases(lbm, lbn);
  // prologue: S_0, S_1 in flight, S_0 complete
  plan(lt, 0); advance_load();
#pragma unroll
  for (int jj = 0; jj < 6; ++jj) if (jj < cp_n) glds16(cp_src[jj], smem + cp_dst[jj]);
  int n1 = 0;
  if (lvalid) {
    plan(lt, 1); advance_load(); n1 = cp_n;
#pragma unroll
    for (int jj = 0; jj < 6; ++jj) if (jj < cp_n) glds16(cp_src[jj], smem + cp_dst[jj]);
  }
  if (n1 == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  else if (n1 == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  zero_acc();

  // The stage sequence is written as structured loops (LoRA pairs, then base stages, then the epilogue) so that the
  // 64 accumulator registers flow straight through: a single `while` over a stage cursor makes hipcc shuffle all of
  // them with v_mov at every iteration (measured: ~0.4 us per stage).
  int slot = 0;
  auto pre = [&]() -> int {                                // put S_{i+2} on the plan; returns its copy count
    cp_n = 0;
    if (lvalid) { plan(lt, slot == 0 ? 2 : slot - 1); advance_load(); }
    if (V & 4) { __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); }
    return (V & 2) ? 0 : cp_n;
  };
  auto post = [&](int n2, bool more) {                     // S_{i+1} complete: only S_{i+2}'s copies may be outstanding
    if (n2 == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else if (n2 == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (more) { __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); }
    slot = slot == 2 ? 0 : slot + 1;
  };
  auto drain_pieces = [&]() { SPQ_PIECE(0); SPQ_PIECE(1); SPQ_PIECE(2); SPQ_PIECE(3); SPQ_PIECE(4); SPQ_PIECE(5); };

#pragma unroll 1
  while (true) {
    const bool more_tiles = cp + gstride < nwg;
    for (int t = 0; t < nl; t += 2) {
      int n2 = pre();
      if (!(V & 8)) lora_stage(smem + slot * U8_SLOT, true); else drain_pieces();
      post(n2, true);
      n2 = pre();
      if (!(V & 8)) lora_stage(smem + slot * U8_SLOT, false); else drain_pieces();
      post(n2, t + 2 < T || more_tiles);
    }
    if (nl > 0) {                                          // LoRA partial sums -> units of the base sum: * 2^-g[m]
#pragma unroll
      for (int tm = 0; tm < 2; ++tm)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int m = cbm + wm * 64 + tm * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
          const float ri = g.rowinv[m];
          acc[tm][0][e] *= ri; acc[tm][1][e] *= ri;
        }
    }
    for (int t = nl; t < T; ++t) {
      const int n2 = pre();
      if (!(V & 8)) base_stage(smem + slot * U8_SLOT); else drain_pieces();
      post(n2, t + 1 < T || more_tiles);
    }
    {
      // ---- epilogue: y = acc * 2^-e[n] + bias[n]; 8 rows x 32 cols at a time through a private LDS slice, 16-B stores
      char* eb = smem + 3 * U8_SLOT + w * U8_EPI_WAVE;
      const int c4 = (lane & 7) * 4, r8 = lane >> 3;
      const bool interior = (cbm + GM <= g.M) && (cbn + GN <= g.N);
#pragma unroll
      for (int tn = 0; tn < 2; ++tn) {
        const int n = cbn + wn * 64 + tn * 32 + c4;
        const bool n_ok = n < g.N;
        float4 rs = make_float4(0.f, 0.f, 0.f, 0.f), bv = rs;
        if (n_ok) {
          rs = *reinterpret_cast<const float4*>(g.rowscale + n);
          if (g.bias) bv = *reinterpret_cast<const float4*>(g.bias + n);
        }
#pragma unroll
        for (int tm = 0; tm < 2; ++tm)
#pragma unroll
          for (int q = 0; q < 4; ++q) {                    // rows 8q .. 8q+7 of the 32x32 tile: registers 4q..4q+3
#pragma unroll
            for (int e4 = 0; e4 < 4; ++e4)
              *reinterpret_cast<float*>(eb + (e4 + 4 * h) * 144 + l31 * 4) = acc[tm][tn][4 * q + e4];
            const float4 v = *reinterpret_cast<const float4*>(eb + r8 * 144 + c4 * 4);
            const int m = cbm + wm * 64 + tm * 32 + 8 * q + r8;
            float4 o;
            o.x = v.x * rs.x + bv.x; o.y = v.y * rs.y + bv.y; o.z = v.z * rs.z + bv.z; o.w = v.w * rs.w + bv.w;
            float* dst = g.y + (int64_t)m * g.N + n;
            if (V & 16) { if (v.x == 12345.f) *reinterpret_cast<float4*>(dst) = o; }
            else if (interior) *reinterpret_cast<float4*>(dst) = o;
            else if (n_ok && m < g.M) *reinterpret_cast<float4*>(dst) = o;
          }
      }
    }
    if (!more_tiles) break;
    zero_acc();
    cp += gstride;
    tile_of(cp, cbm, cbn);
  }
}

#undef SPQ_PIECE
}  // namespace spq
