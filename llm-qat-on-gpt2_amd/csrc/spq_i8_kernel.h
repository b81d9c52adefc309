// SPQ_PATH_I8: the dense contraction on the int8 matrix cores (v_mfma_i32_32x32x32_i8, 2x the f16 rate), exact integer sums.
//
// For a symmetric minmax input quantizer of <= 8 bits the activation operand is the integer level q[m,k] itself (int8).
//   NL = 3 (any weight quantizer, per-channel or per-tensor input scale): the weight operand W'[n,k] = FQ(W)[n,k] * sx[k] is a
//       real number; scaled per row by 2^e[n] it is rounded to a 23-bit integer I (|I| < 2^22) and cut into three balanced
//       base-256 digits I = l2 * 65536 + l1 * 256 + l0, each an int8.  Three i32 accumulators per output,
//           y[m,n] = 2^-e[n] * ( 65536 * sum_k q l2 + 256 * sum_k q l1 + sum_k q l0 ),
//       every partial sum exact; the merge is three exact int -> float conversions and two fp32 roundings.
//   NL = 1 (minmax weights of <= 8 bits AND a per-tensor input scale: sx leaves the sum): the weight operand is the weight's own
//       integer level, y[m,n] = (sw[n] * sx) * sum_k q[m,k] * wq[n,k]: ONE int8 product per algorithmic product.
// The LoRA branch (raw fp32 x, lora.py:149) keeps its f16 two-limb form (thi/tlo from the activation pass, Bhi/Blo with a row
// exponent of their own) and runs after the base sum on v_mfma_f32_32x32x16_f16 into a second accumulator set:
//           y = base * rowscale[n] + ( u * 2^-g[m] ) * bscale[n] + bias[n].
//
// Kernel: 256 x 128 tile, 8 waves (4 x 2 of 64 x 64), persistent, XCD-aware tile order, a ring of three 40-KB LDS slots
// (A 256 x 64 B, three 128 x 64 B limb planes) filled by direct global->LDS copies two stages ahead (counted vmcnt, raw
// barriers), copies issued between MFMA groups, epilogue through per-wave LDS slices for 16-B stores.
#pragma once
#include "spq_common.h"

namespace spq {

typedef int i32x4v __attribute__((ext_vector_type(4)));
typedef int i32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef float fl32x16 __attribute__((ext_vector_type(16)));

constexpr int I8_GM = 256, I8_GN = 128, I8_GK = 64;
constexpr int I8_SLOT_A = I8_GM * 64;                       // 16 KB
constexpr int I8_SLOT_B = I8_GN * 64;                       // 8 KB per limb plane (LoRA: Bhi, Blo of a 32-wide block)
constexpr int I8_SLOT = I8_SLOT_A + 3 * I8_SLOT_B;          // 40 KB
constexpr int I8_EPI_WAVE = 8 * 144;                        // 8 rows x (32 floats + pad) per wave
constexpr int I8_LDS = 3 * I8_SLOT + 8 * I8_EPI_WAVE;       // 120 KB + 9 KB
constexpr int I8_THREADS = 512;

struct GemmI8Args {
  const signed char* qx;                    // [Mp, Kp] levels
  const signed char* W;                     // NL planes [Np, Kp], plane p at W + p * plane_stride
  int64_t plane_stride;
  const _Float16 *thi, *tlo;                // [Mp, Rp]
  const _Float16 *Bhi, *Blo;                // [Np, Rp]
  const float *rowinv, *rowscale, *bscale, *bias;   // [Mp] 2^-g[m]; [Np]; [Np] 2^-eb[n]; [N] nullable
  float* y;
  int M, N, Kp, Rp;                         // Rp = 0: no LoRA-up
  int tiles_m, tiles_n;
  int epilogue;                             // SPQ_EPILOGUE_*
};

__device__ __forceinline__ void i8_glds16(const void* gsrc, void* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

__device__ __forceinline__ float i8_gelu_erf(float x) { return (x * 0.5f) * (1.0f + erff(x * 0.70710678118654752440f)); }

// V (tools/i8_gemm_bench only; the library instantiates 0): 2 = no copies after the prologue, 8 = no MFMAs / fragment reads,
// 16 = no epilogue stores
template <int NL, int V, int EPI>
__global__ __launch_bounds__(I8_THREADS, 2) void gemm_i8_kernel(GemmI8Args g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = w >> 1, wn = w & 1;
  const int l31 = lane & 31, h = lane >> 5;

  const int nwg = g.tiles_m * g.tiles_n;
  const int nb = g.Kp / I8_GK;              // base stages per tile
  const int nl = (g.Rp / 32) * 2;           // LoRA stages per tile (LORA2, LORA1 per 32-wide block of r), after the base
  const int T = nb + nl;
  const int gstride = (int)gridDim.x;

  auto tile_of = [&](int p, int& bm, int& bn) {            // XCD-aware band order: blocks p, p+8, ... share an L2
    const int q8 = nwg >> 3, r8 = nwg & 7, xcd = p & 7;
    const int wgid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (p >> 3);
    constexpr int GROUP_M = 8;
    const int band = wgid / (GROUP_M * g.tiles_n);
    const int band_rows = min(GROUP_M, g.tiles_m - band * GROUP_M);
    const int in_band = wgid - band * GROUP_M * g.tiles_n;
    bm = (band * GROUP_M + in_band % band_rows) * I8_GM;
    bn = (in_band / band_rows) * I8_GN;
  };

  // ---- copies: 1-KB pieces = 16 rows x 64 B, lane * 16 linear in LDS, swizzle applied to the per-lane SOURCE address:
  // source 16-B chunk = pos ^ ((row >> 2) & 3).  A: 16 pieces (wave w: 2w, 2w+1); each B plane: 8 pieces (wave w: piece w).
  const int r64 = lane >> 2, p64 = lane & 3;
  const int a_row0 = (2 * w) * 16 + r64, a_row1 = a_row0 + 16;
  const int a_c0 = (p64 ^ ((a_row0 >> 2) & 3)) * 16, a_c1 = (p64 ^ ((a_row1 >> 2) & 3)) * 16;      // byte offsets in the row
  const int b_row = w * 16 + r64;
  const int b_c = (p64 ^ ((b_row >> 2) & 3)) * 16;

  // Copy sources are (wave-uniform base) + (per-lane 32-bit byte offset): four lane offsets in all -- the 64-bit per-lane
  // pointers of every operand cost ~30 VGPRs, which the three accumulator sets of NL = 3 do not leave.
  const int a_off = a_row0 * g.Kp + a_c0;                  // row a_row1 = a_row0 + 16 has the same swizzle key: + 16 * Kp (uniform)
  const int w_off = b_row * g.Kp + b_c;
  const int t_off = a_row0 * g.Rp * 2 + a_c0;
  const int l_off = b_row * g.Rp * 2 + b_c;
  // A stage's copies are CPN pieces whatever its kind (a kind with fewer real pieces repeats its last one: same bytes to the
  // same place), so the in-flight count the waits rely on is a constant and the issue sites carry one wave-uniform test.
  constexpr int CPN = NL == 3 ? 5 : 4;
  const char *cb0 = nullptr, *cb1 = nullptr, *cb2 = nullptr, *cb3 = nullptr, *cb4 = nullptr;     // wave-uniform bases
  int cd0 = 0, cd1 = 0, cd2 = 0, cd3 = 0, cd4 = 0;                                                // wave-uniform LDS offsets
  const int dt_off = t_off - a_off, dl_off = l_off - w_off;
  int offA = a_off, offB = w_off;                          // per-lane byte offsets of the planned stage's kind
  bool cp_on = false;
  int lp_bm = 0, lp_bn = 0;                                // tile of the load cursor
  auto tile_bases = [&](int tbm, int tbn) { lp_bm = tbm; lp_bn = tbn; };
  const int d_a0 = (2 * w) * 1024, d_a1 = d_a0 + 1024;
  const int d_b = I8_SLOT_A + w * 1024;
  auto plan = [&](int t, int slot) {
    const int sb = slot * I8_SLOT;
    cd0 = sb + d_a0; cd1 = sb + d_a1; cd2 = sb + d_b; cd3 = sb + d_b + I8_SLOT_B; cd4 = sb + d_b + 2 * I8_SLOT_B;
    cp_on = true;
    if (t >= nb) {                                         // LoRA stage: 32 fp16 = 64 B per row
      const int tl = t - nb, off = (tl >> 1) * 64;
      // (arithmetic instead of a select between the two pointers: hipcc turns the select into a table in scratch memory)
      const char* A = reinterpret_cast<const char*>(g.thi) +
                      (int64_t)(tl & 1) * (reinterpret_cast<const char*>(g.tlo) - reinterpret_cast<const char*>(g.thi)) +
                      (int64_t)lp_bm * g.Rp * 2 + off;
      cb0 = A; cb1 = A + (int64_t)16 * g.Rp * 2;
      cb2 = reinterpret_cast<const char*>(g.Bhi) + (int64_t)lp_bn * g.Rp * 2 + off;
      cb3 = reinterpret_cast<const char*>(g.Blo) + (int64_t)lp_bn * g.Rp * 2 + off;       // LORA1 ignores it
      cb4 = cb3; cd4 = cd3;
      offA = a_off + dt_off; offB = w_off + dl_off;
      return;
    }
    const int ka = t * I8_GK;
    const char* A = reinterpret_cast<const char*>(g.qx) + (int64_t)lp_bm * g.Kp + ka;
    cb0 = A; cb1 = A + (int64_t)16 * g.Kp;
    cb2 = reinterpret_cast<const char*>(g.W) + (int64_t)lp_bn * g.Kp + ka;
    if (NL == 3) { cb3 = cb2 + g.plane_stride; cb4 = cb3 + g.plane_stride; }
    else { cb3 = cb2; cd3 = cd2; cb4 = cb2; cd4 = cd2; }
    offA = a_off; offB = w_off;
  };
#define SPQ_I8_PIECE(J)                                                                              \
  do {                                                                                               \
    if ((J) < CPN && cp_on && !(V & 2))                                                              \
      i8_glds16(((J) == 0 ? cb0 : (J) == 1 ? cb1 : (J) == 2 ? cb2 : (J) == 3 ? cb3 : cb4) + ((J) < 2 ? offA : offB), \
                smem + ((J) == 0 ? cd0 : (J) == 1 ? cd1 : (J) == 2 ? cd2 : (J) == 3 ? cd3 : cd4));  \
  } while (0)

  // ---- fragment addressing (64-B rows): lane (r = l31, h) reads the 16-B chunk c of row r at position c ^ ((r >> 2) & 3)
  const int s3 = (l31 >> 2) & 3;
  const int fa = (wm * 64 + l31) * 64;                     // + tm * 2048
  const int fb = I8_SLOT_A + (wn * 64 + l31) * 64;         // + tn * 2048 + plane * I8_SLOT_B

  i32x16 acc[NL][2][2];
  auto zero_acc = [&]() {
#pragma unroll
    for (int p = 0; p < NL; ++p)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int jj = 0; jj < 2; ++jj)
#pragma unroll
          for (int e = 0; e < 16; ++e) acc[p][i][jj][e] = 0;
  };
  // read through the fp16 vector type: hipcc puts s_waitcnt vmcnt(0) (draining the copies in flight) in front of an
  // integer-typed LDS read that follows an LDS-DMA, but not in front of a half-typed one (measured on the f16 kernels)
  auto ld16 = [&](const char* p) -> i32x4v { return __builtin_bit_cast(i32x4v, *reinterpret_cast<const h16x8*>(p)); };
  auto base_stage = [&](const char* sb) {
#pragma unroll
    for (int s = 0; s < 2; ++s) {                          // two k32 steps; lane half h owns bytes 32 s + 16 h .. + 15
      const int c = ((2 * s + h) ^ s3) << 4;
      i32x4v a[2];
      a[0] = ld16(sb + fa + c); a[1] = ld16(sb + fa + 2048 + c);
#pragma unroll
      for (int p = 0; p < NL; ++p) {
        i32x4v b[2];
        b[0] = ld16(sb + fb + p * I8_SLOT_B + c); b[1] = ld16(sb + fb + p * I8_SLOT_B + 2048 + c);
#pragma unroll
        for (int tm = 0; tm < 2; ++tm)
#pragma unroll
          for (int tn = 0; tn < 2; ++tn)
            acc[p][tm][tn] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[tm], b[tn], acc[p][tm][tn], 0, 0, 0);
        if (NL == 3) {
          if (s == 0) { if (p == 0) SPQ_I8_PIECE(0); if (p == 1) SPQ_I8_PIECE(1); if (p == 2) SPQ_I8_PIECE(2); }
          else { if (p == 0) SPQ_I8_PIECE(3); if (p == 1) SPQ_I8_PIECE(4); }
        } else {
          if (s == 0) { SPQ_I8_PIECE(0); SPQ_I8_PIECE(1); } else { SPQ_I8_PIECE(2); SPQ_I8_PIECE(3); }
        }
      }
    }
  };

  // cursors: compute (cp, cbm, cbn); load (lp, lt, lbm, lbn) runs two stages ahead
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
  SPQ_I8_PIECE(0); SPQ_I8_PIECE(1); SPQ_I8_PIECE(2); SPQ_I8_PIECE(3); SPQ_I8_PIECE(4);
  int n1 = 0;
  if (lvalid) {
    plan(lt, 1); advance_load(); n1 = CPN;
    SPQ_I8_PIECE(0); SPQ_I8_PIECE(1); SPQ_I8_PIECE(2); SPQ_I8_PIECE(3); SPQ_I8_PIECE(4);
  }
  auto wait_keep = [&](int n) {                            // all but the n (= CPN or 0) youngest copies have landed
    if (n == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if (CPN == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  };
  wait_keep(n1);
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");

  int slot = 0;
  auto pre = [&]() -> int {                                // put S_{i+2} on the plan; returns its copy count
    cp_on = false;
    if (lvalid) { plan(lt, slot == 0 ? 2 : slot - 1); advance_load(); }
    return ((V & 2) || !cp_on) ? 0 : CPN;
  };
  auto post = [&](int n2, bool more) {                     // S_{i+1} complete: only S_{i+2}'s copies may be outstanding
    wait_keep(n2);
    if (more) { __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); }
    slot = slot == 2 ? 0 : slot + 1;
  };
  auto drain_pieces = [&]() { SPQ_I8_PIECE(0); SPQ_I8_PIECE(1); SPQ_I8_PIECE(2); SPQ_I8_PIECE(3); SPQ_I8_PIECE(4); };

#pragma unroll 1
  while (true) {
    const bool more_tiles = cp + gstride < nwg;
    zero_acc();
    for (int t = 0; t < nb; ++t) {
      const int n2 = pre();
      if (!(V & 8)) base_stage(smem + slot * I8_SLOT); else drain_pieces();
      post(n2, t + 1 < T || more_tiles);
    }
    // ---- merge the digit sums: base[m,n] * 2^e[n] = 65536 a2 + 256 a1 + a0 (each conversion exact: |a_j| < 2^24)
    fl32x16 base[2][2];
#pragma unroll
    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
      for (int tn = 0; tn < 2; ++tn)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          if (NL == 3)
            base[tm][tn][e] = (float)acc[NL - 1][tm][tn][e] * 65536.f + ((float)acc[NL == 3 ? 1 : 0][tm][tn][e] * 256.f + (float)acc[0][tm][tn][e]);
          else
            base[tm][tn][e] = (float)acc[0][tm][tn][e];
        }
    // per-lane column constants of the epilogue (col = l31 of the 32-wide tile), fetched while the LoRA stages run
    float rs[2], bs[2], bv[2];
#pragma unroll
    for (int tn = 0; tn < 2; ++tn) {
      const int n = cbn + wn * 64 + tn * 32 + l31;
      rs[tn] = g.rowscale[n];                              // padded to Np
      bs[tn] = nl > 0 ? g.bscale[n] : 0.f;
      bv[tn] = (g.bias && n < g.N) ? g.bias[n] : 0.f;
    }
    if (nl > 0) {
      fl32x16 u[2][2];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int jj = 0; jj < 2; ++jj)
#pragma unroll
          for (int e = 0; e < 16; ++e) u[i][jj][e] = 0.f;
      auto lora_stage = [&](const char* sb, bool two) {
#pragma unroll
        for (int s = 0; s < 2; ++s) {                      // two k16 blocks of the 32-wide block; k = 16 s + 8 h
          h16x8 a[2], bh[2], bl[2];
          const int c = ((2 * s + h) ^ s3) << 4;
#pragma unroll
          for (int t = 0; t < 2; ++t) {
            a[t] = *reinterpret_cast<const h16x8*>(sb + fa + t * 2048 + c);
            bh[t] = *reinterpret_cast<const h16x8*>(sb + fb + t * 2048 + c);
            if (two) bl[t] = *reinterpret_cast<const h16x8*>(sb + fb + I8_SLOT_B + t * 2048 + c);
          }
#pragma unroll
          for (int tm = 0; tm < 2; ++tm)
#pragma unroll
            for (int tn = 0; tn < 2; ++tn) {
              u[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[tm], bh[tn], u[tm][tn], 0, 0, 0);
              if (two) u[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[tm], bl[tn], u[tm][tn], 0, 0, 0);
            }
          if (s == 0) { SPQ_I8_PIECE(0); SPQ_I8_PIECE(1); SPQ_I8_PIECE(2); } else { SPQ_I8_PIECE(3); SPQ_I8_PIECE(4); }
        }
      };
      for (int t = nb; t < T; t += 2) {
        int n2 = pre();
        if (!(V & 8)) lora_stage(smem + slot * I8_SLOT, true); else drain_pieces();
        post(n2, true);
        n2 = pre();
        if (!(V & 8)) lora_stage(smem + slot * I8_SLOT, false); else drain_pieces();
        post(n2, t + 2 < T || more_tiles);
      }
      // y = base * rowscale[n] + (u * 2^-g[m]) * bscale[n] + bias[n]
#pragma unroll
      for (int tm = 0; tm < 2; ++tm)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int m = cbm + wm * 64 + tm * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
          const float ri = g.rowinv[m];
#pragma unroll
          for (int tn = 0; tn < 2; ++tn) base[tm][tn][e] = base[tm][tn][e] * rs[tn] + (u[tm][tn][e] * ri) * bs[tn] + bv[tn];
        }
    } else {
#pragma unroll
      for (int tm = 0; tm < 2; ++tm)
#pragma unroll
        for (int tn = 0; tn < 2; ++tn)
#pragma unroll
          for (int e = 0; e < 16; ++e) base[tm][tn][e] = base[tm][tn][e] * rs[tn] + bv[tn];
    }
    {
      // ---- epilogue: 8 rows x 32 cols at a time through a private LDS slice, 16-B stores of whole 128-B lines
      char* eb = smem + 3 * I8_SLOT + w * I8_EPI_WAVE;
      const int c4 = (lane & 7) * 4, r8 = lane >> 3;
      const bool interior = (cbm + I8_GM <= g.M) && (cbn + I8_GN <= g.N);
#pragma unroll
      for (int tn = 0; tn < 2; ++tn) {
        const int n = cbn + wn * 64 + tn * 32 + c4;
        const bool n_ok = n < g.N;
#pragma unroll
        for (int tm = 0; tm < 2; ++tm)
#pragma unroll
          for (int q = 0; q < 4; ++q) {                    // rows 8q .. 8q+7 of the 32x32 tile: registers 4q..4q+3
#pragma unroll
            for (int e4 = 0; e4 < 4; ++e4)
              *reinterpret_cast<float*>(eb + (e4 + 4 * h) * 144 + l31 * 4) = base[tm][tn][4 * q + e4];
            float4 o = *reinterpret_cast<const float4*>(eb + r8 * 144 + c4 * 4);
            const int m = cbm + wm * 64 + tm * 32 + 8 * q + r8;
            if (EPI == 1) { o.x = i8_gelu_erf(o.x); o.y = i8_gelu_erf(o.y); o.z = i8_gelu_erf(o.z); o.w = i8_gelu_erf(o.w); }
            float* dst = g.y + (int64_t)m * g.N + n;
            if (V & 16) { if (o.x == 12345.f) *reinterpret_cast<float4*>(dst) = o; }
            else if (interior) *reinterpret_cast<float4*>(dst) = o;
            else if (m < g.M) {
              if (n + 3 < g.N && (g.N & 3) == 0) *reinterpret_cast<float4*>(dst) = o;
              else { if (n < g.N) dst[0] = o.x; if (n + 1 < g.N) dst[1] = o.y; if (n + 2 < g.N) dst[2] = o.z; if (n + 3 < g.N) dst[3] = o.w; }
            }
            (void)n_ok;
          }
      }
    }
    if (!more_tiles) break;
    cp += gstride;
    tile_of(cp, cbm, cbn);
  }
#undef SPQ_I8_PIECE
}


// -------------------------------------------------------------------------------------------------------------------
// 128 x 128 tiles, 4 waves (2 x 2 of 64 x 64), NBUF stage buffers of 32 KB (A 128 x 64 B + three 128 x 64 B planes): two
// workgroups share a CU and cover each other's copy latency, barriers, tile prologues and epilogues -- the structure that
// measured best for the fp16-limb operands (gemm_f16x2_t128_kernel), here with 25 % fewer matrix-pipe cycles and a third
// fewer bytes through LDS per product.  NBUF = 1: wait + barrier, MFMAs, barrier, issue the next stage into the same
// buffer.  NBUF = 2: the next stage is issued BEFORE the MFMAs of the current one (other buffer) and waited for after them.
// -------------------------------------------------------------------------------------------------------------------
constexpr int I8T_A = 128 * 64;                             // 8 KB
constexpr int I8T_STAGE = I8T_A + 3 * I8_SLOT_B;            // 32 KB
constexpr int I8T_EPI_WAVE = 8 * 144;
constexpr int i8t_lds(int nbuf) { return nbuf * I8T_STAGE + 4 * I8T_EPI_WAVE; }

template <int NL, int NBUF, int EPI>
__global__ __launch_bounds__(256, 2) void gemm_i8_t128_kernel(GemmI8Args g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = w >> 1, wn = w & 1;
  const int l31 = lane & 31, h = lane >> 5;
  const int tiles_m = g.tiles_m * 2;                        // g.tiles_m counts 256-row tiles (Mp is a multiple of 256)
  const int nwg = tiles_m * g.tiles_n;
  const int nb = g.Kp / I8_GK;
  const int nl = (g.Rp / 32) * 2;
  const int T = nb + nl;
  const int gstride = (int)gridDim.x;
  auto tile_of = [&](int p, int& bm, int& bn) {             // XCD-aware band order, 8 tile rows per band
    const int q8 = nwg >> 3, r8 = nwg & 7, xcd = p & 7;
    const int wgid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (p >> 3);
    constexpr int GROUP_M = 8;
    const int band = wgid / (GROUP_M * g.tiles_n);
    const int band_rows = min(GROUP_M, tiles_m - band * GROUP_M);
    const int in_band = wgid - band * GROUP_M * g.tiles_n;
    bm = (band * GROUP_M + in_band % band_rows) * 128;
    bn = (in_band / band_rows) * I8_GN;
  };
  int p = blockIdx.x;
  if (p >= nwg) return;
  int bm, bn;
  tile_of(p, bm, bn);

  // copies: wave w owns pieces 2w, 2w+1 (rows 32w .. 32w+31) of A and of every plane
  const int r64 = lane >> 2, p64 = lane & 3;
  const int crow = 32 * w + r64;                            // piece 2w; piece 2w+1 is 16 rows further (same swizzle key)
  const int cchunk = (p64 ^ ((crow >> 2) & 3)) * 16;
  const int off_k = crow * g.Kp + cchunk;                   // byte offsets: int8 rows of Kp bytes
  const int off_r = crow * g.Rp * 2 + cchunk;               // fp16 rows of Rp elements
  const int dstw = (2 * w) * 1024;
  auto issue = [&](int t, int tbm, int tbn, int buf) {
    char* sb = smem + buf * I8T_STAGE;
    if (t < nb) {
      const int ka = t * I8_GK;
      const char* A = reinterpret_cast<const char*>(g.qx) + (int64_t)tbm * g.Kp + ka + off_k;
      i8_glds16(A, sb + dstw); i8_glds16(A + (int64_t)16 * g.Kp, sb + dstw + 1024);
      const char* B = reinterpret_cast<const char*>(g.W) + (int64_t)tbn * g.Kp + ka + off_k;
#pragma unroll
      for (int pl = 0; pl < NL; ++pl) {
        i8_glds16(B + pl * g.plane_stride, sb + I8T_A + pl * I8_SLOT_B + dstw);
        i8_glds16(B + pl * g.plane_stride + (int64_t)16 * g.Kp, sb + I8T_A + pl * I8_SLOT_B + dstw + 1024);
      }
    } else {
      const int tl = t - nb, off = (tl >> 1) * 64;
      const char* A = reinterpret_cast<const char*>(g.thi) +
                      (int64_t)(tl & 1) * (reinterpret_cast<const char*>(g.tlo) - reinterpret_cast<const char*>(g.thi)) +
                      (int64_t)tbm * g.Rp * 2 + off + off_r;
      i8_glds16(A, sb + dstw); i8_glds16(A + (int64_t)16 * g.Rp * 2, sb + dstw + 1024);
      const char* Bh = reinterpret_cast<const char*>(g.Bhi) + (int64_t)tbn * g.Rp * 2 + off + off_r;
      i8_glds16(Bh, sb + I8T_A + dstw); i8_glds16(Bh + (int64_t)16 * g.Rp * 2, sb + I8T_A + dstw + 1024);
      if (!(tl & 1)) {
        const char* Bl = reinterpret_cast<const char*>(g.Blo) + (int64_t)tbn * g.Rp * 2 + off + off_r;
        i8_glds16(Bl, sb + I8T_A + I8_SLOT_B + dstw); i8_glds16(Bl + (int64_t)16 * g.Rp * 2, sb + I8T_A + I8_SLOT_B + dstw + 1024);
      }
    }
  };

  const int s3 = (l31 >> 2) & 3;
  const int fa = (wm * 64 + l31) * 64;
  const int fb = I8T_A + (wn * 64 + l31) * 64;
  auto ld16 = [&](const char* q) -> i32x4v { return __builtin_bit_cast(i32x4v, *reinterpret_cast<const h16x8*>(q)); };

  // stage sequencing.  `cur` = buffer of the stage being computed.
  int cur = 0;
  auto stage_begin = [&](bool have_next, int nt, int nbm, int nbn) {
    if (NBUF == 1) {
      __syncthreads();                                       // vmcnt(0) + barrier: the stage has landed
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // my pieces of this stage have landed ...
      __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory");   // ... everybody's have, and the other buffer is free
      if (have_next) issue(nt, nbm, nbn, cur ^ 1);           // in flight under this stage's MFMAs
    }
  };
  auto stage_end = [&](bool have_next, int nt, int nbm, int nbn) {
    if (NBUF == 1) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // my fragment reads are complete (and may not sink below)
      __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory");   // every wave has read its fragments
      if (have_next) issue(nt, nbm, nbn, 0);
    } else {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      cur ^= 1;
    }
  };

  issue(0, bm, bn, 0);
  while (true) {
    const int pn = p + gstride;
    const bool more = pn < nwg;
    int nbm = 0, nbn = 0;
    if (more) tile_of(pn, nbm, nbn);
    i32x16 acc[NL][2][2];
#pragma unroll
    for (int pl = 0; pl < NL; ++pl)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int jj = 0; jj < 2; ++jj)
#pragma unroll
          for (int e = 0; e < 16; ++e) acc[pl][i][jj][e] = 0;
    for (int t = 0; t < nb; ++t) {
      const bool last = (t + 1 == T);
      const bool have_next = !last || more;
      const int nt = last ? 0 : t + 1, xbm = last ? nbm : bm, xbn = last ? nbn : bn;
      stage_begin(have_next, nt, xbm, xbn);
      const char* sb = smem + (NBUF == 1 ? 0 : cur) * I8T_STAGE;
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const int c = ((2 * s + h) ^ s3) << 4;
        i32x4v a[2];
        a[0] = ld16(sb + fa + c); a[1] = ld16(sb + fa + 2048 + c);
#pragma unroll
        for (int pl = 0; pl < NL; ++pl) {
          i32x4v b[2];
          b[0] = ld16(sb + fb + pl * I8_SLOT_B + c); b[1] = ld16(sb + fb + pl * I8_SLOT_B + 2048 + c);
#pragma unroll
          for (int tm = 0; tm < 2; ++tm)
#pragma unroll
            for (int tn = 0; tn < 2; ++tn)
              acc[pl][tm][tn] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[tm], b[tn], acc[pl][tm][tn], 0, 0, 0);
        }
      }
      stage_end(have_next, nt, xbm, xbn);
    }
    fl32x16 base[2][2];
#pragma unroll
    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
      for (int tn = 0; tn < 2; ++tn)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          if (NL == 3)
            base[tm][tn][e] = (float)acc[NL - 1][tm][tn][e] * 65536.f + ((float)acc[NL == 3 ? 1 : 0][tm][tn][e] * 256.f + (float)acc[0][tm][tn][e]);
          else
            base[tm][tn][e] = (float)acc[0][tm][tn][e];
        }
    float rs[2], bs[2], bv[2];
#pragma unroll
    for (int tn = 0; tn < 2; ++tn) {
      const int n = bn + wn * 64 + tn * 32 + l31;
      rs[tn] = g.rowscale[n];
      bs[tn] = nl > 0 ? g.bscale[n] : 0.f;
      bv[tn] = (g.bias && n < g.N) ? g.bias[n] : 0.f;
    }
    if (nl > 0) {
      fl32x16 u[2][2];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int jj = 0; jj < 2; ++jj)
#pragma unroll
          for (int e = 0; e < 16; ++e) u[i][jj][e] = 0.f;
      for (int t = nb; t < T; ++t) {
        const bool two = !((t - nb) & 1);
        const bool last = (t + 1 == T);
        const bool have_next = !last || more;
        const int nt = last ? 0 : t + 1, xbm = last ? nbm : bm, xbn = last ? nbn : bn;
        stage_begin(have_next, nt, xbm, xbn);
        const char* sb = smem + (NBUF == 1 ? 0 : cur) * I8T_STAGE;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          h16x8 a[2], bh[2], bl[2];
          const int c = ((2 * s + h) ^ s3) << 4;
#pragma unroll
          for (int q = 0; q < 2; ++q) {
            a[q] = *reinterpret_cast<const h16x8*>(sb + fa + q * 2048 + c);
            bh[q] = *reinterpret_cast<const h16x8*>(sb + fb + q * 2048 + c);
            if (two) bl[q] = *reinterpret_cast<const h16x8*>(sb + fb + I8_SLOT_B + q * 2048 + c);
          }
#pragma unroll
          for (int tm = 0; tm < 2; ++tm)
#pragma unroll
            for (int tn = 0; tn < 2; ++tn) {
              u[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[tm], bh[tn], u[tm][tn], 0, 0, 0);
              if (two) u[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[tm], bl[tn], u[tm][tn], 0, 0, 0);
            }
        }
        stage_end(have_next, nt, xbm, xbn);
      }
#pragma unroll
      for (int tm = 0; tm < 2; ++tm)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int m = bm + wm * 64 + tm * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
          const float ri = g.rowinv[m];
#pragma unroll
          for (int tn = 0; tn < 2; ++tn) base[tm][tn][e] = base[tm][tn][e] * rs[tn] + (u[tm][tn][e] * ri) * bs[tn] + bv[tn];
        }
    } else {
#pragma unroll
      for (int tm = 0; tm < 2; ++tm)
#pragma unroll
        for (int tn = 0; tn < 2; ++tn)
#pragma unroll
          for (int e = 0; e < 16; ++e) base[tm][tn][e] = base[tm][tn][e] * rs[tn] + bv[tn];
    }
    {
      char* eb = smem + NBUF * I8T_STAGE + w * I8T_EPI_WAVE;
      const int c4 = (lane & 7) * 4, r8 = lane >> 3;
      const bool interior = (bm + 128 <= g.M) && (bn + I8_GN <= g.N);
#pragma unroll
      for (int tn = 0; tn < 2; ++tn) {
        const int n = bn + wn * 64 + tn * 32 + c4;
#pragma unroll
        for (int tm = 0; tm < 2; ++tm)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
#pragma unroll
            for (int e4 = 0; e4 < 4; ++e4)
              *reinterpret_cast<float*>(eb + (e4 + 4 * h) * 144 + l31 * 4) = base[tm][tn][4 * q + e4];
            float4 o = *reinterpret_cast<const float4*>(eb + r8 * 144 + c4 * 4);
            const int m = bm + wm * 64 + tm * 32 + 8 * q + r8;
            if (EPI == 1) { o.x = i8_gelu_erf(o.x); o.y = i8_gelu_erf(o.y); o.z = i8_gelu_erf(o.z); o.w = i8_gelu_erf(o.w); }
            float* dst = g.y + (int64_t)m * g.N + n;
            if (interior) *reinterpret_cast<float4*>(dst) = o;
            else if (m < g.M) {
              if (n + 3 < g.N && (g.N & 3) == 0) *reinterpret_cast<float4*>(dst) = o;
              else { if (n < g.N) dst[0] = o.x; if (n + 1 < g.N) dst[1] = o.y; if (n + 2 < g.N) dst[2] = o.z; if (n + 3 < g.N) dst[3] = o.w; }
            }
          }
      }
    }
    if (!more) break;
    p = pn; bm = nbm; bn = nbn;
  }
}


// -------------------------------------------------------------------------------------------------------------------
// 128 x 128 tiles, 4 waves, ONE stage buffer, 128-deep stages: rows of 128 B (whole cache lines: the 64-B rows of the kernels
// above make every copy request half a line) and half as many stages per tile.  Measured on the kernels above: time per
// stage (~1.8 us per workgroup) hardly depends on what the stage holds -- copy issue -> landed latency and the two barriers
// dominate it -- so fewer, fatter stages are what pays.  LDS 64 KB + 4.6 KB: two workgroups per CU cover each other's waits.
// Base stage: A 128 x 128 B + NL planes 128 x 128 B, 48 (NL = 3) MFMAs per wave.  LoRA stages: 64 fp16 = 128 B per row:
// (thi x {Bhi, Blo}) then (tlo x Bhi) per 64-wide block of r.  Needs Kp % 128 == 0.
// -------------------------------------------------------------------------------------------------------------------
constexpr int I8K_PLANE = 128 * 128;                        // 16 KB: A, or one plane / LoRA-B limb
constexpr int i8k_lds(int nl) { return (1 + (nl == 3 ? 3 : 2)) * I8K_PLANE + 4 * I8T_EPI_WAVE; }   // NL = 1 still stages Bhi + Blo

template <int NL, int EPI>
__global__ __launch_bounds__(256, 2) void gemm_i8_k128_kernel(GemmI8Args g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int NPL = NL == 3 ? 3 : 2;                      // B-side 16-KB regions in the stage buffer
  constexpr int STAGE = (1 + NPL) * I8K_PLANE;
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = w >> 1, wn = w & 1;
  const int l31 = lane & 31, h = lane >> 5;
  const int tiles_m = g.tiles_m * 2;                        // g.tiles_m counts 256-row tiles (Mp is a multiple of 256)
  const int nwg = tiles_m * g.tiles_n;
  const int nb = g.Kp / 128;
  const int nl = (g.Rp / 64) * 2;
  const int T = nb + nl;
  const int gstride = (int)gridDim.x;
  auto tile_of = [&](int p, int& bm, int& bn) {             // XCD-aware band order, 8 tile rows per band
    const int q8 = nwg >> 3, r8 = nwg & 7, xcd = p & 7;
    const int wgid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (p >> 3);
    constexpr int GROUP_M = 8;
    const int band = wgid / (GROUP_M * g.tiles_n);
    const int band_rows = min(GROUP_M, tiles_m - band * GROUP_M);
    const int in_band = wgid - band * GROUP_M * g.tiles_n;
    bm = (band * GROUP_M + in_band % band_rows) * 128;
    bn = (in_band / band_rows) * I8_GN;
  };
  int p = blockIdx.x;
  if (p >= nwg) return;
  int bm, bn;
  tile_of(p, bm, bn);

  // copies: 1-KB pieces = 8 rows x 128 B; wave w owns rows 32w .. 32w+31 of every 128-row operand = pieces 4w .. 4w+3.
  // Piece i of the wave: row 32w + 8i + (lane >> 3), source chunk (lane & 7) ^ ((row >> 1) & 7); (row >> 1) & 7 = (4i + (lane >> 4)) & 7,
  // so pieces i and i + 2 share the key and pieces i, i + 1 differ by chunk ^ 4.
  const int prow = 32 * w + (lane >> 3);
  const int key0 = (lane >> 4) & 7;                         // pieces 0, 2
  const int ch0 = ((lane & 7) ^ key0) * 16, ch1 = ((lane & 7) ^ key0 ^ 4) * 16;
  const int ok0 = prow * g.Kp + ch0, ok1 = (prow + 8) * g.Kp + ch1;          // int8 operands: rows of Kp bytes
  const int or0 = prow * g.Rp * 2 + ch0, or1 = (prow + 8) * g.Rp * 2 + ch1;  // fp16 operands: rows of Rp elements
  const int dstw = (4 * w) * 1024;
  auto copy4 = [&](const char* base, int o0, int o1, int64_t ld16, char* dst) {   // the wave's four pieces of one operand
    i8_glds16(base + o0, dst); i8_glds16(base + o1, dst + 1024);
    i8_glds16(base + ld16 + o0, dst + 2048); i8_glds16(base + ld16 + o1, dst + 3072);
  };
  auto issue = [&](int t, int tbm, int tbn) {
    char* sb = smem;
    if (t < nb) {
      const int ka = t * 128;
      const int64_t ld16 = (int64_t)16 * g.Kp;
      copy4(reinterpret_cast<const char*>(g.qx) + (int64_t)tbm * g.Kp + ka, ok0, ok1, ld16, sb + dstw);
      const char* B = reinterpret_cast<const char*>(g.W) + (int64_t)tbn * g.Kp + ka;
#pragma unroll
      for (int pl = 0; pl < NL; ++pl) copy4(B + pl * g.plane_stride, ok0, ok1, ld16, sb + (1 + pl) * I8K_PLANE + dstw);
    } else {
      const int tl = t - nb, off = (tl >> 1) * 128;
      const int64_t ld16 = (int64_t)16 * g.Rp * 2;
      const char* A = reinterpret_cast<const char*>(g.thi) +
                      (int64_t)(tl & 1) * (reinterpret_cast<const char*>(g.tlo) - reinterpret_cast<const char*>(g.thi)) +
                      (int64_t)tbm * g.Rp * 2 + off;
      copy4(A, or0, or1, ld16, sb + dstw);
      copy4(reinterpret_cast<const char*>(g.Bhi) + (int64_t)tbn * g.Rp * 2 + off, or0, or1, ld16, sb + I8K_PLANE + dstw);
      if (!(tl & 1)) copy4(reinterpret_cast<const char*>(g.Blo) + (int64_t)tbn * g.Rp * 2 + off, or0, or1, ld16, sb + 2 * I8K_PLANE + dstw);
    }
  };

  const int s7 = (l31 >> 1) & 7;
  const int fa = (wm * 64 + l31) * 128;                     // + tm * 4096
  const int fb = I8K_PLANE + (wn * 64 + l31) * 128;         // + tn * 4096 + plane * I8K_PLANE
  auto ld16v = [&](const char* q) -> i32x4v { return __builtin_bit_cast(i32x4v, *reinterpret_cast<const h16x8*>(q)); };
  auto stage_begin = [&]() { __syncthreads(); };            // vmcnt(0) + barrier: the stage has landed
  auto stage_end = [&](bool have_next, int nt, int nbm, int nbn) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // my fragment reads are complete (and may not sink below)
    __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory");   // every wave has read its fragments
    if (have_next) issue(nt, nbm, nbn);
  };

  issue(0, bm, bn);
  while (true) {
    const int pn = p + gstride;
    const bool more = pn < nwg;
    int nbm = 0, nbn = 0;
    if (more) tile_of(pn, nbm, nbn);
    i32x16 acc[NL][2][2];
#pragma unroll
    for (int pl = 0; pl < NL; ++pl)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int jj = 0; jj < 2; ++jj)
#pragma unroll
          for (int e = 0; e < 16; ++e) acc[pl][i][jj][e] = 0;
    for (int t = 0; t < nb; ++t) {
      const bool last = (t + 1 == T);
      stage_begin();
#pragma unroll
      for (int s = 0; s < 4; ++s) {                          // four k32 steps; lane half h owns bytes 32 s + 16 h .. + 15
        const int c = ((2 * s + h) ^ s7) << 4;
        i32x4v a[2];
        a[0] = ld16v(smem + fa + c); a[1] = ld16v(smem + fa + 4096 + c);
#pragma unroll
        for (int pl = 0; pl < NL; ++pl) {
          i32x4v b[2];
          b[0] = ld16v(smem + fb + pl * I8K_PLANE + c); b[1] = ld16v(smem + fb + pl * I8K_PLANE + 4096 + c);
#pragma unroll
          for (int tm = 0; tm < 2; ++tm)
#pragma unroll
            for (int tn = 0; tn < 2; ++tn)
              acc[pl][tm][tn] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[tm], b[tn], acc[pl][tm][tn], 0, 0, 0);
        }
      }
      stage_end(!last || more, last ? 0 : t + 1, last ? nbm : bm, last ? nbn : bn);
    }
    fl32x16 base[2][2];
#pragma unroll
    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
      for (int tn = 0; tn < 2; ++tn)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          if (NL == 3)
            base[tm][tn][e] = (float)acc[NL - 1][tm][tn][e] * 65536.f + ((float)acc[NL == 3 ? 1 : 0][tm][tn][e] * 256.f + (float)acc[0][tm][tn][e]);
          else
            base[tm][tn][e] = (float)acc[0][tm][tn][e];
        }
    float rs[2], bs[2], bv[2];
#pragma unroll
    for (int tn = 0; tn < 2; ++tn) {
      const int n = bn + wn * 64 + tn * 32 + l31;
      rs[tn] = g.rowscale[n];
      bs[tn] = nl > 0 ? g.bscale[n] : 0.f;
      bv[tn] = (g.bias && n < g.N) ? g.bias[n] : 0.f;
    }
    if (nl > 0) {
      fl32x16 u[2][2];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int jj = 0; jj < 2; ++jj)
#pragma unroll
          for (int e = 0; e < 16; ++e) u[i][jj][e] = 0.f;
      for (int t = nb; t < T; ++t) {
        const bool two = !((t - nb) & 1);
        const bool last = (t + 1 == T);
        stage_begin();
#pragma unroll
        for (int s = 0; s < 4; ++s) {                        // four k16 steps of the 64-wide block; k = 16 s + 8 h
          h16x8 a[2], bh[2], bl[2];
          const int c = ((2 * s + h) ^ s7) << 4;
#pragma unroll
          for (int q = 0; q < 2; ++q) {
            a[q] = *reinterpret_cast<const h16x8*>(smem + fa + q * 4096 + c);
            bh[q] = *reinterpret_cast<const h16x8*>(smem + fb + q * 4096 + c);
            if (two) bl[q] = *reinterpret_cast<const h16x8*>(smem + fb + I8K_PLANE + q * 4096 + c);
          }
#pragma unroll
          for (int tm = 0; tm < 2; ++tm)
#pragma unroll
            for (int tn = 0; tn < 2; ++tn) {
              u[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[tm], bh[tn], u[tm][tn], 0, 0, 0);
              if (two) u[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[tm], bl[tn], u[tm][tn], 0, 0, 0);
            }
        }
        stage_end(!last || more, last ? 0 : t + 1, last ? nbm : bm, last ? nbn : bn);
      }
#pragma unroll
      for (int tm = 0; tm < 2; ++tm)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int m = bm + wm * 64 + tm * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
          const float ri = g.rowinv[m];
#pragma unroll
          for (int tn = 0; tn < 2; ++tn) base[tm][tn][e] = base[tm][tn][e] * rs[tn] + (u[tm][tn][e] * ri) * bs[tn] + bv[tn];
        }
    } else {
#pragma unroll
      for (int tm = 0; tm < 2; ++tm)
#pragma unroll
        for (int tn = 0; tn < 2; ++tn)
#pragma unroll
          for (int e = 0; e < 16; ++e) base[tm][tn][e] = base[tm][tn][e] * rs[tn] + bv[tn];
    }
    {
      char* eb = smem + STAGE + w * I8T_EPI_WAVE;
      const int c4 = (lane & 7) * 4, r8 = lane >> 3;
      const bool interior = (bm + 128 <= g.M) && (bn + I8_GN <= g.N);
#pragma unroll
      for (int tn = 0; tn < 2; ++tn) {
        const int n = bn + wn * 64 + tn * 32 + c4;
#pragma unroll
        for (int tm = 0; tm < 2; ++tm)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
#pragma unroll
            for (int e4 = 0; e4 < 4; ++e4)
              *reinterpret_cast<float*>(eb + (e4 + 4 * h) * 144 + l31 * 4) = base[tm][tn][4 * q + e4];
            float4 o = *reinterpret_cast<const float4*>(eb + r8 * 144 + c4 * 4);
            const int m = bm + wm * 64 + tm * 32 + 8 * q + r8;
            if (EPI == 1) { o.x = i8_gelu_erf(o.x); o.y = i8_gelu_erf(o.y); o.z = i8_gelu_erf(o.z); o.w = i8_gelu_erf(o.w); }
            float* dst = g.y + (int64_t)m * g.N + n;
            if (interior) *reinterpret_cast<float4*>(dst) = o;
            else if (m < g.M) {
              if (n + 3 < g.N && (g.N & 3) == 0) *reinterpret_cast<float4*>(dst) = o;
              else { if (n < g.N) dst[0] = o.x; if (n + 1 < g.N) dst[1] = o.y; if (n + 2 < g.N) dst[2] = o.z; if (n + 3 < g.N) dst[3] = o.w; }
            }
          }
      }
    }
    if (!more) break;
    p = pn; bm = nbm; bn = nbn;
  }
}

}  // namespace spq
