// SPQ_PATH_F16X2: the dense contraction on f16 matrix cores at fp32 accuracy.
//
// For a symmetric minmax input quantizer the fake-quantised activation is  FQ(x)[m,k] = q[m,k] * sx[k]  with q an
// INTEGER level (|q| <= 2^(b-1)-1).  The per-channel scale sx[k] lives on the contraction axis, so it is folded into
// the weight operand once per (weights, scales):
//        y[m,n] = sum_k q[m,k] * W'[n,k],      W'[n,k] = FQ(W)[n,k] * sx[k]      (exact product, formed in fp64)
// q is exact in fp16 for b <= 12; W' is split into two fp16 limbs  W' * 2^e[n] = hi + lo  (22 significant bits,
// |err| <= 2^-22 |W'|, the per-row power of two e[n] keeps the limbs in fp16's normal range), so
//        y[m,n] = 2^-e[n] * sum_k ( q*hi + q*lo )                                 two f16 MFMAs per k-block,
// every product exact, accumulated in fp32 by the MFMA.  The LoRA branch consumes the RAW fp32 x (lora.py:149):
//        t = x . FQ(A)   on fp32-input MFMA in the activation pass (below),  t * 2^g[m] = thi + tlo,
//        u[m,n] = 2^-g[m] 2^-e[n] * sum_j ( thi*Bhi + thi*Blo + tlo*Bhi ),   B' = scaling * FQ(B)^T * 2^e[n] = Bhi + Blo.
//
// Kernels
//   prep_f16x2_kernel : per output row n: FQ(W) row, fold sx, FQ(B) column, common exponent, limb split   (HBM scan)
//   xpass_kernel      : one read of x: integer levels -> fp16 [M,K]; t = x.FQ(A) (f32 MFMA) -> 2 limbs + row scale
//   gemm_f16x2_kernel : 256x128 tile, 8 waves (4x2, 64x64 each), BK=64, direct global->LDS (16 B), XOR-swizzled,
//                       double-buffered; v_mfma_f32_32x32x16_f16; epilogue scale + bias, full-line stores
#include <hip/hip_fp16.h>
#include <stdlib.h>

#include <mutex>
#include <algorithm>
#include <type_traits>

#include "spq_common.h"
#include "spq_i8_kernel.h"

namespace spq {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

static inline int64_t pad_to(int64_t v, int64_t a) { return (v + a - 1) / a * a; }

// direct global -> LDS copy, 16 B per lane: LDS destination = wave-uniform base + lane*16, source address per lane
__device__ __forceinline__ void glds16(const void* gsrc, void* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

constexpr int GM = 256, GN = 128, GK = 64;     // GEMM block tile
constexpr int XR = 32, XK = 64;                // activation-pass tile: rows per block, k per chunk

struct F16x2Layout {                            // workspace carve-up, all offsets 256-B aligned
  int64_t Mp, Kp, Rp;
  size_t off_qx, off_xl, off_thi, off_tlo, off_rowinv, off_skcnt, off_skpart, total;
};

// Split-K of the contraction (gemm_f16x2_t128_kernel): how many workgroups share a tile's k range.  nwg 128 x 128 tiles of T
// stages each on `slots` resident workgroups (three per CU).  One workgroup's stage takes ~1.2 us + ~0.45 us per workgroup
// resident on its CU (they share the CU's copy path) and a unit costs about three stages on top of its own (operand fetch,
// epilogue or partial-sum exchange): take the S in 1..4 with the smallest estimate, S > 1 only if it wins by 10 %.
// Measured (tools/config_bench.py, M = 8192, contraction alone): K = 3072, N = 768 (384 tiles x 50 stages) 63 -> 47 us with S = 2,
// as estimated; K = 768, N = 768 (14 stages) 29.5 -> 29 us -- launch, fill / drain and the exchange are what is left of a unit
// that short -- and K = 4096, N = 1024 on the three-product path (512 tiles x 130 stages) S = 3 = two rounds of units: 413 ->
// 431 us.  Hence: the automatic choice keeps every unit resident at once (one round) and at least 12 stages long; SPQ_SPLIT_K=2..4
// forces S wherever it is legal (>= 4 stages per unit, the LoRA stages within the first unit, at most two rounds).
static int t128_split(int64_t nwg, int T, int nl, int64_t slots, int forced) {
  if (forced == 0 || nwg <= 0 || nwg >= slots) return 1;
  int best = 1; double best_t = 0;
  for (int S = 1; S <= 4; ++S) {
    const int share = (T + S - 1) / S;
    if (S > 1 && (share < 4 || nl > share || nwg * S > 2 * slots)) break;
    if (forced > 1) { if (S == forced) return S; continue; }
    if (S > 1 && (share < 12 || nwg * S > slots)) break;
    const double per_cu = (double)(nwg * S) / (double)(slots / 3);
    const double t = (double)(share + 3) * (1.2 + 0.45 * per_cu);
    if (S == 1) best_t = t;
    else if (t < 0.9 * best_t) { best = S; best_t = t; }
  }
  return best;
}
unsigned gemm_grid(int ntiles);
int split_k_switch();
// units (workgroups) of the split-K form to reserve scratch for: the largest split the forward could take for this shape under the
// current SPQ_SPLIT_K, over both activation-limb counts and with / without LoRA stages (the workspace is sized before those are known)
static int64_t sk_reserve_units(int64_t M, int64_t K, int64_t N, int64_t r, int al_only) {   // al_only: 1 / 2 activation limbs, 0 = either
  const int64_t nwg = (pad_to(M, 256) / 128) * (pad_to(N, 128) / 128);
  const int64_t slots = 3 * (int64_t)gemm_grid(1 << 30);
  const int forced = split_k_switch();
  if (al_only < 0) return 0;                               // (the int8 kernels have no split-K form)
  const int kb = (int)(pad_to(K, 64) / 64), nlr = r > 0 ? (int)(pad_to(r, 64) / 64) * 2 : 0;
  int S = 1;
  for (int al = (al_only ? al_only : 1); al <= (al_only ? al_only : 2); ++al)
    for (int nl : {0, nlr}) S = std::max(S, t128_split(nwg, nl + al * kb, nl, slots, forced));
  return S > 1 ? nwg * S : 0;
}

static F16x2Layout make_layout(int64_t M, int64_t K, int64_t r, int64_t N = 0, int al = 0) {
  F16x2Layout L;
  L.Mp = pad_to(M, GM); L.Kp = pad_to(K, GK); L.Rp = r > 0 ? pad_to(r, GK) : 0;
  size_t o = 0;
  L.off_qx = o; o += pad_to((size_t)L.Mp * L.Kp * 2, 256);
  L.off_xl = o; o += pad_to((size_t)L.Mp * L.Kp * 2, 256);      // lo limb of FQ(x), SPQ_PATH_F16X3 only
  L.off_thi = o; o += pad_to((size_t)L.Mp * L.Rp * 2, 256);
  L.off_tlo = o; o += pad_to((size_t)L.Mp * L.Rp * 2, 256);
  L.off_rowinv = o; o += pad_to((size_t)L.Mp * 4, 256);
  // split-K scratch: {tickets, done} per tile, then one 64-KB register image per unit (N = 0: a layout without it)
  const int64_t sku = N > 0 ? sk_reserve_units(M, K, N, r, al) : 0;
  L.off_skcnt = o; o += sku ? 8192 : 0;
  L.off_skpart = o; o += (size_t)sku * 65536;
  L.total = o + 256;
  return L;
}

size_t fwd_f16x2_workspace_bytes(int64_t M, int64_t K, int64_t N, int64_t r, int path) {
  return make_layout(M, K, r, N, path == SPQ_PATH_F16X3 ? 2 : path == SPQ_PATH_I8 ? -1 : 1).total;
}

// power of two p with  v_max * p  in [2^13, 2^14)   (p = 1 for v_max == 0 or non-finite)
__device__ __forceinline__ float pow2_scale_for(float vmax) {
  if (!(vmax > 0.f) || !(vmax < INFINITY)) return 1.f;
  int ex;
  (void)frexpf(vmax, &ex);            // vmax = f * 2^ex, f in [0.5, 1)
  return ldexpf(1.f, 14 - ex);
}

// =================================================================================================
// Weight-side operand preparation.  One workgroup per output row n (grid = Np rows; rows >= N are zero).
//   Whi/Wlo [Np, Kp]  <- FQ(W)[n,:] * sx[:]  * 2^e[n]
//   Bhi/Blo [Np, Rp]  <- scaling * FQ(B)[:, n] * 2^e[n]
//   rowscale[n] = 2^-e[n]
// =================================================================================================
struct PrepArgs {
  const float* W; const float* sw; const float* zw;   // [N,K], weight quantizer params ([N] or [1])
  const float* B; const float* sb; const float* zb;   // [r,N], LoRA-B quantizer params ([N] or [1]); B may be null
  const float* sx;                                    // input scale [K] or [1]
  _Float16 *Whi, *Wlo, *Bhi, *Blo;
  float* rowscale;
  int N, K, r, Kp, Rp;
  int w_pc, w_bits, w_qtype, w_sym;
  int b_pc, b_bits, b_qtype, b_sym;
  int x_pc;
  float scaling;
  // fused job: aT[c, k] = FQ(A)[k, c] for A [K, r] (32x32 tiles in the workgroups after the row workgroups)
  const float* A; const float* sa; const float* za; float* aT;
  int a_pc, a_bits, a_qtype, a_sym;
  int row_blocks;
  // SPQ_PATH_I8 (nl = 1, else 0): the int8 plane of weight levels [Np, Kp]; the LoRA-B limbs then carry a row exponent of
  // their own, bscale[n] = 2^-eb[n]
  signed char* W8; int64_t plane_stride; float* bscale; int nl;
};

template <int QT, bool SYM>
__device__ __forceinline__ float fq_value(float v, float s, float z, int bits, bool log_direct = false, const float* qn_lut = nullptr) {
  if (QT == SPQ_MINMAX) {
    float qlo, qhi;
    if (SYM) { qhi = (float)((1 << (bits - 1)) - 1); qlo = -qhi; }
    else { qlo = 0.f; qhi = (float)((1u << bits) - 1u); }
    return minmax_dequant<SYM>(minmax_level<SYM>(v, s, z, qlo, qhi), s, z);
  } else {
    const LogParams lp = make_log_params(bits, SYM, log_direct);
    return log_dequant<SYM>(v, log_level<SYM>(v, z, s, lp), z, s, lp, qn_lut);
  }
}

__device__ __forceinline__ float fq_dispatch(float v, float s, float z, int bits, int qtype, int sym, const float* qn_lut = nullptr) {
  if (bits >= 32) return v;
  if (qtype == SPQ_MINMAX) return sym ? fq_value<SPQ_MINMAX, true>(v, s, z, bits) : fq_value<SPQ_MINMAX, false>(v, s, z, bits);
  const bool direct = qtype == SPQ_LOG_DIRECT;
  return sym ? fq_value<SPQ_LOG, true>(v, s, z, bits, direct, qn_lut) : fq_value<SPQ_LOG, false>(v, s, z, bits, direct, qn_lut);
}

// log quantizers of at most 8 bits: log_qn of every level, tabulated once per workgroup (256 floats of LDS); returns the table or
// null.  Every thread of the workgroup must call it (barrier inside).
__device__ __forceinline__ const float* fill_log_qn_lut(float* lut, int bits, int qtype, int sym) {
  const bool use = qtype != SPQ_MINMAX && bits >= 1 && bits <= 8;
  if (use) {
    const LogParams lp = make_log_params(bits, sym != 0, qtype == SPQ_LOG_DIRECT);
    const int n = (int)(lp.qhi - lp.qlo) + 1;
    for (int i = threadIdx.x; i < n; i += blockDim.x) lut[i] = sym ? log_qn<true>(lp.qlo + (float)i, lp) : log_qn<false>(lp.qlo + (float)i, lp);
  }
  __syncthreads();
  return use ? lut : nullptr;
}

// nn.GELU() (models_sp.py:107): x * 0.5 * (1 + erf(x / sqrt(2))), the operation order of ATen's CPU kernel
__device__ __forceinline__ float gelu_erf(float x) { return (x * 0.5f) * (1.0f + erff(x * 0.70710678118654752440f)); }

// v = hi + lo up to 2^-22 |v|: hi = RN_f16(v), lo = RN_f16(v - hi); the residual is exact in fp32 (hi carries 11 of
// v's 24 significant bits).  All in fp32: gfx950 has no f64->f16 conversion (clang expands one to ~30 instructions).
__device__ __forceinline__ void split2(float v, _Float16& hi, _Float16& lo) {
  hi = (_Float16)v;
  lo = (_Float16)(v - (float)hi);
}

// aT[c, k] = FQ(A)[k, c]: one 32x32 tile per workgroup of 256 threads (lora.py:49: FQ(A) is recomputed every forward)
__device__ __forceinline__ void fq_transpose_tile(const PrepArgs& a, int tile) {
  __shared__ float tl[32][33];
  const int tiles_c = (a.r + 31) / 32;
  const int tc = tile % tiles_c, tk = tile / tiles_c;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int c = tc * 32 + tx;
  for (int j = ty; j < 32; j += 8) {
    const int k = tk * 32 + j;
    float o = 0.f;
    if (k < a.K && c < a.r) {
      const int ch = a.a_pc ? c : 0;
      o = fq_dispatch(a.A[(int64_t)k * a.r + c], a.sa[ch], a.za[ch], a.a_bits, a.a_qtype, a.a_sym);
    }
    tl[j][tx] = o;
  }
  __syncthreads();
  const int k = tk * 32 + tx;
  for (int j = ty; j < 32; j += 8) {
    const int cc = tc * 32 + j;
    if (k < a.K && cc < a.r) a.aT[(int64_t)cc * a.K + k] = tl[tx][j];
  }
}

__global__ __launch_bounds__(256) void prep_f16x2_kernel(PrepArgs a) {
  if ((int)blockIdx.x >= a.row_blocks) { fq_transpose_tile(a, blockIdx.x - a.row_blocks); return; }
  const int n = blockIdx.x;
  const int tid = threadIdx.x;
  __shared__ float s_red[4];
  __shared__ float s_scale;
  _Float16* whi = a.Whi + (int64_t)n * a.Kp;
  _Float16* wlo = a.Wlo + (int64_t)n * a.Kp;
  _Float16* bhi = a.Bhi ? a.Bhi + (int64_t)n * a.Rp : nullptr;
  _Float16* blo = a.Blo ? a.Blo + (int64_t)n * a.Rp : nullptr;
  if (n >= a.N) {                       // padding rows: zeros
    for (int k = tid; k < a.Kp; k += 256) { whi[k] = (_Float16)0.f; wlo[k] = (_Float16)0.f; }
    if (bhi) for (int j = tid; j < a.Rp; j += 256) { bhi[j] = (_Float16)0.f; blo[j] = (_Float16)0.f; }
    if (tid == 0) a.rowscale[n] = 1.f;
    return;
  }
  const float swn = a.sw[a.w_pc ? n : 0], zwn = a.zw[a.w_pc ? n : 0];
  // pass 1: row maximum of |W'| and |B'|
  float vmax = 0.f;
  for (int k = tid; k < a.K; k += 256) {
    const float wq = fq_dispatch(a.W[(int64_t)n * a.K + k], swn, zwn, a.w_bits, a.w_qtype, a.w_sym);
    vmax = fmaxf(vmax, fabsf(wq * a.sx[a.x_pc ? k : 0]));
  }
  if (a.B) {
    const float sbn = a.sb[a.b_pc ? n : 0], zbn = a.zb[a.b_pc ? n : 0];
    for (int j = tid; j < a.r; j += 256) {
      const float bq = fq_dispatch(a.B[(int64_t)j * a.N + n], sbn, zbn, a.b_bits, a.b_qtype, a.b_sym);
      vmax = fmaxf(vmax, fabsf(bq * a.scaling));
    }
  }
  vmax = wave_max(vmax);
  if ((tid & 63) == 0) s_red[tid >> 6] = vmax;
  __syncthreads();
  if (tid == 0) {
    const float m = fmaxf(fmaxf(s_red[0], s_red[1]), fmaxf(s_red[2], s_red[3]));
    const float p = pow2_scale_for(m);
    s_scale = p;
    a.rowscale[n] = 1.0f / p;                          // exact: p is a power of two
  }
  __syncthreads();
  const float p = s_scale;
  // pass 2: limbs (recomputed rather than staged: the row is tiny and L2-resident)
  for (int k = tid; k < a.Kp; k += 256) {
    _Float16 h = (_Float16)0.f, l = (_Float16)0.f;
    if (k < a.K) {
      const float wq = fq_dispatch(a.W[(int64_t)n * a.K + k], swn, zwn, a.w_bits, a.w_qtype, a.w_sym);
      split2((wq * a.sx[a.x_pc ? k : 0]) * p, h, l);            // W' = fl32(FQ(W) * sx), exactly scaled by 2^e
    }
    whi[k] = h; wlo[k] = l;
  }
  if (bhi) {
    const float sbn = a.B ? a.sb[a.b_pc ? n : 0] : 1.f, zbn = a.B ? a.zb[a.b_pc ? n : 0] : 0.f;
    for (int j = tid; j < a.Rp; j += 256) {
      _Float16 h = (_Float16)0.f, l = (_Float16)0.f;
      if (a.B && j < a.r) {
        const float bq = fq_dispatch(a.B[(int64_t)j * a.N + n], sbn, zbn, a.b_bits, a.b_qtype, a.b_sym);
        split2((bq * a.scaling) * p, h, l);            // (t@Bq)*scaling == t@(Bq*scaling) up to one fp32 rounding
      }
      bhi[j] = h; blo[j] = l;
    }
  }
}

// =================================================================================================
// Activation pass.  Block = 256 threads, 32 rows of x.
//   qx[m, k]  = (fp16) clamp(rint(x/sx), -n, n)                          (quantization_methods.py:14-15)
//   t[m, j]   = sum_k x[m,k] * FQ(A)[k,j]      fp32-input MFMA, K split over the 4 waves, summed in LDS
//   thi/tlo   = two fp16 limbs of t * 2^g[m];  rowinv[m] = 2^-g[m]
// =================================================================================================
struct XPassArgs {
  int* zero_ptr; int zero_n;                // the streaming kernels' block 0 zeroes these ints (the contraction's split-K counters)
  const float* x; const float* sx; const float* zx; const float* aT;   // x [M,K]; aT = FQ(A)^T [r,K] fp32
  _Float16 *qx, *thi, *tlo; float* rowinv;
  int M, K, r, Kp, Rp;
  int x_pc, bits;
  int a8;                                   // 1: levels are written as bytes q + 128 (bits <= 8), 2: as int8 q, else as fp16
  // limbs != 0 (SPQ_PATH_F16X3): any input quantizer; FQ(x) * xscale[0] is written as two fp16 limbs (qx = hi, xl = lo)
  int limbs, qtype, symmetric;
  _Float16* xl;
  const float* xscale;                      // device {2^G, 2^-G}: power of two that puts the quantizer's range bound at 2^14
  float* t_out;                             // optional fp32 [M, r]: the LoRA-down product itself
  int lora_fq;                              // 1: the LoRA-down product consumes FQ(x) (part2 CPTLinear), 0: raw x (part1)
  // optional LayerNorm prologue: x is the INPUT of the SwitchableLayerNorm in front of the layer
  // (switchable_batchnorm.py:102-109); the pass normalises each row on the fly -- weight * ((x - mean) / sqrt(var + eps)) + bias,
  // bit for bit what spq_layernorm writes -- so the normalised fp32 activation is never stored or re-read.  K <= 1024.
  const float* ln_w; const float* ln_b; float ln_eps;
};

// LayerNorm prologue of the panel kernels.  ln_panel_stats: wave w takes rows 4w .. 4w+3 of the workgroup's row block (row in
// registers, two-pass statistics: ln_row_stats) and leaves {mean, den} per row in `st`.  ln_panel_apply: every thread normalises
// its own 16-byte slot of each landed chunk in place (the slot the level pass reads), before anything else reads the panel.
__device__ __forceinline__ void ln_panel_stats(const XPassArgs& a, int m0, int rows, float* st) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  float4 v[4][4];                                           // the wave's four rows: all 16 loads in flight together
#pragma unroll
  for (int i = 0; i < 4; ++i) ln_row_load<4>(a.x + (int64_t)min(m0 + min(4 * w + i, rows - 1), a.M - 1) * a.K, a.K, lane, v[i]);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = 4 * w + i;
    float mean, den;
    ln_row_reduce<4>(v[i], a.K, a.ln_eps, lane, mean, den);
    if (row < rows && lane == 0) { st[2 * row] = mean; st[2 * row + 1] = den; }
  }
}
__device__ __forceinline__ void ln_panel_apply(const XPassArgs& a, char* xs, int chunk_stride, int nch, int p0, int q_row, int q_pos,
                                               int q_kof, const float* st) {
  const float mean = st[2 * q_row], den = st[2 * q_row + 1];
  for (int c = 0; c < nch; ++c) {
    float4* slot = reinterpret_cast<float4*>(xs + c * chunk_stride + q_row * 256 + q_pos * 16);
    const int k = p0 + c * 64 + q_kof;
    const float4 wv = *reinterpret_cast<const float4*>(a.ln_w + k), bv = *reinterpret_cast<const float4*>(a.ln_b + k);
    float4 v = *slot;
    v.x = ln_apply(v.x, mean, den, wv.x, bv.x); v.y = ln_apply(v.y, mean, den, wv.y, bv.y);
    v.z = ln_apply(v.z, mean, den, wv.z, bv.z); v.w = ln_apply(v.w, mean, den, wv.w, bv.w);
    *slot = v;
  }
}

// One WAVE per output row (4 rows per workgroup): the row's FQ(W) values stay in registers between the max pass and
// the limb pass, reductions are 64-lane shuffles, loads are 16 B per lane and stores 8 B per lane.  Needs K % 4 == 0,
// K <= 4096, 16-B aligned W / sx.
constexpr int PREP_MAXI = 16;
// SPQ_PATH_I8: one output row of the int8 weight operand -- the weight's own integer levels (symmetric minmax, <= 8 bit) --
// by ONE wave; rowscale[n] = sw[n] * sx (per-tensor input scale); the LoRA-B limbs get a row exponent of their own (bscale).
template <int MAXI>
__device__ __forceinline__ void prep_row_wave_i8(const PrepArgs& a, int n, int lane, const float* sb_row) {
  signed char* w8 = a.W8 + (int64_t)n * a.Kp;
  _Float16* bhi = a.Bhi ? a.Bhi + (int64_t)n * a.Rp : nullptr;
  _Float16* blo = a.Blo ? a.Blo + (int64_t)n * a.Rp : nullptr;
  const int kp4 = a.Kp >> 2, k4n = a.K >> 2;
  if (n >= a.N) {                       // padding rows: zeros
    for (int k4 = lane; k4 < kp4; k4 += 64) *reinterpret_cast<unsigned*>(w8 + 4 * k4) = 0u;
    if (bhi) for (int j = lane; j < a.Rp; j += 64) { bhi[j] = (_Float16)0.f; blo[j] = (_Float16)0.f; }
    if (lane == 0) { a.rowscale[n] = 1.f; if (a.bscale) a.bscale[n] = 1.f; }
    return;
  }
  const float swn = a.sw[a.w_pc ? n : 0];
  const float4* Wrow = reinterpret_cast<const float4*>(a.W + (int64_t)n * a.K);
  const float qn = (float)((1 << (a.w_bits - 1)) - 1);
#pragma unroll
  for (int i = 0; i < MAXI; ++i) {
    const int k4 = lane + 64 * i;
    if (k4 < kp4) {
      unsigned d = 0u;
      if (k4 < k4n) {
        const float4 v = Wrow[k4];      // integer levels (quantization_methods.py:14-15)
        const int l0 = (int)minmax_level<true>(v.x, swn, 0.f, -qn, qn), l1 = (int)minmax_level<true>(v.y, swn, 0.f, -qn, qn);
        const int l2 = (int)minmax_level<true>(v.z, swn, 0.f, -qn, qn), l3 = (int)minmax_level<true>(v.w, swn, 0.f, -qn, qn);
        d = (unsigned)(l0 & 255) | ((unsigned)(l1 & 255) << 8) | ((unsigned)(l2 & 255) << 16) | ((unsigned)(l3 & 255) << 24);
      }
      *reinterpret_cast<unsigned*>(w8 + 4 * k4) = d;
    }
  }
  if (lane == 0) a.rowscale[n] = swn * a.sx[0];          // the product of the two scales, one fp32 rounding
  if (bhi) {
    float bq[2] = {0.f, 0.f};
    float bmax = 0.f;
    if (a.B) {
      const float sbn = a.sb[a.b_pc ? n : 0], zbn = a.zb[a.b_pc ? n : 0];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int j = lane + 64 * i;
        if (j < a.r) {
          const float braw = sb_row ? sb_row[j] : a.B[(int64_t)j * a.N + n];
          bq[i] = fq_dispatch(braw, sbn, zbn, a.b_bits, a.b_qtype, a.b_sym) * a.scaling;
          bmax = fmaxf(bmax, fabsf(bq[i]));
        }
      }
    }
    bmax = wave_max(bmax);
    const float pb = pow2_scale_for(bmax);
    if (lane == 0) a.bscale[n] = 1.0f / pb;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int j = lane + 64 * i;
      if (j < a.Rp) {
        _Float16 hh = (_Float16)0.f, ll = (_Float16)0.f;
        if (a.B && j < a.r) split2(bq[i] * pb, hh, ll);
        bhi[j] = hh; blo[j] = ll;
      }
    }
  }
}

// one output row n of the weight-side operands, by ONE wave (lane = threadIdx.x & 63).  `sb_row` (nullable): the row's raw
// LoRA-B column B[0..r-1][n] staged contiguously by the caller (LDS) instead of the strided global read.
// MODE 0: two fp16 limbs of W' * 2^e[n] (F16X2 / F16X3).  MODE 1: the weight's own integer levels as int8 (SPQ_PATH_I8).
template <int MODE, int MAXI = PREP_MAXI>    // MAXI: float4 loads per lane that cover a row (K <= 256 * MAXI)
__device__ __forceinline__ void prep_row_wave(const PrepArgs& a, int n, int lane, const float* sb_row, float* stage_dst0 = nullptr,
                                              float* stage_dst1 = nullptr, float stage_v0 = 0.f, float stage_v1 = 0.f) {
  // stage_dst*: the workgroup's LoRA-B columns were fetched into registers BEFORE this row's loads were issued (one memory round
  // trip for both); they go to LDS -- and the workgroup meets -- only where the row first needs them.  Every wave of the
  // workgroup must then come through here (no padding rows in a staged workgroup).
  auto stage_b = [&]() {
    if (stage_dst0) *stage_dst0 = stage_v0;
    if (stage_dst1) *stage_dst1 = stage_v1;
    __syncthreads();
  };
  if constexpr (MODE != 0) { if (stage_dst0 || stage_dst1) stage_b(); prep_row_wave_i8<MAXI>(a, n, lane, sb_row); return; }
  _Float16* whi = a.Whi + (int64_t)n * a.Kp;
  _Float16* wlo = a.Wlo + (int64_t)n * a.Kp;
  _Float16* bhi = a.Bhi ? a.Bhi + (int64_t)n * a.Rp : nullptr;
  _Float16* blo = a.Blo ? a.Blo + (int64_t)n * a.Rp : nullptr;
  const int kp4 = a.Kp >> 2, k4n = a.K >> 2;
  if (n >= a.N) {                       // padding rows: zeros
    for (int k4 = lane; k4 < kp4; k4 += 64) {
      *reinterpret_cast<uint2*>(whi + 4 * k4) = make_uint2(0, 0);
      *reinterpret_cast<uint2*>(wlo + 4 * k4) = make_uint2(0, 0);
    }
    if (bhi) for (int j = lane; j < a.Rp; j += 64) { bhi[j] = (_Float16)0.f; blo[j] = (_Float16)0.f; }
    if (lane == 0) a.rowscale[n] = 1.f;
    return;
  }
  const float swn = a.sw[a.w_pc ? n : 0], zwn = a.zw[a.w_pc ? n : 0];
  const float4* Wrow = reinterpret_cast<const float4*>(a.W + (int64_t)n * a.K);
  // every load of the row first (W and the input scales), then the arithmetic: written as one loop the compiler waited for each
  // 16-byte piece and then for its scales before asking for the next -- six dependent memory round trips per 768-element row --
  // and the quantizer dispatch (min-max / log, the fp64 log2 of the latter) sat inside it, 33 000 instructions of ISA
  float4 wq[MAXI], sc[MAXI];
  // (branch-free inside the loops: a guarded load made the compiler wait for it at the join)
  if (a.x_pc) {
#pragma unroll
    for (int i = 0; i < MAXI; ++i) {
      const int kc = min(lane + 64 * i, k4n - 1);
      wq[i] = Wrow[kc];
      sc[i] = *reinterpret_cast<const float4*>(a.sx + 4 * kc);
    }
  } else {
    const float s1 = a.sx[0];
#pragma unroll
    for (int i = 0; i < MAXI; ++i) {
      wq[i] = Wrow[min(lane + 64 * i, k4n - 1)];
      sc[i] = make_float4(s1, s1, s1, s1);
    }
  }
  float vmax = 0.f;
  auto fold = [&](auto fq) {                                // W' = fl32(FQ(W) * sx), row maximum
#pragma unroll
    for (int i = 0; i < MAXI; ++i) {
      const int k4 = lane + 64 * i;
      if (k4 < k4n) {
        float4 q;
        q.x = fq(wq[i].x) * sc[i].x; q.y = fq(wq[i].y) * sc[i].y; q.z = fq(wq[i].z) * sc[i].z; q.w = fq(wq[i].w) * sc[i].w;
        wq[i] = q;
        vmax = fmaxf(vmax, fmaxf(fmaxf(fabsf(q.x), fabsf(q.y)), fmaxf(fabsf(q.z), fabsf(q.w))));
      }
    }
  };
  if (a.w_bits >= 32) fold([](float v) { return v; });
  else if (a.w_qtype == SPQ_MINMAX && a.w_sym) fold([&](float v) { return fq_value<SPQ_MINMAX, true>(v, swn, zwn, a.w_bits); });
  else if (a.w_qtype == SPQ_MINMAX) fold([&](float v) { return fq_value<SPQ_MINMAX, false>(v, swn, zwn, a.w_bits); });
  else fold([&](float v) { return fq_dispatch(v, swn, zwn, a.w_bits, a.w_qtype, a.w_sym); });
  float bq[2] = {0.f, 0.f};
  if (a.B) {
    const float sbn = a.sb[a.b_pc ? n : 0], zbn = a.zb[a.b_pc ? n : 0];
    if (stage_dst0 || stage_dst1) stage_b();
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int j = lane + 64 * i;
      if (j < a.r) {
        const float braw = sb_row ? sb_row[j] : a.B[(int64_t)j * a.N + n];
        bq[i] = fq_dispatch(braw, sbn, zbn, a.b_bits, a.b_qtype, a.b_sym) * a.scaling;
        vmax = fmaxf(vmax, fabsf(bq[i]));
      }
    }
  }
  vmax = wave_max(vmax);
  const float p = pow2_scale_for(vmax);
  if (lane == 0) a.rowscale[n] = 1.0f / p;               // exact: p is a power of two
#pragma unroll
  for (int i = 0; i < MAXI; ++i) {
    const int k4 = lane + 64 * i;
    if (k4 < kp4) {
      union { _Float16 h[4]; uint2 u; } hi, lo;
      hi.u = make_uint2(0, 0); lo.u = make_uint2(0, 0);
      if (k4 < k4n) {
        split2(wq[i].x * p, hi.h[0], lo.h[0]);
        split2(wq[i].y * p, hi.h[1], lo.h[1]);
        split2(wq[i].z * p, hi.h[2], lo.h[2]);
        split2(wq[i].w * p, hi.h[3], lo.h[3]);
      }
      *reinterpret_cast<uint2*>(whi + 4 * k4) = hi.u;
      *reinterpret_cast<uint2*>(wlo + 4 * k4) = lo.u;
    }
  }
  if (bhi) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int j = lane + 64 * i;
      if (j < a.Rp) {
        _Float16 hh = (_Float16)0.f, ll = (_Float16)0.f;
        if (a.B && j < a.r) split2(bq[i] * p, hh, ll);
        bhi[j] = hh; blo[j] = ll;
      }
    }
  }
}

template <int MODE, int MAXI = PREP_MAXI>   // MAXI = 4: rows of at most 1024 elements
__global__ __launch_bounds__(256) void prep_wave_kernel(PrepArgs a, int at_blocks) {
  // the FQ(A)^T tiles come FIRST in the grid (they are the longer workgroups: a tail of them cost ~2 us)
  if ((int)blockIdx.x < at_blocks) { fq_transpose_tile(a, blockIdx.x); return; }
  const int n0 = ((int)blockIdx.x - at_blocks) * 4;
  // the four rows' LoRA-B columns B[j][n0 .. n0+3] are 16-byte runs: one coalesced-per-j load by the workgroup, staged in LDS,
  // instead of a strided scalar read per (row, j) from every wave
  __shared__ float sB[4][128];
  const bool staged = a.B && n0 + 3 < a.N && (a.N & 3) == 0;      // (then r <= 128: at most two elements per thread)
  float* d0 = nullptr; float* d1 = nullptr;
  float v0 = 0.f, v1 = 0.f;
  if (staged) {
    const int e0 = threadIdx.x, e1 = threadIdx.x + 256;
    d0 = &sB[e0 & 3][e0 >> 2];                             // every thread takes part in the barrier, with or without an element
    if (e0 < a.r * 4) v0 = a.B[(int64_t)(e0 >> 2) * a.N + n0 + (e0 & 3)];
    if (e1 < a.r * 4) { d1 = &sB[e1 & 3][e1 >> 2]; v1 = a.B[(int64_t)(e1 >> 2) * a.N + n0 + (e1 & 3)]; }
  }
  const int wv = threadIdx.x >> 6;
  prep_row_wave<MODE, MAXI>(a, n0 + wv, threadIdx.x & 63, staged ? &sB[wv][0] : nullptr, d0, d1, v0, v1);
}

// FQ(A)^T alone (the first of the two operand launches when the row work rides in the activation pass)
__global__ __launch_bounds__(256) void fq_transpose_kernel(PrepArgs a) { fq_transpose_tile(a, blockIdx.x); }

// Shared tail of the activation pass: sum the NW per-wave K-partials of t (fixed order), per-row power-of-two scale,
// two fp16 limbs.  `red` must hold NW * XR * RP floats and be free of other use (caller synchronised).  All threads
// of the workgroup enter; threads >= 256 only contribute their partial.
template <int RT, int NW>
__device__ __forceinline__ void xpass_finish(const XPassArgs& a, const f32x16 (&acc)[RT], float* red, int m0, int tid) {
  constexpr int RP = RT * 32;
  const int lane = tid & 63, w = tid >> 6;
  {
    const int l31 = lane & 31, h = lane >> 5;
#pragma unroll
    for (int t = 0; t < RT; ++t)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = (e & 3) + 8 * (e >> 2) + 4 * h;
        red[(w * XR + row) * RP + t * 32 + l31] = acc[t][e];
      }
  }
  __syncthreads();
  if (tid >= 256) return;
  // thread -> (row = tid>>3, 8-column segment(s)); 8 consecutive lanes share a row
  const int row = tid >> 3, seg = tid & 7;
  constexpr int NSEG = RP / 64;            // 8-col segments per thread
  float tv[NSEG][8];
  float rmax = 0.f;
#pragma unroll
  for (int sgi = 0; sgi < NSEG; ++sgi) {
    const int c0 = (sgi * 8 + seg) * 8;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float part[NW];
#pragma unroll
      for (int q = 0; q < NW; ++q) part[q] = red[(q * XR + row) * RP + c0 + j];
#pragma unroll
      for (int st = 1; st < NW; st *= 2)                  // fixed pairwise tree: ((p0+p1)+(p2+p3))+...
#pragma unroll
        for (int q = 0; q + st < NW; q += 2 * st) part[q] = part[q] + part[q + st];
      tv[sgi][j] = part[0];
      rmax = fmaxf(rmax, fabsf(part[0]));
    }
  }
  rmax = fmaxf(rmax, __shfl_xor(rmax, 1, 64));
  rmax = fmaxf(rmax, __shfl_xor(rmax, 2, 64));
  rmax = fmaxf(rmax, __shfl_xor(rmax, 4, 64));
  const float p = pow2_scale_for(rmax);
  const int m = m0 + row;
  if (m < a.M) {
    if (seg == 0) a.rowinv[m] = 1.0f / p;
#pragma unroll
    for (int sgi = 0; sgi < NSEG; ++sgi) {
      const int c0 = (sgi * 8 + seg) * 8;
      union { _Float16 h[8]; uint4 u; } hi, lo;
#pragma unroll
      for (int j = 0; j < 8; ++j) split2(tv[sgi][j] * p, hi.h[j], lo.h[j]);   // * p exact (power of two)
      *reinterpret_cast<uint4*>(a.thi + (int64_t)m * a.Rp + c0) = hi.u;
      *reinterpret_cast<uint4*>(a.tlo + (int64_t)m * a.Rp + c0) = lo.u;
      if (a.t_out) {
        float* to = a.t_out + (int64_t)m * a.r + c0;
        if ((a.r & 3) == 0 && c0 + 8 <= a.r) {
          *reinterpret_cast<float4*>(to) = make_float4(tv[sgi][0], tv[sgi][1], tv[sgi][2], tv[sgi][3]);
          *reinterpret_cast<float4*>(to + 4) = make_float4(tv[sgi][4], tv[sgi][5], tv[sgi][6], tv[sgi][7]);
        } else {
#pragma unroll
          for (int j = 0; j < 8; ++j) if (c0 + j < a.r) to[j] = tv[sgi][j];
        }
      }
    }
  }
}

// activation operand of four consecutive elements at element index idx: exact integer levels (fp16, or bytes q + 128
// when a8) for the symmetric-minmax path, or the two fp16 limbs of FQ(x) * 2^G for any other input quantizer
__device__ __forceinline__ void store_act4(const XPassArgs& a, int64_t idx, float4 v, float4 sc, float4 zp, float qlo, float qhi,
                                           float pscale, const float* qn_lut = nullptr) {
  if (a.limbs) {
    const float f0 = fq_dispatch(v.x, sc.x, zp.x, a.bits, a.qtype, a.symmetric, qn_lut) * pscale;     // * 2^G exact
    const float f1 = fq_dispatch(v.y, sc.y, zp.y, a.bits, a.qtype, a.symmetric, qn_lut) * pscale;
    const float f2 = fq_dispatch(v.z, sc.z, zp.z, a.bits, a.qtype, a.symmetric, qn_lut) * pscale;
    const float f3 = fq_dispatch(v.w, sc.w, zp.w, a.bits, a.qtype, a.symmetric, qn_lut) * pscale;
    union { _Float16 h[4]; uint2 u; } hi, lo;
    split2(f0, hi.h[0], lo.h[0]); split2(f1, hi.h[1], lo.h[1]); split2(f2, hi.h[2], lo.h[2]); split2(f3, hi.h[3], lo.h[3]);
    *reinterpret_cast<uint2*>(a.qx + idx) = hi.u;
    *reinterpret_cast<uint2*>(a.xl + idx) = lo.u;
    return;
  }
  const float q0 = minmax_level<true>(v.x, sc.x, 0.f, qlo, qhi), q1 = minmax_level<true>(v.y, sc.y, 0.f, qlo, qhi);
  const float q2 = minmax_level<true>(v.z, sc.z, 0.f, qlo, qhi), q3 = minmax_level<true>(v.w, sc.w, 0.f, qlo, qhi);
  if (a.a8) {
    const int b8 = a.a8 == 1 ? 128 : 0;                  // 1: bytes q + 128 (SPQ_PATH_U8X2); 2: int8 q (SPQ_PATH_I8)
    const unsigned u = (unsigned)(((int)q0 + b8) & 255) | ((unsigned)(((int)q1 + b8) & 255) << 8) |
                       ((unsigned)(((int)q2 + b8) & 255) << 16) | ((unsigned)(((int)q3 + b8) & 255) << 24);
    *reinterpret_cast<unsigned*>(reinterpret_cast<unsigned char*>(a.qx) + idx) = u;
  } else {
    union { _Float16 hh[4]; uint2 u; } q;
    q.hh[0] = (_Float16)q0; q.hh[1] = (_Float16)q1; q.hh[2] = (_Float16)q2; q.hh[3] = (_Float16)q3;
    *reinterpret_cast<uint2*>(a.qx + idx) = q.u;
  }
}

// FQ of four consecutive activations (the LoRA-down operand of part2's CPTLinear, cpt_model.py:112)
__device__ __forceinline__ float4 fq_act4(const XPassArgs& a, float4 v, float4 sc, float4 zp, const float* qn_lut = nullptr) {
  return make_float4(fq_dispatch(v.x, sc.x, zp.x, a.bits, a.qtype, a.symmetric, qn_lut), fq_dispatch(v.y, sc.y, zp.y, a.bits, a.qtype, a.symmetric, qn_lut),
                     fq_dispatch(v.z, sc.z, zp.z, a.bits, a.qtype, a.symmetric, qn_lut), fq_dispatch(v.w, sc.w, zp.w, a.bits, a.qtype, a.symmetric, qn_lut));
}

// four consecutive integer levels -> level matrix at element index idx (fp16, or bytes q + 128 when a8)
__device__ __forceinline__ void store_levels4(_Float16* qx, int64_t idx, float q0, float q1, float q2, float q3, int a8) {
  if (a8) {
    const int b8 = a8 == 1 ? 128 : 0;
    const unsigned u = (unsigned)(((int)q0 + b8) & 255) | ((unsigned)(((int)q1 + b8) & 255) << 8) |
                       ((unsigned)(((int)q2 + b8) & 255) << 16) | ((unsigned)(((int)q3 + b8) & 255) << 24);
    *reinterpret_cast<unsigned*>(reinterpret_cast<unsigned char*>(qx) + idx) = u;
  } else {
    union { _Float16 hh[4]; uint2 u; } q;
    q.hh[0] = (_Float16)q0; q.hh[1] = (_Float16)q1; q.hh[2] = (_Float16)q2; q.hh[3] = (_Float16)q3;
    *reinterpret_cast<uint2*>(qx + idx) = q.u;
  }
}

constexpr int XLD = XK + 4;   // fp32 LDS row stride 272 B: slot = 17*row + c (mod 16) -> conflict-free b128 reads

template <int RT>  // RT = Rp / 32 column tiles of t (2 for r<=64, 4 for r<=128)
__global__ __launch_bounds__(256) void xpass_kernel(XPassArgs a) {
  constexpr int RP = RT * 32;
  constexpr int SM_X = XR * XLD, SM_A = RP * XLD;
  constexpr int SM_RED = 4 * XR * RP;
  __shared__ __attribute__((aligned(16))) float smem[(SM_X + SM_A) > SM_RED ? (SM_X + SM_A) : SM_RED];
  float* xs = smem;
  float* as = smem + SM_X;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int m0 = blockIdx.x * XR;
  const float qhi = (float)((1 << (a.bits - 1)) - 1), qlo = -qhi;
  const bool with_lora = a.r > 0;

  f32x16 acc[RT];
#pragma unroll
  for (int i = 0; i < RT; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;

  float4 rx[2];
  float4 ra[RT * 2];
  auto load_chunk = [&](int k0) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {                 // x: 32 rows x 16 float4
      const int idx = tid + 256 * i, row = idx >> 4, c = (idx & 15) << 2;
      const int m = m0 + row, k = k0 + c;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (m < a.M) {
        const float* p = a.x + (int64_t)m * a.K + k;
        if (k + 3 < a.K && ((a.K & 3) == 0)) v = *reinterpret_cast<const float4*>(p);
        else {
          if (k + 0 < a.K) v.x = p[0];
          if (k + 1 < a.K) v.y = p[1];
          if (k + 2 < a.K) v.z = p[2];
          if (k + 3 < a.K) v.w = p[3];
        }
      }
      rx[i] = v;
    }
    if (with_lora) {
#pragma unroll
      for (int i = 0; i < RT * 2; ++i) {          // FQ(A)^T: RP rows x 16 float4
        const int idx = tid + 256 * i, row = idx >> 4, c = (idx & 15) << 2;
        const int k = k0 + c;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (row < a.r) {
          const float* p = a.aT + (int64_t)row * a.K + k;
          if (k + 3 < a.K && ((a.K & 3) == 0)) v = *reinterpret_cast<const float4*>(p);
          else {
            if (k + 0 < a.K) v.x = p[0];
            if (k + 1 < a.K) v.y = p[1];
            if (k + 2 < a.K) v.z = p[2];
            if (k + 3 < a.K) v.w = p[3];
          }
        }
        ra[i] = v;
      }
    }
  };

  load_chunk(0);
  for (int k0 = 0; k0 < a.Kp; k0 += XK) {
    __syncthreads();
    // quantise + stage this chunk
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int idx = tid + 256 * i, row = idx >> 4, c = (idx & 15) << 2;
      const int m = m0 + row, k = k0 + c;
      const float4 v = rx[i];
      float s0, s1, s2, s3;
      if (a.x_pc) {
        s0 = (k + 0 < a.K) ? a.sx[k + 0] : 1.f; s1 = (k + 1 < a.K) ? a.sx[k + 1] : 1.f;
        s2 = (k + 2 < a.K) ? a.sx[k + 2] : 1.f; s3 = (k + 3 < a.K) ? a.sx[k + 3] : 1.f;
      } else { s0 = s1 = s2 = s3 = a.sx[0]; }
      float4 zp4 = make_float4(0.f, 0.f, 0.f, 0.f);
      if (a.limbs || a.lora_fq) {
        if (a.x_pc) {
          zp4.x = (k + 0 < a.K) ? a.zx[k + 0] : 0.f; zp4.y = (k + 1 < a.K) ? a.zx[k + 1] : 0.f;
          zp4.z = (k + 2 < a.K) ? a.zx[k + 2] : 0.f; zp4.w = (k + 3 < a.K) ? a.zx[k + 3] : 0.f;
        } else { const float z1 = a.zx[0]; zp4 = make_float4(z1, z1, z1, z1); }
      }
      {
        float4 lv = v;                                                               // LoRA-down operand: raw x, or FQ(x)
        if (a.lora_fq) {
          lv = fq_act4(a, v, make_float4(s0, s1, s2, s3), zp4);
          if (k + 0 >= a.K) lv.x = 0.f; if (k + 1 >= a.K) lv.y = 0.f; if (k + 2 >= a.K) lv.z = 0.f; if (k + 3 >= a.K) lv.w = 0.f;
        }
        *reinterpret_cast<float4*>(xs + row * XLD + c) = lv;
      }
      if (m < a.M) {                                                                 // k < Kp always; pad k -> level 0
        float4 vv = v;
        if (a.limbs) {                                                               // pad columns must come out as zero limbs
          if (k + 0 >= a.K) vv.x = 0.f; if (k + 1 >= a.K) vv.y = 0.f; if (k + 2 >= a.K) vv.z = 0.f; if (k + 3 >= a.K) vv.w = 0.f;
        }
        store_act4(a, (int64_t)m * a.Kp + k, vv, make_float4(s0, s1, s2, s3), zp4, qlo, qhi, a.limbs ? a.xscale[0] : 1.f);
        if (a.limbs && k + 3 >= a.K) {                                               // FQ(0) need not be 0 (asymmetric / log): force pads
          _Float16* ph = a.qx + (int64_t)m * a.Kp + k; _Float16* pl = a.xl + (int64_t)m * a.Kp + k;
          for (int j = 0; j < 4; ++j) if (k + j >= a.K) { ph[j] = (_Float16)0.f; pl[j] = (_Float16)0.f; }
        }
      }
    }
    if (with_lora) {
#pragma unroll
      for (int i = 0; i < RT * 2; ++i) {
        const int idx = tid + 256 * i, row = idx >> 4, c = (idx & 15) << 2;
        *reinterpret_cast<float4*>(as + row * XLD + c) = ra[i];
      }
    }
    __syncthreads();
    if (k0 + XK < a.Kp) load_chunk(k0 + XK);
    if (with_lora) {
      // wave w owns k in [16w, 16w+16) of the chunk; lane half h owns 8 contiguous k of those
      const int l31 = lane & 31, h = lane >> 5;
      const float* pa = xs + l31 * XLD + 16 * w + 8 * h;
      float av[8];
      *reinterpret_cast<float4*>(av) = *reinterpret_cast<const float4*>(pa);
      *reinterpret_cast<float4*>(av + 4) = *reinterpret_cast<const float4*>(pa + 4);
#pragma unroll
      for (int t = 0; t < RT; ++t) {
        const float* pb = as + (t * 32 + l31) * XLD + 16 * w + 8 * h;
        float bv[8];
        *reinterpret_cast<float4*>(bv) = *reinterpret_cast<const float4*>(pb);
        *reinterpret_cast<float4*>(bv + 4) = *reinterpret_cast<const float4*>(pb + 4);
#pragma unroll
        for (int s = 0; s < 8; ++s) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s], bv[s], acc[t], 0, 0, 0);
      }
    }
  }
  if (!with_lora) return;
  __syncthreads();
  xpass_finish<RT, 4>(a, acc, smem, m0, tid);
}

// -------------------------------------------------------------------------------------------------------------------
// Fast activation pass (K % 64 == 0, r <= 64, 16-B aligned rows): the whole 32-row x panel (up to 768 columns =
// 96 KB) is put in flight at once with direct global->LDS copies, so HBM latency is paid once per workgroup instead of
// once per 64-column chunk (the generic kernel above is latency-bound at one workgroup per CU).  FQ(A)^T streams
// through a double buffer one chunk ahead.  LDS images are [chunk][row][64 floats]; the 16-B position p of row r holds
// source chunk p ^ (r & 15), so the MFMA operand reads (ds_read_b128, row = lane & 31) are conflict-free.
// -------------------------------------------------------------------------------------------------------------------
constexpr int XP_CHUNKS = 12;                              // panel = 12 chunks x 64 columns (one workgroup per CU)
constexpr int XP_CHUNKS_SMALL = 4;                         // 66 KB of LDS: two workgroups per CU, for M large enough to have them
constexpr int XP_AS = 64 * 64 * 4;                         // 16 KB per FQ(A)^T chunk (64 rows of r)
constexpr int XP_NAS = 2;                                  // FQ(A)^T chunk buffers
constexpr int xp_xs(int ch) { return ch * XR * 64 * 4; }   // x panel image: 96 KB / 32 KB
constexpr int xp_sx(int ch) { return 2 * ch * 64 * 4; }    // the panel's input scales and zero points
constexpr int xp_lds(int ch) { return xp_xs(ch) + XP_NAS * XP_AS + xp_sx(ch) + 256; }   // 131 KB / 66 KB (+ LayerNorm row statistics)
constexpr int XP_XS = xp_xs(XP_CHUNKS);
constexpr int XP_LDS = xp_lds(XP_CHUNKS);

// CH = chunks per panel: 12 (whole 768-column rows in flight, one workgroup per CU) while there is at most one 32-row block
// per CU, 4 (two resident workgroups that cover each other's load phases: 276 -> 223 us at 32768 x 3072) beyond that.
template <int CH>
__global__ __launch_bounds__(512) void xpass_panel_kernel(XPassArgs a) {
  extern __shared__ __attribute__((aligned(16))) char xsm[];
  char* xs = xsm;
  char* as = xsm + xp_xs(CH);
  float* sxs = reinterpret_cast<float*>(xsm + xp_xs(CH) + XP_NAS * XP_AS);
  float* lnst = reinterpret_cast<float*>(xsm + xp_xs(CH) + XP_NAS * XP_AS + xp_sx(CH));
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);   // 8 waves: two per SIMD, so one's VALU runs under the other's MFMAs
  const int m0 = blockIdx.x * XR;
  if (a.ln_w) ln_panel_stats(a, m0, XR, lnst);              // visible to every thread after the first panel's barrier
  const float qhi = (float)((1 << (a.bits - 1)) - 1), qlo = -qhi;
  const float pscale = a.limbs ? a.xscale[0] : 1.f;
  const bool with_lora = a.r > 0;
  const int l31 = lane & 31, h = lane >> 5;
  __shared__ float s_qn[256];
  const float* qn_lut = (a.limbs || a.lora_fq) ? fill_log_qn_lut(s_qn, a.bits, a.qtype, a.symmetric) : nullptr;   // (uniform branch)

  f32x16 acc[2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;

  // x copy pieces: 1 KB = 4 rows x 256 B; lane -> (row = lane>>4, position = lane&15); source chunk = position ^ (row&15)
  // a 64-column chunk of the 32-row panel is 8 pieces: wave w issues piece w
  const int prow = w * 4 + (lane >> 4), ppos = lane & 15;
  const float* x_src = a.x + (int64_t)min(m0 + prow, a.M - 1) * a.K + ((ppos ^ (prow & 15)) << 2);   // rows past M re-read row M-1
  // this thread's (row, 16-B position) slot of a chunk for the level pass, fixed for the whole kernel
  const int q_row = tid >> 4, q_pos = tid & 15;
  const int q_kof = (q_pos ^ (q_row & 15)) << 2;
  const int64_t q_dst = (int64_t)min(m0 + q_row, a.M - 1) * a.Kp + q_kof;

  // FQ(A)^T chunk g (global chunk index over all panels) lives in buffer g & 1.  It is staged through registers
  // (global_load at the top of iteration g-1, ds_write at its bottom): hipcc drains every outstanding LDS-DMA with
  // vmcnt(0) in front of an LDS read it cannot prove disjoint, which would expose the copy's latency in every chunk.
  // thread -> 2 x (row = tid>>4 (+32), source chunk = tid&15) of a 64-row x 256-B chunk
  const int total_chunks = a.K / 64;
  float4 ra0, ra1;
  ra0 = ra1 = make_float4(0.f, 0.f, 0.f, 0.f);
  const int a_r = tid >> 4, a_c = tid & 15;
  const float* a_src = a.aT + (int64_t)a_r * a.K + (a_c << 2);
  const int64_t a_step = (int64_t)32 * a.K;
  const int a_dst = a_r * 256 + ((a_c ^ (a_r & 15)) << 4);       // (a_r + 32) & 15 == a_r & 15
#define SPQ_LOAD_A(k0)                                                                 \
  do {                                                                                 \
    ra0 = *reinterpret_cast<const float4*>(a_src + (k0));                              \
    ra1 = *reinterpret_cast<const float4*>(a_src + a_step + (k0));                     \
  } while (0)
#define SPQ_STORE_A(buf)                                                               \
  do {                                                                                 \
    char* d_ = as + (buf) * XP_AS + a_dst;                                             \
    *reinterpret_cast<float4*>(d_) = ra0;                                              \
    *reinterpret_cast<float4*>(d_ + 32 * 256) = ra1;                                   \
  } while (0)

  int gc = 0;
  if (with_lora) SPQ_LOAD_A(0);
  for (int p0 = 0; p0 < a.K; p0 += CH * 64) {
    const int nch = min(CH, (a.K - p0) / 64);
    for (int c = 0; c < nch; ++c) glds16(x_src + p0 + c * 64, xs + c * (XR * 256) + w * 1024);
    for (int k = tid; k < nch * 64; k += 512) {
      sxs[k] = a.x_pc ? a.sx[p0 + k] : a.sx[0];
      sxs[CH * 64 + k] = (a.limbs || a.lora_fq) ? (a.x_pc ? a.zx[p0 + k] : a.zx[0]) : 0.f;
    }
    if (with_lora && p0 == 0) SPQ_STORE_A(0);
    __syncthreads();                                       // vmcnt(0): the panel landed; FQ(A)^T chunk gc is in LDS
    if (a.ln_w) { ln_panel_apply(a, xs, XR * 256, nch, p0, q_row, q_pos, q_kof, lnst); __syncthreads(); }
    for (int c = 0; c < nch; ++c, ++gc) {
      const int k0 = p0 + c * 64;
      const bool next_a = with_lora && gc + 1 < total_chunks;
      if (next_a) SPQ_LOAD_A((gc + 1) * 64);                // lands under this chunk's work
      // ---- integer levels of this chunk: one float4 per thread
      {
        const float4 v = *reinterpret_cast<const float4*>(xs + c * (XR * 256) + q_row * 256 + q_pos * 16);
        const float4 sc = *reinterpret_cast<const float4*>(sxs + c * 64 + q_kof);
        const float4 zp = *reinterpret_cast<const float4*>(sxs + CH * 64 + c * 64 + q_kof);
        store_act4(a, q_dst + k0, v, sc, zp, qlo, qhi, pscale, qn_lut);
      }
      // ---- t += x . FQ(A): wave w owns k in [8w, 8w+8) of the chunk, lane half h the 4 contiguous k 8w+4h..+3
      if (with_lora) {
        const int pa = 2 * w + h;                          // 16-B source chunk of this lane
        float4 av = *reinterpret_cast<const float4*>(xs + c * (XR * 256) + l31 * 256 + ((pa ^ (l31 & 15)) << 4));
        if (a.lora_fq)                                     // every element is read by exactly one lane: FQ it in place
          av = fq_act4(a, av, *reinterpret_cast<const float4*>(sxs + c * 64 + 4 * pa),
                       *reinterpret_cast<const float4*>(sxs + CH * 64 + c * 64 + 4 * pa), qn_lut);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const int rb = t * 32 + l31;
          const float4 bv = *reinterpret_cast<const float4*>(as + (gc & 1) * XP_AS + rb * 256 + ((pa ^ (rb & 15)) << 4));
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bv.x, acc[t], 0, 0, 0);
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bv.y, acc[t], 0, 0, 0);
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, bv.z, acc[t], 0, 0, 0);
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, bv.w, acc[t], 0, 0, 0);
        }
        if (next_a) SPQ_STORE_A((gc + 1) & 1);             // the other buffer: every wave left it at the last barrier
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // my LDS writes are done (the level stores may stay in flight)
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
    }
    __syncthreads();                                       // panel images are free for the next panel
  }
#undef SPQ_LOAD_A
#undef SPQ_STORE_A
  if (!with_lora) return;
  xpass_finish<2, 8>(a, acc, reinterpret_cast<float*>(xsm), m0, tid);
}

constexpr int XR16 = 16;
// finish of the streaming kernels (16 / 32 rows per workgroup): sum the 4 per-wave k-partials (fixed order), per-row power-of-two scale, two fp16 limbs (+ t_out).
// red: [4][ROWS][64] floats (16 / 32 KB) of LDS that nothing else is using.
template <int ROWS = XR16>          // ROWS = 32: waves 4..7 hold rows 16..31 (wave = 4 * row group + k-slice)
__device__ __forceinline__ void xpass16_finish(const XPassArgs& a, const f32x4 (&acc)[4], float* red, int m0, int tid) {
  const int lane = tid & 63, w = (tid >> 6) & 3, rg = tid >> 8, l15 = lane & 15, q4 = lane >> 4;
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int e = 0; e < 4; ++e) red[(w * ROWS + rg * 16 + 4 * q4 + e) * 64 + t * 16 + l15] = acc[t][e];   // C/D: col = lane&15, row = 4 (lane>>4) + e
  __syncthreads();
  const int row = tid >> 4, c0 = (tid & 15) * 4;
  float tv[4];
  float rmax = 0.f;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float s01 = red[(0 * ROWS + row) * 64 + c0 + j] + red[(1 * ROWS + row) * 64 + c0 + j];
    const float s23 = red[(2 * ROWS + row) * 64 + c0 + j] + red[(3 * ROWS + row) * 64 + c0 + j];
    tv[j] = s01 + s23;
    rmax = fmaxf(rmax, fabsf(tv[j]));
  }
  rmax = fmaxf(rmax, __shfl_xor(rmax, 1, 64)); rmax = fmaxf(rmax, __shfl_xor(rmax, 2, 64));
  rmax = fmaxf(rmax, __shfl_xor(rmax, 4, 64)); rmax = fmaxf(rmax, __shfl_xor(rmax, 8, 64));
  const float p = pow2_scale_for(rmax);
  const int m = m0 + row;
  if (m < a.M) {
    if ((tid & 15) == 0) a.rowinv[m] = 1.0f / p;
    union { _Float16 hh[4]; uint2 u; } hi, lo;
#pragma unroll
    for (int j = 0; j < 4; ++j) split2(tv[j] * p, hi.hh[j], lo.hh[j]);
    *reinterpret_cast<uint2*>(a.thi + (int64_t)m * a.Rp + c0) = hi.u;
    *reinterpret_cast<uint2*>(a.tlo + (int64_t)m * a.Rp + c0) = lo.u;
    if (a.t_out) {
      float* to = a.t_out + (int64_t)m * a.r + c0;
      if ((a.r & 3) == 0 && c0 + 4 <= a.r) *reinterpret_cast<float4*>(to) = make_float4(tv[0], tv[1], tv[2], tv[3]);
      else {
#pragma unroll
        for (int j = 0; j < 4; ++j) if (c0 + j < a.r) to[j] = tv[j];
      }
    }
  }
}

#ifndef SPQ_XP_DIAG    // tools/xpass_probe.py only (the library builds 0): 1 = no FQ(A)^T loads / LDS stores after the first chunk,
#define SPQ_XP_DIAG 0  // 2 = no MFMAs, 4 = no level stores, 8 = no x copies after the first chunk, 16 = no per-chunk barrier
#endif
// -------------------------------------------------------------------------------------------------------------------
// Streaming form of the pass for the common case: symmetric min-max levels (fp16 / bytes), LoRA-down on the raw rows, no weight
// rows.  The panel kernels above alternate "copy a panel" and "work on it", and their parts add up (measured with SPQ_XP_DIAG at
// 8192 x 768: copies 4.4 + FQ(A)^T 3.5 + fp32 MFMAs 5 + level pass 9 of 21 us, nothing overlaps).  Here every 64-column chunk --
// the ROWS x 64 piece of x, the 64 x 64 piece of FQ(A)^T (16 KB) and the chunk's 64 scales (+ LayerNorm weight / bias) -- is an
// LDS-DMA stage of a three-slot ring, issued two chunks ahead: one counted s_waitcnt + one barrier per chunk, no register
// staging, no LDS writes in the loop, and the level pass runs in the shadow of the chunk's MFMAs.
//   ROWS = 32 (512 threads: waves 0..3 rows 0..15, waves 4..7 rows 16..31, both halves read the same FQ(A)^T stage) when that
//   still gives every CU a workgroup: FQ(A)^T is re-read once per workgroup through L2 (M/ROWS x 4 r K bytes: 100 MB at
//   8192 x 768 with 16 rows against 25 MB of x) and that traffic is what bounds the 16-row form.
//   vmcnt accounting (gfx9: one in-order counter for loads AND stores): a wave issues CP = 6 (16 rows) or 4 (32 rows) copies
//   per chunk and one level store.  At the top of chunk c the operations younger than chunk c's copies are {store c-2, CP copies
//   of c+1, store c-1}: s_waitcnt vmcnt(CP + 2).  (Waiting for vmcnt(CP) instead made every chunk wait for the write
//   acknowledgement of the level store issued just before it: 4 of the pass's 16 us.)
// -------------------------------------------------------------------------------------------------------------------
constexpr int XS_A = 64 * 256, XS_AUX = 1024;               // FQ(A)^T stage; {scales, LayerNorm weight, bias, (unused)} x 256 B
constexpr int xs_slot(int rows) { return rows * 256 + XS_A + XS_AUX; }
constexpr int xs_lds(int rows) { return 3 * xs_slot(rows) + 256; }            // 63.25 KB / 75.25 KB: two workgroups per CU
typedef _Float16 xs_h8 __attribute__((ext_vector_type(8)));
// read through the fp16 vector type: hipcc drains the copies in flight (s_waitcnt vmcnt(0)) in front of a float-typed LDS read
// that follows an LDS-DMA, but not in front of a half-typed one
__device__ __forceinline__ f32x4 xs_ld16(const char* p) { return __builtin_bit_cast(f32x4, *reinterpret_cast<const xs_h8*>(p)); }
__device__ __forceinline__ void glds4(const void* gsrc, void* lds_wave_base) {   // 4 B per lane: 256 B per instruction
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 4, 0, 0);
}

// LN: the LayerNorm prologue (XPassArgs::ln_w): row statistics while the first two chunks are in flight, then every thread
// normalises its own 16-byte slot of a landed chunk in place (one more barrier per chunk) and keeps it for its level pass.
#if SPQ_XP_DIAG & 128   // in-kernel stamps (s_memtime) of the loop's segments, summed per wave: tools/xpass_probe.py --stamps
__device__ unsigned long long g_xp_stamps[2048 * 8 * 4];
#define SPQ_XP_STAMP(var) const unsigned long long var = __builtin_readcyclecounter()
#else
#define SPQ_XP_STAMP(var) const unsigned long long var = 0
#endif
// A8 = 3 (round 2, later): the LIMB form of the pass -- any other input quantizer (log, asymmetric, 13..24 bit): every thread
// applies the quantize-dequantize to its four elements (fq_dispatch: the same function as the panel kernels, the log normalised
// levels from an LDS table) and stores the two fp16 limbs of FQ(x) * 2^G (two 8-byte stores per chunk, hence vmcnt counts of
// CP + 4 / 4); the chunk's zero points (log: minima) ride in the aux line that carries the LayerNorm weight otherwise, so the
// LayerNorm prologue and the limb form exclude each other.  The panel kernels spent the log arithmetic (ALU-bound) and the
// copies one after the other; here the ring keeps two chunks of copies in flight under it.
template <int ROWS, int A8, bool LN>   // A8 0: fp16 levels; 1: bytes q + 128; 2: int8 q; 3: two fp16 limbs of FQ(x) * 2^G
__device__ __forceinline__ void xpass_stream_body(const XPassArgs& a, char* xsm, const int block) {
  constexpr int NW = ROWS / 4, SLOT = xs_slot(ROWS), XS_X = ROWS * 256, APW = 16 / NW;   // waves; FQ(A)^T pieces per wave
  float* lnst = reinterpret_cast<float*>(xsm + 3 * SLOT);   // LN only: {mean, den} per row
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ws = w & 3, rg = w >> 2;                        // k-slice of a chunk; 16-row group
  const int m0 = block * ROWS;
  const float qhi = (float)((1 << (a.bits - 1)) - 1), qlo = -qhi;
  const int l15 = lane & 15, q4 = lane >> 4;
  const int nck = (SPQ_XP_DIAG & 32) ? 2 : a.K / 64;
  const int prow = 4 * w + q4;                              // == tid >> 4: my row of the x chunk, my row (+ 4 NW i) of FQ(A)^T
  const int kof = (l15 ^ (prow & 15)) << 2;                 // LDS slot l15 of row prow holds source columns kof .. kof + 3

  const float* x_src = a.x + (int64_t)min(m0 + prow, a.M - 1) * a.K + kof;
  const float* a_src = a.aT + (int64_t)prow * a.K + kof;
  const int64_t a_step = (int64_t)4 * NW * a.K;
  // the chunk's constants: wave 0 brings the 64 scales, wave 1 / 2 the LayerNorm weight / bias, the others a dummy line
  const int auxw = ws < 3 && rg == 0 ? ws : 3;
  // aux lines: 0 scales, 1 / 2 LayerNorm weight / bias, 3 the limb form's zero points (log: minima) when LayerNorm is on too (else
  // line 1).  Line 3 is written by every wave that has no line of its own -- with the same bytes.
  constexpr int ZP_LINE = LN ? 3 : 1;
  const float* aux_src = (auxw == 0 && a.x_pc) ? a.sx + lane : (LN && auxw == 1) ? a.ln_w + lane : (LN && auxw == 2) ? a.ln_b + lane
                         : (A8 == 3 && auxw == ZP_LINE && a.x_pc) ? a.zx + lane : nullptr;
  auto issue = [&](int c, int slot) {
    char* s = xsm + slot * SLOT;
    if (!(SPQ_XP_DIAG & 8) || c < 2) glds16(x_src + c * 64, s + w * 1024); else glds4(x_src, s + XS_X + XS_A + 768);
#pragma unroll
    for (int i = 0; i < APW; ++i)
      if (!(SPQ_XP_DIAG & 1) || c < 2) glds16(a_src + i * a_step + c * 64, s + XS_X + (NW * i + w) * 1024); else glds4(x_src, s + XS_X + XS_A + 768);
    glds4(aux_src ? aux_src + c * 64 : x_src, s + XS_X + XS_A + auxw * 256);
  };
  f32x4 acc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int e = 0; e < 4; ++e) acc[i][e] = 0.f;
  const float s_pt = a.x_pc ? 0.f : a.sx[0];
  const float z_pt = (A8 == 3 && !a.x_pc) ? a.zx[0] : 0.f;
  const float pscale = (A8 == 3) ? a.xscale[0] : 1.f;
  __shared__ float s_qn[A8 == 3 ? 256 : 1];
  const float* qn_lut = (A8 == 3) ? fill_log_qn_lut(s_qn, a.bits, a.qtype, a.symmetric) : nullptr;
  if (block == 0 && a.zero_ptr) for (int i = tid; i < a.zero_n; i += ROWS * 16) a.zero_ptr[i] = 0;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // the scalar's load (and those stores) are out of the count
  issue(0, 0);
  if (nck > 1) issue(1, 1);
  if (LN) ln_panel_stats(a, m0, ROWS, lnst);                // its loads drain the count: chunks 0 and 1 land with them
  const int64_t q_dst = (int64_t)min(m0 + prow, a.M - 1) * a.Kp + kof;
  const int pa = 4 * ws + q4;                               // 16-B source chunk of this lane's MFMA operands: k = 16 ws + 4 q4 ..
  int slot = 0;
  unsigned long long st_wait = 0, st_bar = 0, st_body = 0;
  SPQ_XP_STAMP(t_begin);
  for (int c = 0; c < nck; ++c) {
    SPQ_XP_STAMP(t0);
    // chunk c's copies are done once only the younger operations remain: {store c-2, CP copies of c+1, store c-1}
    // (A8 == 3: two limb stores per chunk instead of one level store -> CP + 4 and 4)
    if (c + 1 < nck && c >= 2) {
      if (A8 == 3) { if (APW == 4) asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); }
      else if (APW == 4) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    } else if (c + 1 < nck) {
      if (APW == 4) asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    } else if (c >= 2) { if (A8 == 3) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); }
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    SPQ_XP_STAMP(t1);
    __builtin_amdgcn_s_barrier();                           // chunk c is in LDS; every wave is done with chunk c - 1
    asm volatile("" ::: "memory");
    SPQ_XP_STAMP(t2);
    if (c + 2 < nck) issue(c + 2, slot == 0 ? 2 : slot - 1);
    SPQ_XP_STAMP(t2a);
    char* xs = xsm + slot * SLOT;
    const char* as = xs + XS_X;
    const char* aux = as + XS_A;
    f32x4 v;
    if (LN) {
      const float mean = lnst[2 * prow], den = lnst[2 * prow + 1];
      v = xs_ld16(xs + prow * 256 + l15 * 16);
      const f32x4 wv = xs_ld16(aux + 256 + kof * 4), bb = xs_ld16(aux + 512 + kof * 4);
      v.x = ln_apply(v.x, mean, den, wv.x, bb.x); v.y = ln_apply(v.y, mean, den, wv.y, bb.y);
      v.z = ln_apply(v.z, mean, den, wv.z, bb.z); v.w = ln_apply(v.w, mean, den, wv.w, bb.w);
      *reinterpret_cast<xs_h8*>(xs + prow * 256 + l15 * 16) = __builtin_bit_cast(xs_h8, v);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();                         // the chunk is normalised for every reader
      asm volatile("" ::: "memory");
    }
    // One scheduling region: the chunk's LDS reads, then its 16 fp32 MFMAs with the level arithmetic between them.  Stamped
    // (SPQ_XP_DIAG=128, cycles per chunk and wave at 8192 x 768): copies' issue 230-460, barrier 80-590, the rest 1400-1660 --
    // of which the MFMAs alone are 750 (two waves share a SIMD's pipe) and the level pass alone 715.  Moving the level pass a
    // chunk later (on registers) or running the two row groups in opposite phase order did not shorten the chunk.
    const f32x4 av = xs_ld16(xs + (rg * 16 + l15) * 256 + ((pa ^ l15) << 4));
    f32x4 bv[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) bv[t] = xs_ld16(as + (t * 16 + l15) * 256 + ((pa ^ l15) << 4));
    if (!LN) v = xs_ld16(xs + prow * 256 + l15 * 16);
    f32x4 sc = xs_ld16(aux + kof * 4);
    if (!a.x_pc) { sc.x = s_pt; sc.y = s_pt; sc.z = s_pt; sc.w = s_pt; }
    if (A8 == 3) {                                          // limb form: MFMAs on the raw x, FQ + limb split of my four elements
      f32x4 zp = xs_ld16(aux + ZP_LINE * 256 + kof * 4);
      if (!a.x_pc) { zp.x = z_pt; zp.y = z_pt; zp.z = z_pt; zp.w = z_pt; }
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[e], bv[t][e], acc[t], 0, 0, 0);
      union { _Float16 h[4]; uint2 u; } hi, lo;
#pragma unroll
      for (int e = 0; e < 4; ++e) split2(fq_dispatch(v[e], sc[e], zp[e], a.bits, a.qtype, a.symmetric, qn_lut) * pscale, hi.h[e], lo.h[e]);
      *reinterpret_cast<uint2*>(a.qx + q_dst + c * 64) = hi.u;
      *reinterpret_cast<uint2*>(a.xl + q_dst + c * 64) = lo.u;
      slot = slot == 2 ? 0 : slot + 1;
      continue;
    }
    float q[4];
    // levels by reciprocal multiply (minmax_level_fast); a lane whose quotient sits on a rounding tie sends its wave through
    // the IEEE division below (about one wave-chunk in 500 on Gaussian rows)
    bool unsafe = !(fmaxf(fmaxf(sc[0], sc[1]), fmaxf(sc[2], sc[3])) < 0x1p100f);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      if (!(SPQ_XP_DIAG & 2)) {
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[e], bv[t][e], acc[t], 0, 0, 0);
      } else acc[e][0] += av[e] + bv[e][e];
      q[e] = (SPQ_XP_DIAG & 4) ? 0.f : fminf(fmaxf(minmax_level_fast(v[e], __builtin_amdgcn_rcpf(sc[e]), unsafe), qlo), qhi);
    }
    __builtin_amdgcn_sched_group_barrier(0x100, LN ? 6 : 7, 0);   // every LDS read of the chunk up front
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);    // one MFMA
      __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);    // VALU instructions in its shadow
    }
    if (!(SPQ_XP_DIAG & 4) && __builtin_amdgcn_ballot_w64(unsafe) != 0) {
#pragma unroll
      for (int e = 0; e < 4; ++e) q[e] = minmax_level<true>(v[e], sc[e], 0.f, qlo, qhi);
    }
    if ((!(SPQ_XP_DIAG & 4) && !(SPQ_XP_DIAG & 64)) || a.M == 12345 || ((SPQ_XP_DIAG & 64) && q[0] + q[1] + q[2] + q[3] == 12345.f))
      store_levels4(a.qx, q_dst + c * 64, q[0], q[1], q[2], q[3], A8);
    slot = slot == 2 ? 0 : slot + 1;
    if (SPQ_XP_DIAG & 128) {
      asm volatile("s_nop 0" :: "v"(acc[0][0]), "v"(acc[1][0]), "v"(acc[2][0]), "v"(acc[3][0]) : "memory");   // the MFMAs are done
      SPQ_XP_STAMP(t3);
      st_wait += t2a - t2; st_bar += t2 - t1; st_body += t3 - t2a;   // [0]: the copies' issue, [1]: barrier, [2]: the rest of the body
    }
  }
#if SPQ_XP_DIAG & 128
  if (lane == 0 && block < 2048) {
    unsigned long long* o = g_xp_stamps + ((int64_t)block * 8 + w) * 4;
    o[0] = st_wait; o[1] = st_bar; o[2] = st_body; o[3] = __builtin_readcyclecounter() - t_begin;
  }
#endif
  __syncthreads();                                          // the ring is free for the reduction
  if ((SPQ_XP_DIAG & 16) && a.M != 12345) return;
  xpass16_finish<ROWS>(a, acc, reinterpret_cast<float*>(xsm), m0, tid);
}

template <int ROWS, int A8, bool LN>
__global__ __launch_bounds__(ROWS * 16, 2) void xpass_stream_kernel(XPassArgs a) {
  extern __shared__ __attribute__((aligned(16))) char xsm[];
  xpass_stream_body<ROWS, A8, LN>(a, xsm, (int)blockIdx.x);
}

// The same launch with the weight-side row work as EXTRA workgroups (spq_fwd_args.prepare): blocks 0 .. nx-1 are the activation
// pass, every later block prepares ROWS / 4 weight rows, one wave per row (prep_row_wave: FQ(W), fold sx, FQ(B) column, exponent,
// limb split).  The two are independent once FQ(A)^T exists (fq_transpose_kernel, a few dozen workgroups, goes first), both are
// latency-bound streams, and at the shapes where 32-row activation workgroups cover the chip once a CU has room for a second
// workgroup of this size: the row blocks run beside the activation blocks instead of as a launch of their own before them.
// MODE as prep_wave_kernel (0: fp16 limb planes, 1: int8 levels).
template <int ROWS, int A8, int MODE>
__global__ __launch_bounds__(ROWS * 16, 2) void xpass_stream_prep_kernel(XPassArgs a, PrepArgs pa, int nx) {
  extern __shared__ __attribute__((aligned(16))) char xsm[];
  if ((int)blockIdx.x < nx) { xpass_stream_body<ROWS, A8, false>(a, xsm, (int)blockIdx.x); return; }
  constexpr int NW = ROWS / 4;                              // waves = rows per block
  const int n0 = ((int)blockIdx.x - nx) * NW;
  float (*sB)[128] = reinterpret_cast<float (*)[128]>(xsm);   // the rows' raw LoRA-B columns B[j][n0 .. n0 + NW - 1], as prep_wave_kernel
  const bool staged = pa.B && n0 + NW - 1 < pa.N;
  float* d0 = nullptr; float* d1 = nullptr;
  float v0 = 0.f, v1 = 0.f;
  if (staged) {
    const int e0 = threadIdx.x, e1 = threadIdx.x + NW * 64;
    d0 = &sB[e0 % NW][e0 / NW];
    if (e0 < pa.r * NW) v0 = pa.B[(int64_t)(e0 / NW) * pa.N + n0 + (e0 % NW)];
    if (e1 < pa.r * NW) { d1 = &sB[e1 % NW][e1 / NW]; v1 = pa.B[(int64_t)(e1 / NW) * pa.N + n0 + (e1 % NW)]; }
  }
  const int wv = threadIdx.x >> 6;
  prep_row_wave<MODE, 4>(pa, n0 + wv, threadIdx.x & 63, staged ? &sB[wv][0] : nullptr, d0, d1, v0, v1);
}

// =================================================================================================
// The contraction.
// =================================================================================================
struct GemmF16Args {
  const _Float16 *qx, *thi, *tlo;           // [Mp,Kp], [Mp,Rp], [Mp,Rp]
  const _Float16 *Whi, *Wlo, *Bhi, *Blo;    // [Np,Kp] x2, [Np,Rp] x2
  const float *rowinv, *rowscale, *bias;    // [Mp], [Np], [N] (nullable)
  float* y;
  int M, N, Kp, Rp;                         // Rp = 0: no LoRA
  int tiles_m, tiles_n;
  unsigned long long* dbg;                  // tools/gemm_bench only (DIAG & 16): per-workgroup clock stamps
  // SPQ_PATH_F16X3 (gemm_f16x2_s16_kernel only): the activation operand has two limbs (qx = hi, xl = lo) scaled by
  // xscale[0] = 2^G; per 64-deep k block the stages are (hi x {Whi,Wlo}) then (lo x {Whi})
  const _Float16* xl;
  const float* xscale;                      // device {2^G, 2^-G}; null when a_limbs == 1
  int a_limbs;
  // levels-out store (spq_fwd_args::out_levels; gemm_f16x2_t128_kernel<.., LV = 1> only): y may be null then
  _Float16* lv; const float* lv_scale; int lv_ld, lv_pc; float lv_qhi;
  // LV = 2: two fp16 limbs of FQ(v) * 2^G for any consumer quantizer (hi -> lv, lo -> lv_lo)
  _Float16* lv_lo; const float* lv_zero; const float* lv_xscale; int lv_qtype, lv_sym, lv_bits;
  // split-K (gemm_f16x2_t128_kernel): `split` workgroups share a tile's k range; partial sums meet in sk_part, see the kernel
  int split; int* sk_cnt; float* sk_part;
};

// ---- LDS: two 64-deep stage buffers (A 256x64 f16 = 32 KB, B hi/lo 128x64 f16 = 16 KB each) + a dedicated
// epilogue region.  128-byte rows of eight 16-B chunks; chunk c of row r is stored at chunk position c ^ ((r>>1)&7):
// the 16 lanes of every ds_read_b128 group (row = lane&31 per fragment) hit 16 distinct 16-B slots of the bank row.
constexpr int STAGE_A = GM * GK * 2;               // 32 KB
constexpr int STAGE_B = GN * GK * 2;               // 16 KB per limb
constexpr int STAGE_BYTES = STAGE_A + 2 * STAGE_B; // 64 KB
constexpr int EPI_WAVE = 16 * 144;                 // 16 rows x (32 floats + pad) per compute wave
constexpr int GEMM_LDS = 2 * STAGE_BYTES + 8 * EPI_WAVE;   // 128 KB + 18 KB
constexpr int GEMM_THREADS = 512;                  // 8 waves = 4(M) x 2(N), 64x64 outputs each

__device__ __forceinline__ int swz(int row, int chunk) { return chunk ^ ((row >> 1) & 7); }

// -------------------------------------------------------------------------------------------------------------------
// 128 x 128 tiles, 4 waves (2 x 2 of 64 x 64), ONE 48 KB stage buffer: 57 KB of LDS, so two workgroups share a CU and cover
// each other's copy latency, barriers, tile prologues and epilogues (inside one workgroup of gemm_f16x2_s16_kernel those
// phases serialise).  Per stage: wait for the copies + barrier, MFMAs, barrier, issue the next stage's copies into the same
// buffer.  Same operand layout, same MFMA, same epilogue as the 256 x 128 kernel.
// -------------------------------------------------------------------------------------------------------------------
constexpr int T128_STAGE_A = 128 * GK * 2;                 // 16 KB
constexpr int T128_STAGE = T128_STAGE_A + 2 * STAGE_B;     // 48 KB
// The epilogue's per-wave transpose slices live INSIDE the stage buffer (free once the tile's last stage has been read), so a
// workgroup needs 48 KB and THREE share a CU (the kernel takes 164-168 VGPRs: three waves per SIMD fit).  Measured: time per
// stage per workgroup (~1.9 us) is mostly copy issue -> landed latency and barriers, hardly the 0.5 us of MFMAs, so a third
// workgroup per CU fills what two leave idle.  T128_WGS = 2 keeps the separate 9-KB epilogue region (the round-1 kernel).
#ifndef T128_WGS
#define T128_WGS 3
#endif
#ifndef T128_DIAG      // tools/gemm_bench only (the library builds 0): 1 = no copies after a tile's first stage, 2 = no MFMAs /
#define T128_DIAG 0    // fragment reads, 4 = no epilogue stores
#endif
#ifndef T128_ORDER
#define T128_ORDER 0
#endif
// Issue priority of the CU's three resident workgroups.  Left alone (0) the SIMD arbiter favours the oldest wave: stamped at the
// headline shape, the first-dispatched third of the grid finishes its two tiles after 56 us, the second after 60 us and the
// last-dispatched after 75 us, alone on its CU for the last 15 us -- the kernel lasts as long as the starved third.  7: every
// workgroup takes the priorities 0/1/2 in turn, a sixth of its stages each, offset by its dispatch third, so the three advance
// together (ends 64-73 us, kernel 80.5 -> 77.2 us).  1-6: other schedules tried in tools/t128_bench.hip (fixed priorities only move
// the starvation to another third; rotation every stage 77.1-77.5; mirrored per tile 77.4-78.1).
#ifndef T128_PRIO
#define T128_PRIO 7
#endif
#ifndef T128_PRIO_DIV
#define T128_PRIO_DIV 6
#endif
#ifndef T128_PRIO_MID
#define T128_PRIO_MID 1
#endif
#ifndef T128_PRIO_HI
#define T128_PRIO_HI 2
#endif
#if T128_DIAG & 8      // in-kernel stamps of the stage's segments, summed per wave into GemmF16Args::dbg[(block * 4 + wave) * 8 ..]
#define T128_STAMP(v) const unsigned long long v = __builtin_readcyclecounter()
#else
#define T128_STAMP(v) const unsigned long long v = 0
#endif
constexpr bool T128_DEFER = T128_WGS == 3;
constexpr int T128_LDS = T128_DEFER ? T128_STAGE : T128_STAGE + 4 * EPI_WAVE;
#ifndef T128_REUSE_BHI
#define T128_REUSE_BHI 1
#endif
#ifndef T128_GROUP_M
#define T128_GROUP_M 8      // measured at the headline shape: 8 -> 81.5 us, 16 -> 82.9, 32 -> 84.5, 4 -> 84.5
#endif
// LV = 1: the store also writes the next layer's level matrix (GemmF16Args::lv): q = clamp(rint(o / s[n]), +-qhi) of the very
// value o it stores -- IEEE division, round half to even, as quantization_methods.py:14-15 -- four fp16 levels (8 B) per lane
// next to (or, with g.y null, instead of) the 16 B of fp32.
template <int AL, int EPI, int LV = 0, bool SK = false, bool RAG = false>   // SK: the split-K form; RAG: N % 4 != 0 (scalar stores); both with the plain epilogue only (EPI = LV = 0)
__global__ __launch_bounds__(256, T128_WGS) void gemm_f16x2_t128_kernel(GemmF16Args g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = w >> 1, wn = w & 1;
  const int l15 = lane & 15, q4 = lane >> 4;
  const int tiles_m = g.tiles_m * 2;                        // g.tiles_m counts 256-row tiles (Mp is a multiple of 256)
  const int nwg = tiles_m * g.tiles_n;
  const int nl = (g.Rp / GK) * 2;
  const int T = nl + AL * (g.Kp / GK);
  const int gstride = (int)gridDim.x;
  auto tile_of = [&](int p, int& bm, int& bn) {             // same XCD-aware band order, T128_GROUP_M tile rows per band
    const int q8 = nwg >> 3, r8 = nwg & 7, xcd = p & 7;
    const int wgid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (p >> 3);
    constexpr int GROUP_M = T128_GROUP_M;
    const int band = wgid / (GROUP_M * g.tiles_n);
    const int band_rows = min(GROUP_M, tiles_m - band * GROUP_M);
    const int in_band = wgid - band * GROUP_M * g.tiles_n;
    bm = (band * GROUP_M + in_band % band_rows) * 128;
    bn = (in_band / band_rows) * GN;
  };
  const float lora_to_base = (AL == 2) ? g.xscale[0] : 1.f;
  const float out_scale = (AL == 2) ? g.xscale[1] : 1.f;
  // Split-K (g.split = S > 1, chosen by the host when the tiles alone leave workgroup slots empty: N <= 1024 at M = 8192): unit
  // p = h * nwg + tile accumulates the stages [bound(h), bound(h + 1)) of its tile; the LoRA stages and their rescale stay whole
  // in h = 0.  The S partial sums are combined in FIXED order h = 0, 1, .. by whichever unit arrives last (below), so the result
  // does not depend on the arrival order.
  const int S = (SK && g.split > 1) ? g.split : 1;
  const int units = nwg * S;
  int p = blockIdx.x;
  if (p >= units) return;
  int sk_h = 0, tile = p;
  int t_lo = 0, t_hi = T;
  auto unit_of = [&](int u) {
    sk_h = SK ? u / nwg : 0; tile = u - sk_h * nwg;
    t_lo = 0; t_hi = T;
    if (SK && S > 1) {
      const int share = (T + S - 1) / S;
      auto bound = [&](int h) {
        if (h <= 0) return 0;
        if (h >= S) return T;
        int b = max(nl, h * share);
        if (AL == 2) b = nl + ((b - nl + 1) & ~1);            // a two-limb stage and its one-limb stage stay together
        return min(b, T);
      };
      t_lo = bound(sk_h); t_hi = bound(sk_h + 1);
    }
  };
  unit_of(p);
  int bm, bn;
  tile_of(tile, bm, bn);

  const int prow = lane >> 3, pchunk = lane & 7;
  int pc_row[4], pc_col[4];                                 // wave w owns pieces 4w..4w+3 of A and of each B limb
#pragma unroll
  for (int i = 0; i < 4; ++i) { pc_row[i] = (4 * w + i) * 8 + prow; pc_col[i] = swz(pc_row[i], pchunk) * 8; }
  auto issue = [&](int t, int tbm, int tbn) {
    if ((T128_DIAG & 1) && t != 0) return;
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
    const _Float16* Ab = A + (int64_t)tbm * lda + k0;
    const _Float16* Bhb = Bh + (int64_t)tbn * ldb + k0;
    const _Float16* Blb = Bl + (int64_t)tbn * ldb + k0;
    // a one-limb stage (tlo x Bhi, lo x Whi) always follows the two-limb stage of the SAME k block in the same tile, whose B-hi
    // copy is still in the buffer (nothing overwrites that region in between): only its A operand is copied -- 16 KB instead
    // of 32 KB, a fifth of the copy bytes of a three-product tile
    const bool copy_bh = two || !T128_REUSE_BHI || (T128_DIAG & 1);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      glds16(Ab + (pc_row[i] * lda + pc_col[i]), smem + (4 * w + i) * 1024);
      const int off = pc_row[i] * ldb + pc_col[i];
      if (copy_bh) glds16(Bhb + off, smem + T128_STAGE_A + (4 * w + i) * 1024);
      if (two) glds16(Blb + off, smem + T128_STAGE_A + STAGE_B + (4 * w + i) * 1024);
    }
  };

  const int sx7 = (l15 >> 1) & 7;
  const int fa_row = (wm * 64 + l15) * 128;
  const int fb_row = T128_STAGE_A + (wn * 64 + l15) * 128;
  int koff[2];
#pragma unroll
  for (int s = 0; s < 2; ++s) koff[s] = ((4 * s + q4) ^ sx7) * 16;
  struct Frags { f16x8 a[4], bh[4], bl[4]; };
  f32x4 acc[4][4];
  auto load_frags = [&](Frags& f, int s, bool two) {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      f.a[t] = *reinterpret_cast<const f16x8*>(smem + fa_row + t * 2048 + koff[s]);
      f.bh[t] = *reinterpret_cast<const f16x8*>(smem + fb_row + t * 2048 + koff[s]);
      if (two) f.bl[t] = *reinterpret_cast<const f16x8*>(smem + fb_row + STAGE_B + t * 2048 + koff[s]);
    }
  };
  auto mfma_block = [&](const Frags& f, bool two) {
#pragma unroll
    for (int tm = 0; tm < 4; ++tm)
#pragma unroll
      for (int tn = 0; tn < 4; ++tn) {
        acc[tm][tn] = __builtin_amdgcn_mfma_f32_16x16x32_f16(f.a[tm], f.bh[tn], acc[tm][tn], 0, 0, 0);
        if (two) acc[tm][tn] = __builtin_amdgcn_mfma_f32_16x16x32_f16(f.a[tm], f.bl[tn], acc[tm][tn], 0, 0, 0);
      }
  };
  // stage: its copies were issued earlier.  Afterwards the buffer is free and the next stage (nt of tile nbm, nbn) goes in.
  unsigned long long st_sum[4] = {0, 0, 0, 0};
  int prio_ctr = (int)(blockIdx.x / (gridDim.x / 3 > 0 ? gridDim.x / 3 : 1));
  int stage_ctr = 0;
  auto stage = [&](bool two, bool have_next, int nt, int nbm, int nbn) {
#if T128_PRIO == 4     // rotate the issue priority of the CU's three workgroups stage by stage
    prio_ctr = prio_ctr == 2 ? 0 : prio_ctr + 1;
    if (prio_ctr == 0) __builtin_amdgcn_s_setprio(0); else if (prio_ctr == 1) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(2);
#endif
#if T128_PRIO == 6 || T128_PRIO == 7   // the three workgroups of a CU take the three priorities in turn, 1/T128_PRIO_DIV of their stages each
    {
      constexpr int DIV = T128_PRIO == 6 ? 3 : T128_PRIO_DIV;
      const int per = max(1, ((t_hi - t_lo) * ((units + gstride - 1) / gstride) + DIV - 1) / DIV);
      if (stage_ctr % per == 0) {
        const int pr = (prio_ctr + stage_ctr / per) % 3;
        if (pr == 0) __builtin_amdgcn_s_setprio(0); else if (pr == 1) __builtin_amdgcn_s_setprio(T128_PRIO_MID); else __builtin_amdgcn_s_setprio(T128_PRIO_HI);
      }
      ++stage_ctr;
    }
#endif
    T128_STAMP(s0);
    __syncthreads();                                         // vmcnt(0) + barrier: the stage has landed
    Frags f0, f1;
#if T128_ORDER == 1   // tools/t128_bench only
    // fragments into registers, barrier, the NEXT stage's copies, and only then this stage's MFMAs: the copies travel under
    // the workgroup's own 64 MFMAs per wave instead of after them.  Measured: 85.5 -> 81 us with two workgroups per CU (no gain
    // over three workgroups in the order below, 80 us); with three it needs both fragment sets live across the copies' address
    // arithmetic and spills (168 VGPRs + 144 B of scratch: 113 us)
    if (!(T128_DIAG & 2)) { load_frags(f0, 0, two); load_frags(f1, 1, two); }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory");
    if (have_next) issue(nt, nbm, nbn);
    __builtin_amdgcn_sched_barrier(0);
    if (!(T128_DIAG & 2)) { mfma_block(f0, two); mfma_block(f1, two); }
#else
    T128_STAMP(s1);                                          // [0] s0 -> s1: waiting for the stage's copies + barrier
    if (!(T128_DIAG & 2)) {
      load_frags(f0, 0, two);
      load_frags(f1, 1, two); mfma_block(f0, two);
      mfma_block(f1, two);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");           // my fragment reads are complete (and may not sink below)
    if (T128_DIAG & 8) asm volatile("s_nop 0" :: "v"(acc[3][3][0]), "v"(acc[0][0][0]) : "memory");   // ... and the MFMAs (stamps only)
    T128_STAMP(s2);                                          // [1] s1 -> s2: fragment reads + MFMAs
    __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory");   // every wave has read its fragments
    T128_STAMP(s3);                                          // [2] s2 -> s3: barrier
    if (have_next) issue(nt, nbm, nbn);
    T128_STAMP(s4);                                          // [3] s3 -> s4: issuing the next stage's copies
    if (T128_DIAG & 8) { st_sum[0] += s1 - s0; st_sum[1] += s2 - s1; st_sum[2] += s3 - s2; st_sum[3] += s4 - s3; }
#endif
  };

#if T128_PRIO == 1
  if (blockIdx.x >= 2 * (gridDim.x / 3)) __builtin_amdgcn_s_setprio(1);
#elif T128_PRIO == 2
  if (blockIdx.x >= 2 * (gridDim.x / 3)) __builtin_amdgcn_s_setprio(2); else if (blockIdx.x >= gridDim.x / 3) __builtin_amdgcn_s_setprio(1);
#elif T128_PRIO == 5
  if (blockIdx.x >= 2 * (gridDim.x / 3)) __builtin_amdgcn_s_setprio(2); else if (blockIdx.x >= gridDim.x / 3) __builtin_amdgcn_s_setprio(1);
#elif T128_PRIO == 3
  if (blockIdx.x >= 2 * (gridDim.x / 3)) __builtin_amdgcn_s_setprio(3); else if (blockIdx.x >= gridDim.x / 3) __builtin_amdgcn_s_setprio(1);
#endif
  T128_STAMP(t_kernel);
#if T128_DIAG & 8
  const unsigned long long t_real = __builtin_amdgcn_s_memrealtime();
#endif
  if (t_lo < t_hi) issue(t_lo, bm, bn);
  while (true) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int jj = 0; jj < 4; ++jj)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[i][jj][e] = 0.f;
    const int pn = p + gstride;
    const bool more = !SK && pn < units;                // (split-K: one unit per workgroup, the host sizes the grid so)
    int nbm = 0, nbn = 0;
    if (more) tile_of(pn, nbm, nbn);
    for (int t = t_lo; t < min(nl, t_hi); t += 2) {
      stage(true, true, t + 1, bm, bn);
      const bool last = (t + 2 == t_hi);
      stage(false, !last || (more && !T128_DEFER), last ? 0 : t + 2, last ? nbm : bm, last ? nbn : bn);
    }
    if (nl > 0 && t_lo == 0) {                           // LoRA partial sums -> units of the base sum: * 2^-g[m]
      f32x4 riv[4];                                        // the four loads first: one memory round trip, not four
#pragma unroll
      for (int tm = 0; tm < 4; ++tm) riv[tm] = *reinterpret_cast<const f32x4*>(g.rowinv + bm + wm * 64 + tm * 16 + 4 * q4);
#pragma unroll
      for (int tm = 0; tm < 4; ++tm)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float ri = riv[tm][e];
          if (AL == 2) ri *= lora_to_base;
#pragma unroll
          for (int tn = 0; tn < 4; ++tn) acc[tm][tn][e] *= ri;
        }
    }
    if (AL == 1) {
      for (int t = max(nl, t_lo); t < t_hi; ++t) {
        const bool last = (t + 1 == t_hi);
        stage(true, !last || (more && !T128_DEFER), last ? 0 : t + 1, last ? nbm : bm, last ? nbn : bn);
      }
    } else {
      for (int t = max(nl, t_lo); t < t_hi; t += 2) {
        stage(true, true, t + 1, bm, bn);
        const bool last = (t + 2 == t_hi);
        stage(false, !last || (more && !T128_DEFER), last ? 0 : t + 2, last ? nbm : bm, last ? nbn : bn);
      }
    }
    if (SK && S > 1) {
      // The tile's S units take a ticket each.  All but the last arrival store their accumulators (16 B per lane, the register
      // image) to sk_part[tile][h] and leave.  The last arrival waits until S - 1 are done -- they are running or finished and none
      // of them waits for anything, so the wait ends -- and sums h = 0 .. S - 1 in that order with its own registers in its own
      // place: the sum is the same whoever came last.
      // Coherence: every access to the partial sums carries sc1 (agent scope: performed at the point all XCDs share, like a
      // relaxed agent-scope atomic), so no cache write-back / invalidate is needed -- an agent-scope release fence here
      // (buffer_wbl2: write back the XCD's whole L2) in every wave of 768 workgroups cost 110 us at the attn c_proj shape.
      // Order: stores, s_waitcnt vmcnt(0) in every wave, barrier, then the `done` counter; the reader polls `done`, barrier, loads.
      int* cnt = g.sk_cnt + 2 * tile;                       // {tickets, done}; zeroed by the host side of this call
      f32x4* part = reinterpret_cast<f32x4*>(g.sk_part) + (int64_t)tile * S * (16 * 256) + tid;
      int* tk = reinterpret_cast<int*>(smem);               // (the stage buffer is free: the last stage ended with a barrier)
      if (tid == 0) tk[0] = __hip_atomic_fetch_add(cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __syncthreads();
      const int ticket = tk[0];
      __syncthreads();                                      // (the epilogue writes its slices over tk)
      if (ticket != S - 1) {
#pragma unroll
        for (int i = 0; i < 16; ++i)
          asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(part + (sk_h * 16 + i) * 256), "v"(acc[i >> 2][i & 3]) : "memory");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) __hip_atomic_fetch_add(cnt + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        break;
      }
      if (tid == 0) {
        int spins = 0;
        while (__hip_atomic_load(cnt + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != S - 1 && ++spins < (1 << 22)) __builtin_amdgcn_s_sleep(4);
        tk[0] = spins >= (1 << 22);                          // (cannot happen by construction; if it ever does, the tile comes out NaN, not silently wrong)
      }
      __syncthreads();
      const bool sk_lost = tk[0] != 0;
      __syncthreads();                                      // (the epilogue writes its slices over tk)
      auto combine = [&](auto HALF) {                       // 8 accumulator blocks at a time: 32 registers of loads in flight
        constexpr int half = decltype(HALF)::value;
        f32x4 run[8];
#pragma unroll
        for (int h = 0; h < 4; ++h)
          if (h < S) {
            f32x4 v[8];
            if (sk_h != h) {
#pragma unroll
              for (int i = 0; i < 8; ++i) asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v[i]) : "v"(part + (h * 16 + half * 8 + i) * 256) : "memory");
              asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            } else {
#pragma unroll
              for (int i = 0; i < 8; ++i) v[i] = acc[(half * 8 + i) >> 2][i & 3];
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) run[i] = (h == 0) ? v[i] : run[i] + v[i];
          }
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[(half * 8 + i) >> 2][i & 3] = run[i];
      };
      combine(std::integral_constant<int, 0>{});
      combine(std::integral_constant<int, 1>{});
      if (sk_lost) {
#pragma unroll
        for (int i = 0; i < 16; ++i)
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[i >> 2][i & 3][e] = __builtin_nanf("");
      }
    }
    // epilogue operands: fetched here, not at the tile's start (16 registers that three waves per SIMD do not leave through
    // the stage loop); the CU's other workgroups cover the load's latency
    float4 ep_rs[2], ep_bv[2], ep_ls[2], ep_lz[2];
    const float lv_pscale = (LV == 2) ? g.lv_xscale[0] : 1.f;
#pragma unroll
    for (int tn = 0; tn < 2; ++tn) {
      const int n = bn + wn * 64 + tn * 32 + (lane & 7) * 4;
      ep_rs[tn] = make_float4(0.f, 0.f, 0.f, 0.f); ep_bv[tn] = ep_rs[tn];
      if (LV) {
        ep_ls[tn] = make_float4(1.f, 1.f, 1.f, 1.f);
        if (LV == 2) ep_lz[tn] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (n < g.N) {
          if (g.lv_pc) ep_ls[tn] = *reinterpret_cast<const float4*>(g.lv_scale + n);
          else { const float s1 = g.lv_scale[0]; ep_ls[tn] = make_float4(s1, s1, s1, s1); }
          if (LV == 2) {
            if (g.lv_pc) ep_lz[tn] = *reinterpret_cast<const float4*>(g.lv_zero + n);
            else { const float z1 = g.lv_zero[0]; ep_lz[tn] = make_float4(z1, z1, z1, z1); }
          }
        }
      }
      if (n < g.N) {
        ep_rs[tn] = *reinterpret_cast<const float4*>(g.rowscale + n);     // [Np]: padded
        if (AL == 2) { ep_rs[tn].x *= out_scale; ep_rs[tn].y *= out_scale; ep_rs[tn].z *= out_scale; ep_rs[tn].w *= out_scale; }
        if (g.bias) {
          if (!RAG) ep_bv[tn] = *reinterpret_cast<const float4*>(g.bias + n);
          else {                                             // [N], ragged tail
            ep_bv[tn].x = g.bias[n];
            if (n + 1 < g.N) ep_bv[tn].y = g.bias[n + 1];
            if (n + 2 < g.N) ep_bv[tn].z = g.bias[n + 2];
            if (n + 3 < g.N) ep_bv[tn].w = g.bias[n + 3];
          }
        }
      }
    }
    {                                                        // epilogue
      char* eb = smem + (T128_DEFER ? 0 : T128_STAGE) + w * EPI_WAVE;     // T128_DEFER: inside the (now free) stage buffer
      const int c4 = (lane & 7) * 4;
      constexpr bool vecN = !RAG;                           // RAG: rows of y are not 16-B aligned -> scalar stores
      const bool interior = (bm + 128 <= g.M) && (bn + GN <= g.N) && vecN;
#pragma unroll
      for (int tn = 0; tn < 2; ++tn) {
        const int n = bn + wn * 64 + tn * 32 + c4;
        const bool n_ok = n < g.N;
        const float4 rs = ep_rs[tn], bv = ep_bv[tn];
#pragma unroll
        for (int tm = 0; tm < 4; ++tm) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            *reinterpret_cast<float*>(eb + (4 * q4 + e) * 144 + l15 * 4) = acc[tm][2 * tn][e];
            *reinterpret_cast<float*>(eb + (4 * q4 + e) * 144 + (16 + l15) * 4) = acc[tm][2 * tn + 1][e];
          }
#pragma unroll
          for (int it = 0; it < 2; ++it) {
            const int r16 = it * 8 + (lane >> 3);
            const float4 v = *reinterpret_cast<const float4*>(eb + r16 * 144 + c4 * 4);
            const int m = bm + wm * 64 + tm * 16 + r16;
            float4 o;
            o.x = v.x * rs.x + bv.x; o.y = v.y * rs.y + bv.y; o.z = v.z * rs.z + bv.z; o.w = v.w * rs.w + bv.w;
            if (EPI == 1) { o.x = gelu_erf(o.x); o.y = gelu_erf(o.y); o.z = gelu_erf(o.z); o.w = gelu_erf(o.w); }
            float* dst = g.y + (int64_t)m * g.N + n;
            if (LV == 1) {
              const float4 ls = ep_ls[tn];
              const float qhi = g.lv_qhi;
              if (interior || (n_ok && m < g.M))
                store_levels4(g.lv, (int64_t)m * g.lv_ld + n, minmax_level<true>(o.x, ls.x, 0.f, -qhi, qhi), minmax_level<true>(o.y, ls.y, 0.f, -qhi, qhi),
                              minmax_level<true>(o.z, ls.z, 0.f, -qhi, qhi), minmax_level<true>(o.w, ls.w, 0.f, -qhi, qhi), 0);
              if (!g.y) continue;
            }
            if (LV == 2) {                                   // the arithmetic of store_act4's limb branch
              const float4 ls = ep_ls[tn], lz = ep_lz[tn];
              if (interior || (n_ok && m < g.M)) {
                const float f0 = fq_dispatch(o.x, ls.x, lz.x, g.lv_bits, g.lv_qtype, g.lv_sym) * lv_pscale;
                const float f1 = fq_dispatch(o.y, ls.y, lz.y, g.lv_bits, g.lv_qtype, g.lv_sym) * lv_pscale;
                const float f2 = fq_dispatch(o.z, ls.z, lz.z, g.lv_bits, g.lv_qtype, g.lv_sym) * lv_pscale;
                const float f3 = fq_dispatch(o.w, ls.w, lz.w, g.lv_bits, g.lv_qtype, g.lv_sym) * lv_pscale;
                union { _Float16 h[4]; uint2 u; } hi, lo;
                split2(f0, hi.h[0], lo.h[0]); split2(f1, hi.h[1], lo.h[1]); split2(f2, hi.h[2], lo.h[2]); split2(f3, hi.h[3], lo.h[3]);
                *reinterpret_cast<uint2*>(g.lv + (int64_t)m * g.lv_ld + n) = hi.u;
                *reinterpret_cast<uint2*>(g.lv_lo + (int64_t)m * g.lv_ld + n) = lo.u;
              }
              if (!g.y) continue;
            }
            if (T128_DIAG & 4) { if (o.x == 12345.f) *reinterpret_cast<float4*>(dst) = o; }   // (non-temporal stores: no change, 79.3 vs 79.2 us)
            else if (interior) *reinterpret_cast<float4*>(dst) = o;
            else if (n_ok && m < g.M) {
              if constexpr (vecN) *reinterpret_cast<float4*>(dst) = o;
              else {
                dst[0] = o.x;
                if (n + 1 < g.N) dst[1] = o.y;
                if (n + 2 < g.N) dst[2] = o.z;
                if (n + 3 < g.N) dst[3] = o.w;
              }
            }
          }
        }
      }
    }
    if (!more) break;
#if T128_PRIO == 5     // second tile: the priorities of the first, mirrored
    { const int slot = (int)(blockIdx.x / (gridDim.x / 3 > 0 ? gridDim.x / 3 : 1));
      if (slot == 0) __builtin_amdgcn_s_setprio(2); else if (slot == 1) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0); }
#endif
    if (T128_DEFER) {                                       // the slices are done with: the next tile's first stage may land on them
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory");
      issue(0, nbm, nbn);
    }
    p = pn; bm = nbm; bn = nbn; tile = pn;
  }
#if T128_DIAG & 8
  if (g.dbg && lane == 0) {
    unsigned long long* o = g.dbg + ((int64_t)blockIdx.x * 4 + w) * 8;
    for (int i = 0; i < 4; ++i) o[i] = st_sum[i];
    o[4] = __builtin_readcyclecounter() - t_kernel;
    const unsigned long long t_end = __builtin_amdgcn_s_memrealtime();
    o[5] = t_end - t_real;                                  // constant 100 MHz counter: o[4] / o[5] = shader clock / 100 MHz
    o[6] = t_real; o[7] = t_end;
  }
#endif
}

// =================================================================================================
// part2 CPTLinear (cpt_model.py:96-113): the LoRA branch consumes FQ(x) like the base term, so the layer is one contraction
// against  W_eff = FQ(W) + s FQ(B) FQ(A)^T.  Two small kernels build it; the limb split is the ordinary row kernel.
// =================================================================================================
struct CptArgs {
  const float* W; const float* sw; const float* zw;     // [N,K]; weight quantizer params ([N] or [1])
  const float* A; const float* B;                       // [K,r], [N,r]
  const float* sl; const float* zl;                     // the shared LoRA quantizer's params ([r] or [1])
  float* aq; float* bq;                                 // out: FQ(A) [K,r], FQ(B) [N,r]
  float* aqT;                                           // out, nullable: FQ(A)^T [ceil(r/64)*64, K], pad rows zero
  float* w_eff;                                         // out: [N,K]
  int N, K, r;
  int w_pc, w_bits, w_qtype, w_sym;
  int l_pc, l_bits, l_qtype, l_sym;
  float scaling;
};

__global__ __launch_bounds__(256) void cpt_factors_kernel(CptArgs a) {
  const int64_t na = (int64_t)a.K * a.r, nb = (int64_t)a.N * a.r;
  const int rp = (a.r + 63) / 64 * 64;
  const int64_t npad = a.aqT ? (int64_t)(rp - a.r) * a.K : 0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < na + nb + npad; i += (int64_t)gridDim.x * 256) {
    if (i < na) {
      const int j = (int)(i % a.r);
      const int64_t k = i / a.r;
      const float v = fq_dispatch(a.A[i], a.sl[a.l_pc ? j : 0], a.zl[a.l_pc ? j : 0], a.l_bits, a.l_qtype, a.l_sym);
      a.aq[i] = v;
      if (a.aqT) a.aqT[(int64_t)j * a.K + k] = v;
    } else if (i < na + nb) {
      const int64_t e = i - na;
      const int j = (int)(e % a.r);
      a.bq[e] = fq_dispatch(a.B[e], a.sl[a.l_pc ? j : 0], a.zl[a.l_pc ? j : 0], a.l_bits, a.l_qtype, a.l_sym);
    } else {
      a.aqT[(int64_t)a.r * a.K + (i - na - nb)] = 0.f;
    }
  }
}

// 64 x 64 tile of W_eff per workgroup; thread (ty, tx) owns rows ty + 16 i, columns tx + 16 c (i, c < 4)
constexpr int CPT_RMAX = 64;
__global__ __launch_bounds__(256) void cpt_weff_kernel(CptArgs a) {
  __shared__ float bs[64][CPT_RMAX + 1];
  __shared__ float as_[64][CPT_RMAX + 1];
  const int n0 = blockIdx.y * 64, k0 = blockIdx.x * 64;
  const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
  for (int idx = tid; idx < 64 * a.r; idx += 256) {
    const int row = idx / a.r, j = idx - row * a.r;
    bs[row][j] = (n0 + row < a.N) ? a.bq[(int64_t)(n0 + row) * a.r + j] : 0.f;
    as_[row][j] = (k0 + row < a.K) ? a.aq[(int64_t)(k0 + row) * a.r + j] : 0.f;
  }
  __syncthreads();
  float acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[i][c] = 0.f;
  for (int j = 0; j < a.r; ++j) {
    float bv[4], av[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { bv[i] = bs[ty + 16 * i][j]; av[i] = as_[tx + 16 * i][j]; }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int c = 0; c < 4; ++c) acc[i][c] = fmaf(bv[i], av[c], acc[i][c]);
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int n = n0 + ty + 16 * i;
    if (n >= a.N) continue;
    const float swn = a.sw[a.w_pc ? n : 0], zwn = a.zw[a.w_pc ? n : 0];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int k = k0 + tx + 16 * c;
      if (k >= a.K) continue;
      const float wq = fq_dispatch(a.W[(int64_t)n * a.K + k], swn, zwn, a.w_bits, a.w_qtype, a.w_sym);
      a.w_eff[(int64_t)n * a.K + k] = wq + a.scaling * acc[i][c];     // out + lora * scaling, cpt_model.py:113
    }
  }
}

// =================================================================================================
// host side
// =================================================================================================
struct PrepLayout { int64_t Np, Kp, Rp; size_t off_whi, off_wlo, off_bhi, off_blo, total; };
static PrepLayout make_prep_layout(int64_t N, int64_t K, int64_t r) {
  PrepLayout L;
  L.Np = pad_to(N, GN); L.Kp = pad_to(K, GK); L.Rp = r > 0 ? pad_to(r, GK) : 0;
  size_t o = 0;
  L.off_whi = o; o += pad_to((size_t)L.Np * L.Kp * 2, 256);
  L.off_wlo = o; o += pad_to((size_t)L.Np * L.Kp * 2, 256);
  L.off_bhi = o; o += pad_to((size_t)L.Np * L.Rp * 2, 256);
  L.off_blo = o; o += pad_to((size_t)L.Np * L.Rp * 2, 256);
  L.total = o + 256;
  return L;
}

// SPQ_PATH_I8: [int8 plane Np x Kp][Bhi][Blo][bscale Np floats]
struct PrepLayoutI8 { int64_t Np, Kp, Rp; int nl; size_t plane, off_bhi, off_blo, off_bscale, total; };
static PrepLayoutI8 make_prep_layout_i8(int64_t N, int64_t K, int64_t r, int nl) {
  PrepLayoutI8 L;
  L.Np = pad_to(N, GN); L.Kp = pad_to(K, GK); L.Rp = r > 0 ? pad_to(r, GK) : 0; L.nl = nl;
  L.plane = (size_t)L.Np * L.Kp;                           // a multiple of 8192
  size_t o = (size_t)nl * L.plane;
  L.off_bhi = o; o += pad_to((size_t)L.Np * L.Rp * 2, 256);
  L.off_blo = o; o += pad_to((size_t)L.Np * L.Rp * 2, 256);
  L.off_bscale = o; o += pad_to((size_t)L.Np * 4, 256);
  L.total = o + 256;
  return L;
}
static inline int i8_limbs_of(int path) { return path == SPQ_PATH_I8 ? 1 : 0; }

static bool f16x2_shape_ok(int64_t M, int64_t K, int64_t N, int64_t r) {
  return r <= 128 && M < (1 << 30) && N < (1 << 30) && K < (1 << 30);
}

// Per-DEVICE host state (one process may drive several GPUs): the CU count that sizes the persistent grids, and which groups of
// kernels have had their dynamic-LDS limit raised (hipFuncSetAttribute is kept per function per device).
constexpr int kMaxDevices = 64;
static std::mutex g_dev_mu;
static int g_dev_cus[kMaxDevices];
static unsigned g_dev_attr[kMaxDevices];
static int current_device() { int dev = 0; if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); dev = 0; } return dev & (kMaxDevices - 1); }

// persistent grid: one workgroup per CU (a multiple of 8 so that p % 8 stays the XCD label across iterations)
unsigned gemm_grid(int ntiles) {
  const int dev = current_device();
  int cus;
  {
    std::lock_guard<std::mutex> lk(g_dev_mu);
    if (g_dev_cus[dev] == 0) {
      int n = 0;
      g_dev_cus[dev] = (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n >= 8) ? n / 8 * 8 : 256;
    }
    cus = g_dev_cus[dev];
  }
  return (unsigned)(ntiles < cus ? ntiles : cus);
}

// true exactly once per (device, group): the caller then sets the group's function attributes (under the same lock, so a second
// host thread cannot launch on this device before they are in place)
struct AttrOnce {
  std::unique_lock<std::mutex> lk;
  bool first;
  AttrOnce(int group) : lk(g_dev_mu) {
    const int dev = current_device();
    first = !(g_dev_attr[dev] & (1u << group));
    g_dev_attr[dev] |= 1u << group;
  }
};

int prepare_fill(const spq_prepare_args* q, PrepArgs& a, int& at_blocks, bool& wave_ok);

#if SPQ_XP_DIAG & 128
extern "C" int spq_debug_xp_stamps(unsigned long long* host_out, int n) {
  return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_xp_stamps), sizeof(unsigned long long) * n, 0, hipMemcpyDeviceToHost);
}
#endif

// Tuning / A-B switches, read from the environment ONCE (first call) -- nothing on the per-call path calls getenv.
// spq_debug_reload_switches() re-reads them (tests flip a switch inside one process, tests/test_gpu_xpass_stream.py).
//   SPQ_XPASS_STREAM        0: the panel kernel instead of the streaming activation pass; 16 / 32: forced rows per workgroup
//   SPQ_XPASS_STREAM_LIMBS  0: the limb form (F16X3 operand) stays on the panel kernel
//   SPQ_PREP_ROLE           0: the weight rows as a launch of their own instead of extra workgroups of the activation launch
//   SPQ_SPLIT_K             0: never split a tile's k range over several workgroups; 2..4: that many where legal; 1 / unset: by estimate
struct Switches { int stream, stream_limbs, prep_role, split_k; };
static Switches read_switches() {
  Switches w;
  const char* e = getenv("SPQ_XPASS_STREAM");        w.stream = e ? atoi(e) : 1;
  e = getenv("SPQ_XPASS_STREAM_LIMBS");              w.stream_limbs = !(e && e[0] == '0');
  e = getenv("SPQ_PREP_ROLE");                       w.prep_role = !(e && e[0] == '0');
  e = getenv("SPQ_SPLIT_K");                         w.split_k = e ? atoi(e) : 1;
  return w;
}
static Switches g_switches = read_switches();
extern "C" int spq_debug_reload_switches(void) { g_switches = read_switches(); return SPQ_OK; }
int split_k_switch() { return g_switches.split_k; }
// host logic, no launch: how many units per tile the forward would use for this shape under the current switches (CPU-tested)
extern "C" int spq_debug_split_k(int64_t M, int64_t K, int64_t N, int64_t r, int path) {
  if (M <= 0 || K <= 0 || N <= 0 || r < 0 || (path != SPQ_PATH_F16X2 && path != SPQ_PATH_F16X3) || (N & 3) != 0) return 1;
  const int64_t nwg = (pad_to(M, 256) / 128) * (pad_to(N, 128) / 128);
  const int nl = r > 0 ? (int)(pad_to(r, 64) / 64) * 2 : 0;
  const int T = nl + (path == SPQ_PATH_F16X3 ? 2 : 1) * (int)(pad_to(K, 64) / 64);
  return t128_split(nwg, T, nl, (int64_t)T128_WGS * gemm_grid(1 << 30), g_switches.split_k);
}

int fwd_f16x2(const spq_fwd_args* a, hipStream_t st) {
  const bool x3 = a->path == SPQ_PATH_F16X3;
  const int i8nl = i8_limbs_of(a->path);
  if (i8nl) {
    if (!(a->quantize_input && a->qtype == SPQ_MINMAX && a->symmetric && a->bits >= 2 && a->bits <= 8)) {
      set_error("spq_linear_lora_fwd: the int8 operand paths need a symmetric minmax input quantizer with 2..8 bits "
                "(got qtype=%d symmetric=%d bits=%d quantize_input=%d)", a->qtype, a->symmetric, a->bits, a->quantize_input);
      return SPQ_ERR_UNSUPPORTED;
    }
    if (a->x_per_channel) { set_error("spq_linear_lora_fwd: SPQ_PATH_I8 needs a per-tensor input scale"); return SPQ_ERR_UNSUPPORTED; }
    if (a->prepare && a->prepare->path != a->path) { set_error("spq_linear_lora_fwd: prepare args are for another operand path"); return SPQ_ERR_INVALID; }
  } else if (x3) {
    if (!((!a->quantize_input || (a->bits >= 1 && a->bits <= 24)) && a->x_limb_scale)) {
      set_error("spq_linear_lora_fwd: SPQ_PATH_F16X3 needs x_limb_scale (and 1..24 bits when quantize_input is set)");
      return SPQ_ERR_UNSUPPORTED;
    }
  } else if (!(a->quantize_input && a->qtype == SPQ_MINMAX && a->symmetric && a->bits >= 2 && a->bits <= 12)) {
    set_error("spq_linear_lora_fwd: SPQ_PATH_F16X2 needs a symmetric minmax input quantizer with 2..12 bits "
              "(got qtype=%d symmetric=%d bits=%d quantize_input=%d)", a->qtype, a->symmetric, a->bits, a->quantize_input);
    return SPQ_ERR_UNSUPPORTED;
  }
  if (!f16x2_shape_ok(a->M, a->K, a->N, a->r)) {
    set_error("spq_linear_lora_fwd: the F16 operand paths need LoRA rank <= 128 (got r=%lld)", (long long)a->r);
    return SPQ_ERR_UNSUPPORTED;
  }
  SPQ_REQUIRE(a->w_rowscale, "spq_linear_lora_fwd: w_rowscale missing for SPQ_PATH_F16X2");
  const F16x2Layout L = make_layout(a->M, a->K, a->r, a->N, x3 ? 2 : i8nl ? -1 : 1);
  const PrepLayout P = make_prep_layout(a->N, a->K, a->r);
  char* ws = (char*)a->workspace;
  const char* wp = (const char*)a->w_prep;

  XPassArgs x;
  x.x = a->x; x.sx = a->sx; x.zx = a->zx; x.aT = a->a_prep;
  x.qx = (_Float16*)(ws + L.off_qx); x.thi = (_Float16*)(ws + L.off_thi); x.tlo = (_Float16*)(ws + L.off_tlo);
  x.rowinv = (float*)(ws + L.off_rowinv);
  x.M = (int)a->M; x.K = (int)a->K; x.r = (int)a->r; x.Kp = (int)L.Kp; x.Rp = (int)L.Rp;
  x.x_pc = a->x_per_channel; x.bits = a->bits;
  x.limbs = x3 ? 1 : 0; x.qtype = a->qtype; x.symmetric = a->symmetric;
  if (x3 && !a->quantize_input) {   // identity quantizer: limbs of x itself; the scale loads read a valid dummy
    x.bits = 32; x.qtype = SPQ_MINMAX; x.symmetric = 1; x.x_pc = 0; x.sx = a->x_limb_scale; x.zx = a->x_limb_scale;
  }
  x.xl = (_Float16*)(ws + L.off_xl); x.xscale = a->x_limb_scale;
  x.t_out = (a->r > 0) ? a->t_out : nullptr;
  x.lora_fq = (a->lora_on_fq_input && a->quantize_input) ? 1 : 0;
  x.ln_w = a->ln_weight; x.ln_b = a->ln_bias; x.ln_eps = a->ln_eps;
  const bool lora_up = a->r > 0 && a->b_prep != nullptr;     // r > 0 without b_prep: LoRA-down only (t_out)
  if (a->path == SPQ_PATH_U8X2) {
    set_error("spq_linear_lora_fwd: SPQ_PATH_U8X2 (byte-level ring kernel) was measured slower than SPQ_PATH_F16X2 and is no longer in the library "
              "(tools/variants/gemm_u8x2.h); use SPQ_PATH_F16X2 -- same prepared operands, same result");
    return SPQ_ERR_UNSUPPORTED;
  }
  if (a->a_limb_scale) {
    set_error("spq_linear_lora_fwd: a_limb_scale (LoRA-down on the f16 pipe, tools/variants/xpass_panel16.h) is no longer in the library; pass null");
    return SPQ_ERR_UNSUPPORTED;
  }
  x.a8 = i8nl ? 2 : 0;
  const unsigned xgrid = (unsigned)((a->M + XR - 1) / XR);
  const bool panel_ok = (a->K % 64 == 0) && L.Rp <= 64 && aligned16(a->x) && (a->r == 0 || aligned16(a->a_prep)) && aligned16(x.sx) &&
                        aligned16(x.zx);
  const bool do_xpass = a->stage != SPQ_STAGE_CONTRACTION, do_gemm = a->stage != SPQ_STAGE_ACTIVATIONS;
  // a->prepare: the weight-side operands are (re)made by this call.  Where the streaming activation kernel runs, the row work
  // goes into ITS launch as extra workgroups (xpass_stream_prep_kernel; FQ(A)^T, which the pass itself consumes, goes first as a
  // launch of a few dozen workgroups; SPQ_PREP_ROLE=0 turns that off); otherwise the ordinary preparation launch is issued first
  // (SPQ_PREP_INPASS=1: the earlier variant that spreads the rows over the 16-row activation kernel's own workgroups).
  PrepArgs pa;
  int at_blocks = 0;
  const Switches sw = g_switches;
  // split-K of the contraction (f16 limb kernels): decided here because the activation launch zeroes its counters
  int sk = 1;
  const int64_t sk_nwg = (L.Mp / 128) * (P.Np / GN);
  if (!i8nl && do_gemm && a->epilogue == SPQ_EPILOGUE_NONE && !a->out_levels && (a->N & 3) == 0) {
    const int nl = lora_up ? (int)(L.Rp / GK) * 2 : 0;
    const int T = nl + (x3 ? 2 : 1) * (int)(L.Kp / GK);
    sk = t128_split(sk_nwg, T, nl, (int64_t)T128_WGS * gemm_grid(1 << 30), sw.split_k);
    if (sk > 1 && sk_nwg * sk > sk_reserve_units(a->M, a->K, a->N, a->r, x3 ? 2 : 1)) sk = 1;   // (cannot happen: the reserve is the maximum over the cases)
  }
  bool sk_zeroed = false;
  x.zero_ptr = nullptr; x.zero_n = 0;
  const int stream16 = sw.stream;
  const bool stream_ok = panel_ok && stream16 && a->r > 0 && (!x.limbs || (sw.stream_limbs && a->quantize_input)) && !x.lora_fq;
  bool role_prep = false;                                  // the row work as extra workgroups of the streaming activation launch
  if (a->prepare && do_xpass) {
    bool wave_ok = false;
    int prc = prepare_fill(a->prepare, pa, at_blocks, wave_ok);
    if (prc) return prc;
    if (a->prepare->N != a->N || a->prepare->K != a->K || a->prepare->r < a->r) { set_error("spq_linear_lora_fwd: prepare args describe another layer"); return SPQ_ERR_INVALID; }
    if (sw.prep_role && stream_ok && !a->ln_weight && wave_ok && a->K <= 1024) {
      role_prep = true;
      if (at_blocks) {
        fq_transpose_kernel<<<(unsigned)at_blocks, 256, 0, st>>>(pa);
        prc = check_launch("spq_linear_lora_fwd(FQ(A)^T)");
        if (prc) return prc;
      }
    } else {
      prc = spq_prepare_f16x2_args(a->prepare, (spq_stream_t)st);
      if (prc) return prc;
    }
  } else if (a->prepare && !do_xpass) {
    set_error("spq_linear_lora_fwd: prepare args go with the activation stage");
    return SPQ_ERR_INVALID;
  }
  if (a->ln_weight && do_xpass) {
    // the LayerNorm prologue lives in the panel kernels (fp32-MFMA LoRA-down), rows of at most 1024 elements
    if (!a->ln_bias || !panel_ok || a->K > 1024 || !aligned16(a->ln_weight) || !aligned16(a->ln_bias)) {
      set_error("spq_linear_lora_fwd: the LayerNorm prologue needs K %% 64 == 0, K <= 1024, rank <= 64 and 16-byte aligned operands");
      return SPQ_ERR_UNSUPPORTED;
    }
  }
  if (!do_xpass) {
    // the activation pass of this call ran earlier (same arguments, same workspace)
  } else if (panel_ok) {
    if (AttrOnce once(0); once.first) {
      hipError_t e = hipFuncSetAttribute((const void*)xpass_panel_kernel<XP_CHUNKS>, hipFuncAttributeMaxDynamicSharedMemorySize, XP_LDS);
      if (e != hipSuccess) { set_error("hipFuncSetAttribute(xpass LDS %d B): %s", XP_LDS, hipGetErrorString(e)); return SPQ_ERR_LAUNCH; }
      (void)hipFuncSetAttribute((const void*)xpass_panel_kernel<XP_CHUNKS_SMALL>, hipFuncAttributeMaxDynamicSharedMemorySize, xp_lds(XP_CHUNKS_SMALL));
    }
    if (stream_ok) {
      if (sk > 1) { x.zero_ptr = (int*)(ws + L.off_skcnt); x.zero_n = (int)(2 * sk_nwg); sk_zeroed = true; }
      if (AttrOnce once(7); once.first) {
#define SPQ_XS_ATTR(R, A8, LN) (void)hipFuncSetAttribute((const void*)xpass_stream_kernel<R, A8, LN>, hipFuncAttributeMaxDynamicSharedMemorySize, xs_lds(R))
        SPQ_XS_ATTR(16, 0, false); SPQ_XS_ATTR(16, 2, false); SPQ_XS_ATTR(16, 0, true); SPQ_XS_ATTR(16, 2, true);
        SPQ_XS_ATTR(32, 0, false); SPQ_XS_ATTR(32, 2, false); SPQ_XS_ATTR(32, 0, true); SPQ_XS_ATTR(32, 2, true);
        SPQ_XS_ATTR(16, 3, false); SPQ_XS_ATTR(32, 3, false); SPQ_XS_ATTR(16, 3, true); SPQ_XS_ATTR(32, 3, true);
#undef SPQ_XS_ATTR
      }
      // 32-row workgroups halve the FQ(A)^T traffic through L2; taken once they still cover every CU
      const bool r32 = stream16 == 32 || (stream16 != 16 && (a->M + 31) / 32 >= gemm_grid(1 << 30));
      const bool ln = x.ln_w != nullptr;
      if (role_prep) {
        if (AttrOnce once(8); once.first) {
#define SPQ_XSP_ATTR(R, A8, MODE) (void)hipFuncSetAttribute((const void*)xpass_stream_prep_kernel<R, A8, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, xs_lds(R))
          SPQ_XSP_ATTR(16, 0, 0); SPQ_XSP_ATTR(16, 2, 1); SPQ_XSP_ATTR(32, 0, 0); SPQ_XSP_ATTR(32, 2, 1); SPQ_XSP_ATTR(16, 3, 0); SPQ_XSP_ATTR(32, 3, 0);
#undef SPQ_XSP_ATTR
        }
        const int R = r32 ? 32 : 16;
        const unsigned nx = (unsigned)((a->M + R - 1) / R);
        const unsigned nrow = (unsigned)((pad_to(a->N, GN) + R / 4 - 1) / (R / 4));
#define SPQ_XSP_LAUNCH(R, A8, MODE) xpass_stream_prep_kernel<R, A8, MODE><<<nx + nrow, R * 16, xs_lds(R), st>>>(x, pa, (int)nx)
        if (x.limbs) { if (r32) SPQ_XSP_LAUNCH(32, 3, 0); else SPQ_XSP_LAUNCH(16, 3, 0); }
        else if (r32) { if (x.a8 == 0) SPQ_XSP_LAUNCH(32, 0, 0); else SPQ_XSP_LAUNCH(32, 2, 1); }
        else { if (x.a8 == 0) SPQ_XSP_LAUNCH(16, 0, 0); else SPQ_XSP_LAUNCH(16, 2, 1); }
#undef SPQ_XSP_LAUNCH
      } else {
#define SPQ_XS_LAUNCH(R, A8, LN) xpass_stream_kernel<R, A8, LN><<<(unsigned)((a->M + R - 1) / R), R * 16, xs_lds(R), st>>>(x)
#define SPQ_XS_PICK(R) do { if (ln) { if (x.a8 == 0) SPQ_XS_LAUNCH(R, 0, true); else SPQ_XS_LAUNCH(R, 2, true); } \
                            else { if (x.a8 == 0) SPQ_XS_LAUNCH(R, 0, false); else SPQ_XS_LAUNCH(R, 2, false); } } while (0)
        if (x.limbs && ln) { if (r32) SPQ_XS_LAUNCH(32, 3, true); else SPQ_XS_LAUNCH(16, 3, true); }
        else if (x.limbs) { if (r32) SPQ_XS_LAUNCH(32, 3, false); else SPQ_XS_LAUNCH(16, 3, false); }
        else if (r32) SPQ_XS_PICK(32); else SPQ_XS_PICK(16);
#undef SPQ_XS_PICK
#undef SPQ_XS_LAUNCH
      }
    } else if (xgrid >= 2 * gemm_grid(1 << 30)) xpass_panel_kernel<XP_CHUNKS_SMALL><<<xgrid, 512, xp_lds(XP_CHUNKS_SMALL), st>>>(x);
    else xpass_panel_kernel<XP_CHUNKS><<<xgrid, 512, XP_LDS, st>>>(x);
  } else if (L.Rp <= 64) xpass_kernel<2><<<xgrid, 256, 0, st>>>(x);
  else xpass_kernel<4><<<xgrid, 256, 0, st>>>(x);
  int rc = check_launch("spq_linear_lora_fwd(xpass)");
  if (rc || !do_gemm) return rc;

  if (i8nl) {
    const PrepLayoutI8 P8 = make_prep_layout_i8(a->N, a->K, a->r, i8nl);
    GemmI8Args q;
    q.qx = reinterpret_cast<const signed char*>(x.qx); q.W = reinterpret_cast<const signed char*>(wp); q.plane_stride = (int64_t)P8.plane;
    q.thi = x.thi; q.tlo = x.tlo;
    q.Bhi = (const _Float16*)(wp + P8.off_bhi); q.Blo = (const _Float16*)(wp + P8.off_blo);
    q.rowinv = x.rowinv; q.rowscale = a->w_rowscale; q.bscale = (const float*)(wp + P8.off_bscale); q.bias = a->bias; q.y = a->y;
    q.M = (int)a->M; q.N = (int)a->N; q.Kp = (int)L.Kp; q.Rp = lora_up ? (int)L.Rp : 0;
    q.tiles_m = (int)(L.Mp / GM); q.tiles_n = (int)(P8.Np / GN); q.epilogue = a->epilogue;
    // 128-deep stages (whole cache lines, half as many stages: measured 50-53 us against 56-59 at the c_fc shape) whenever K
    // allows; the 64-deep ring kernel otherwise
    const bool k128 = (L.Kp % 128) == 0;
    if (AttrOnce once(1); once.first) {
      hipError_t e = hipFuncSetAttribute((const void*)gemm_i8_kernel<1, 0, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, I8_LDS);
      if (e != hipSuccess) { set_error("hipFuncSetAttribute(LDS %d B): %s", I8_LDS, hipGetErrorString(e)); return SPQ_ERR_LAUNCH; }
      (void)hipFuncSetAttribute((const void*)gemm_i8_kernel<1, 0, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, I8_LDS);
      (void)hipFuncSetAttribute((const void*)gemm_i8_k128_kernel<1, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, i8k_lds(1));
      (void)hipFuncSetAttribute((const void*)gemm_i8_k128_kernel<1, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, i8k_lds(1));
    }
    const bool gelu8 = a->epilogue == SPQ_EPILOGUE_GELU;
    if (a->ev_gemm_begin) (void)hipEventRecord((hipEvent_t)a->ev_gemm_begin, st);
    if (k128) {
      const int ntiles = 2 * q.tiles_m * q.tiles_n;
      const unsigned cus2 = 2 * gemm_grid(1 << 30);
      const unsigned g128 = (unsigned)ntiles < cus2 ? (unsigned)ntiles : cus2;
      if (gelu8) gemm_i8_k128_kernel<1, 1><<<g128, 256, i8k_lds(1), st>>>(q);
      else gemm_i8_k128_kernel<1, 0><<<g128, 256, i8k_lds(1), st>>>(q);
    } else {
      const unsigned grid8 = gemm_grid(q.tiles_m * q.tiles_n);
      if (gelu8) gemm_i8_kernel<1, 0, 1><<<grid8, I8_THREADS, I8_LDS, st>>>(q);
      else gemm_i8_kernel<1, 0, 0><<<grid8, I8_THREADS, I8_LDS, st>>>(q);
    }
    if (a->ev_gemm_end) (void)hipEventRecord((hipEvent_t)a->ev_gemm_end, st);
    return check_launch("spq_linear_lora_fwd(gemm_i8)");
  }
  GemmF16Args g;
  g.qx = x.qx; g.thi = x.thi; g.tlo = x.tlo;
  g.Whi = (const _Float16*)(wp + P.off_whi); g.Wlo = (const _Float16*)(wp + P.off_wlo);
  g.Bhi = (const _Float16*)(wp + P.off_bhi); g.Blo = (const _Float16*)(wp + P.off_blo);
  g.rowinv = x.rowinv; g.rowscale = a->w_rowscale; g.bias = a->bias; g.y = a->y;
  g.M = (int)a->M; g.N = (int)a->N; g.Kp = (int)L.Kp; g.Rp = lora_up ? (int)L.Rp : 0;
  g.tiles_m = (int)(L.Mp / GM); g.tiles_n = (int)(P.Np / GN); g.dbg = nullptr;
  g.xl = x.xl; g.xscale = a->x_limb_scale; g.a_limbs = x3 ? 2 : 1;
  g.lv = (_Float16*)a->out_levels; g.lv_scale = a->out_scale; g.lv_ld = (int)a->out_levels_ld; g.lv_pc = a->out_scale_per_channel;
  g.lv_qhi = 0.f;
  g.lv_lo = (_Float16*)a->out_levels_lo; g.lv_zero = a->out_zero; g.lv_xscale = a->out_limb_scale;
  g.lv_qtype = a->out_qtype; g.lv_sym = a->out_symmetric; g.lv_bits = a->out_bits;
  if (a->out_levels) {
    const bool limbs_out = a->out_levels_lo != nullptr;
    const bool common = a->out_scale && (a->N % 64) == 0 && a->out_levels_ld >= a->N && (a->out_levels_ld % 4) == 0 && aligned16(a->out_levels) &&
                        aligned16(a->out_scale);
    const bool ok = limbs_out ? (common && a->out_zero && a->out_limb_scale && a->out_bits >= 1 && a->out_bits <= 24 && aligned16(a->out_levels_lo) &&
                                 aligned16(a->out_zero) && (a->out_qtype == SPQ_MINMAX || a->out_qtype == SPQ_LOG || a->out_qtype == SPQ_LOG_DIRECT))
                              : (common && a->out_bits >= 2 && a->out_bits <= 12);
    if (!ok) {
      set_error("spq_linear_lora_fwd: levels-out store needs out_scale, N %% 64 == 0, a 16-B aligned level matrix with row pitch >= N, and 2..12 "
                "out_bits (levels) or out_zero + out_limb_scale + 1..24 out_bits (limbs) (got bits=%d N=%lld ld=%lld limbs=%d)", a->out_bits,
                (long long)a->N, (long long)a->out_levels_ld, (int)limbs_out);
      return SPQ_ERR_UNSUPPORTED;
    }
    if (!limbs_out) g.lv_qhi = (float)((1 << (a->out_bits - 1)) - 1);
  }
  if (AttrOnce once(3); once.first) {
    hipError_t e = hipFuncSetAttribute((const void*)gemm_f16x2_t128_kernel<1, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, T128_LDS);
    if (e != hipSuccess) { set_error("hipFuncSetAttribute(LDS %d B): %s", T128_LDS, hipGetErrorString(e)); return SPQ_ERR_LAUNCH; }
#define SPQ_T128_ATTR(AL, EPI, LV) (void)hipFuncSetAttribute((const void*)gemm_f16x2_t128_kernel<AL, EPI, LV>, hipFuncAttributeMaxDynamicSharedMemorySize, T128_LDS)
    SPQ_T128_ATTR(2, 0, 0); SPQ_T128_ATTR(1, 1, 0); SPQ_T128_ATTR(2, 1, 0);
    SPQ_T128_ATTR(1, 0, 1); SPQ_T128_ATTR(2, 0, 1); SPQ_T128_ATTR(1, 1, 1); SPQ_T128_ATTR(2, 1, 1);
    SPQ_T128_ATTR(1, 0, 2); SPQ_T128_ATTR(2, 0, 2); SPQ_T128_ATTR(1, 1, 2); SPQ_T128_ATTR(2, 1, 2);
#undef SPQ_T128_ATTR
    (void)hipFuncSetAttribute((const void*)gemm_f16x2_t128_kernel<1, 0, 0, true>, hipFuncAttributeMaxDynamicSharedMemorySize, T128_LDS);
    (void)hipFuncSetAttribute((const void*)gemm_f16x2_t128_kernel<2, 0, 0, true>, hipFuncAttributeMaxDynamicSharedMemorySize, T128_LDS);
    (void)hipFuncSetAttribute((const void*)gemm_f16x2_t128_kernel<1, 0, 0, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, T128_LDS);
    (void)hipFuncSetAttribute((const void*)gemm_f16x2_t128_kernel<2, 0, 0, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, T128_LDS);
  }
  g.split = sk; g.sk_cnt = (int*)(ws + L.off_skcnt); g.sk_part = (float*)(ws + L.off_skpart);
  if (sk > 1 && !sk_zeroed && hipMemsetAsync(g.sk_cnt, 0, (size_t)(2 * sk_nwg) * sizeof(int), st) != hipSuccess) {
    set_error("spq_linear_lora_fwd: hipMemsetAsync(split-K counters) failed"); (void)hipGetLastError(); return SPQ_ERR_LAUNCH;
  }
  if (a->ev_gemm_begin) (void)hipEventRecord((hipEvent_t)a->ev_gemm_begin, st);
  // ONE contraction kernel for every shape (128 x 128 tiles, three workgroups per CU); the 256 x 128 persistent kernels of round 1
  // live in tools/variants/gemm_256x128.h (measured slower at every shape of SURVEY 8(d): config 3 1.466 -> 1.356 ms, config 5
  // 27.8 -> 27.4 ms)
  const bool gelu = a->epilogue == SPQ_EPILOGUE_GELU;
  const bool ragged = (a->N & 3) != 0;                     // rows of y not 16-B aligned: the scalar-store instantiation (plain epilogue)
  if (ragged && (gelu || g.lv)) { set_error("spq_linear_lora_fwd: the fused GELU / levels-out epilogues need N %% 4 == 0 (got N=%lld)", (long long)a->N); return SPQ_ERR_UNSUPPORTED; }
  const int ntiles = 2 * g.tiles_m * g.tiles_n * sk;
  const unsigned cus3 = (sk > 1 ? 2 : 1) * T128_WGS * gemm_grid(1 << 30);    // split-K: one unit per workgroup
  const unsigned grid128 = (unsigned)ntiles < cus3 ? (unsigned)ntiles : cus3;
#define SPQ_T128_LAUNCH(AL, EPI, LV) gemm_f16x2_t128_kernel<AL, EPI, LV><<<grid128, 256, T128_LDS, st>>>(g)
#define SPQ_T128_PICK(LV) do { if (x3 && gelu) SPQ_T128_LAUNCH(2, 1, LV); else if (x3) SPQ_T128_LAUNCH(2, 0, LV); \
                               else if (gelu) SPQ_T128_LAUNCH(1, 1, LV); else SPQ_T128_LAUNCH(1, 0, LV); } while (0)
  if (g.lv && g.lv_lo) SPQ_T128_PICK(2);
  else if (g.lv) SPQ_T128_PICK(1);
  else if (ragged) { if (x3) gemm_f16x2_t128_kernel<2, 0, 0, false, true><<<grid128, 256, T128_LDS, st>>>(g); else gemm_f16x2_t128_kernel<1, 0, 0, false, true><<<grid128, 256, T128_LDS, st>>>(g); }
  else if (sk > 1) { if (x3) gemm_f16x2_t128_kernel<2, 0, 0, true><<<grid128, 256, T128_LDS, st>>>(g); else gemm_f16x2_t128_kernel<1, 0, 0, true><<<grid128, 256, T128_LDS, st>>>(g); }
  else SPQ_T128_PICK(0);
#undef SPQ_T128_PICK
#undef SPQ_T128_LAUNCH
  if (a->ev_gemm_end) (void)hipEventRecord((hipEvent_t)a->ev_gemm_end, st);
  return check_launch("spq_linear_lora_fwd(gemm_t128)");
}

}  // namespace spq

using namespace spq;

extern "C" size_t spq_prep_f16x2_bytes(int64_t N, int64_t K, int64_t r) {
  if (N <= 0 || K <= 0 || r < 0) return 0;
  return make_prep_layout(N, K, r).total;
}

extern "C" size_t spq_prep_bytes(int64_t N, int64_t K, int64_t r, int path) {
  if (N <= 0 || K <= 0 || r < 0) return 0;
  const int nl = i8_limbs_of(path);
  return nl ? make_prep_layout_i8(N, K, r, nl).total : make_prep_layout(N, K, r).total;
}

namespace spq {
// validation + PrepArgs of one spq_prepare_args; at_blocks = FQ(A)^T tiles, wave_ok = the one-wave-per-row kernel applies
int prepare_fill(const spq_prepare_args* q, PrepArgs& a, int& at_blocks, bool& wave_ok) {
  SPQ_REQUIRE(q, "spq_prepare_f16x2: null args");
  SPQ_REQUIRE(q->W && q->sw && q->zw && q->sx && q->w_prep && q->w_rowscale, "spq_prepare_f16x2: null pointer");
  SPQ_REQUIRE(q->N > 0 && q->K > 0 && q->r >= 0, "spq_prepare_f16x2: bad shape");
  SPQ_REQUIRE(q->r == 0 || (q->B && q->sb && q->zb), "spq_prepare_f16x2: LoRA operands missing");
  SPQ_REQUIRE(!q->A || (q->sa && q->za && q->a_prep), "spq_prepare_f16x2: LoRA-A quantizer parameters / output missing");
  SPQ_REQUIRE(q->w_bits >= 1 && q->b_bits >= 0 && q->a_bits >= 0, "spq_prepare_f16x2: bad bit-width");
  const int64_t N = q->N, K = q->K, r = q->r;
  if (!f16x2_shape_ok(1, K, N, r)) { set_error("spq_prepare_f16x2: needs LoRA rank <= 128 (got r=%lld)", (long long)r); return SPQ_ERR_UNSUPPORTED; }
  const int nl = i8_limbs_of(q->path);
  SPQ_REQUIRE(nl || q->path == 0 || q->path == SPQ_PATH_F16X2 || q->path == SPQ_PATH_U8X2 || q->path == SPQ_PATH_F16X3,
              "spq_prepare_f16x2: unknown operand path %d", q->path);
  const PrepLayout P = make_prep_layout(N, K, r);
  const PrepLayoutI8 P8 = make_prep_layout_i8(N, K, r, nl);
  const size_t need = nl ? P8.total : P.total;
  if (q->w_prep_bytes < need || !aligned16(q->w_prep)) { set_error("spq_prepare_f16x2: buffer too small (%zu < %zu)", q->w_prep_bytes, need); return SPQ_ERR_WORKSPACE; }
  char* wp = (char*)q->w_prep;
  a.W = q->W; a.sw = q->sw; a.zw = q->zw; a.B = r > 0 ? q->B : nullptr; a.sb = q->sb; a.zb = q->zb; a.sx = q->sx;
  a.nl = nl; a.W8 = nullptr; a.plane_stride = 0; a.bscale = nullptr;
  if (nl) {
    if (!(q->w_qtype == SPQ_MINMAX && q->w_symmetric && q->w_bits >= 2 && q->w_bits <= 8 && !q->x_per_channel)) {
      set_error("spq_prepare_f16x2: SPQ_PATH_I8 needs symmetric minmax weights of <= 8 bits and a per-tensor input scale");
      return SPQ_ERR_UNSUPPORTED;
    }
    a.Whi = a.Wlo = nullptr;
    a.W8 = (signed char*)wp; a.plane_stride = (int64_t)P8.plane;
    a.Bhi = P8.Rp ? (_Float16*)(wp + P8.off_bhi) : nullptr; a.Blo = P8.Rp ? (_Float16*)(wp + P8.off_blo) : nullptr;
    a.bscale = (float*)(wp + P8.off_bscale);
  } else {
    a.Whi = (_Float16*)(wp + P.off_whi); a.Wlo = (_Float16*)(wp + P.off_wlo);
    a.Bhi = P.Rp ? (_Float16*)(wp + P.off_bhi) : nullptr; a.Blo = P.Rp ? (_Float16*)(wp + P.off_blo) : nullptr;
  }
  a.rowscale = q->w_rowscale;
  a.N = (int)N; a.K = (int)K; a.r = (int)r; a.Kp = (int)P.Kp; a.Rp = (int)P.Rp;
  a.w_pc = q->w_per_channel; a.w_bits = q->w_bits; a.w_qtype = q->w_qtype; a.w_sym = q->w_symmetric;
  a.b_pc = q->b_per_channel; a.b_bits = q->b_bits; a.b_qtype = q->b_qtype; a.b_sym = q->b_symmetric;
  a.x_pc = q->x_per_channel; a.scaling = q->scaling;
  a.A = q->A; a.sa = q->sa; a.za = q->za; a.aT = q->a_prep;
  a.a_pc = q->a_per_channel; a.a_bits = q->a_bits; a.a_qtype = q->a_qtype; a.a_sym = q->a_symmetric;
  at_blocks = (r > 0 && q->A) ? (int)(((r + 31) / 32) * ((K + 31) / 32)) : 0;   // A == NULL: FQ(A)^T is made elsewhere
  wave_ok = (K % 4 == 0) && K <= 4 * 64 * PREP_MAXI && aligned16(q->W) && (!q->x_per_channel || aligned16(q->sx)) && r <= 128;
  a.row_blocks = wave_ok ? (int)(P.Np / 4) : (int)P.Np;
  if (nl && !wave_ok) {
    set_error("spq_prepare_f16x2: the int8 operand paths need K %% 4 == 0, K <= %d and 16-byte aligned W / sx", 4 * 64 * PREP_MAXI);
    return SPQ_ERR_UNSUPPORTED;
  }
  return SPQ_OK;
}
}  // namespace spq

extern "C" int spq_prepare_f16x2_args(const spq_prepare_args* q, spq_stream_t stream) {
  PrepArgs a;
  int at_blocks = 0;
  bool wave_ok = false;
  const int rc = prepare_fill(q, a, at_blocks, wave_ok);
  if (rc) return rc;
  const unsigned grid = (unsigned)(a.row_blocks + at_blocks);
  const bool short_rows = q->K <= 4 * 64 * 4;
  if (a.nl == 1 && short_rows) prep_wave_kernel<1, 4><<<grid, 256, 0, (hipStream_t)stream>>>(a, at_blocks);
  else if (a.nl == 1) prep_wave_kernel<1><<<grid, 256, 0, (hipStream_t)stream>>>(a, at_blocks);
  else if (wave_ok && short_rows) prep_wave_kernel<0, 4><<<grid, 256, 0, (hipStream_t)stream>>>(a, at_blocks);
  else if (wave_ok) prep_wave_kernel<0><<<grid, 256, 0, (hipStream_t)stream>>>(a, at_blocks);
  else prep_f16x2_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(a);
  return check_launch("spq_prepare_f16x2");
}

extern "C" int spq_prepare_f16x2(const float* W, int64_t N, int64_t K, const float* sw, const float* zw,
                                 int w_per_channel, int w_bits, int w_qtype, int w_symmetric, const float* B,
                                 int64_t r, const float* sb, const float* zb, int b_per_channel, int b_bits,
                                 int b_qtype, int b_symmetric, float scaling, const float* A, const float* sa,
                                 const float* za, int a_per_channel, int a_bits, int a_qtype, int a_symmetric,
                                 const float* sx, int x_per_channel, void* w_prep, size_t w_prep_bytes,
                                 float* w_rowscale, float* a_prep, spq_stream_t stream) {
  spq_prepare_args q;
  q.W = W; q.N = N; q.K = K; q.sw = sw; q.zw = zw; q.w_per_channel = w_per_channel; q.w_bits = w_bits; q.w_qtype = w_qtype;
  q.w_symmetric = w_symmetric; q.B = B; q.r = r; q.sb = sb; q.zb = zb; q.b_per_channel = b_per_channel; q.b_bits = b_bits;
  q.b_qtype = b_qtype; q.b_symmetric = b_symmetric; q.scaling = scaling; q.A = A; q.sa = sa; q.za = za;
  q.a_per_channel = a_per_channel; q.a_bits = a_bits; q.a_qtype = a_qtype; q.a_symmetric = a_symmetric; q.sx = sx;
  q.x_per_channel = x_per_channel; q.w_prep = w_prep; q.w_prep_bytes = w_prep_bytes; q.w_rowscale = w_rowscale; q.a_prep = a_prep;
  q.path = SPQ_PATH_F16X2;
  return spq_prepare_f16x2_args(&q, stream);
}

extern "C" int spq_prepare_cpt(const float* W, int64_t N, int64_t K, const float* sw, const float* zw, int w_per_channel,
                               int w_bits, int w_qtype, int w_symmetric, const float* A, const float* B, int64_t r,
                               const float* sl, const float* zl, int l_per_channel, int l_bits, int l_qtype,
                               int l_symmetric, float scaling, const float* sx, int x_per_channel, int path,
                               void* w_prep, size_t w_prep_bytes, float* w_rowscale, float* w_eff, float* aq, float* bq,
                               float* aq_t, spq_stream_t stream) {
  SPQ_REQUIRE(W && sw && zw && w_eff, "spq_prepare_cpt: null pointer");
  SPQ_REQUIRE(N > 0 && K > 0 && r >= 0 && r <= CPT_RMAX, "spq_prepare_cpt: bad shape N=%lld K=%lld r=%lld (rank <= %d)",
              (long long)N, (long long)K, (long long)r, CPT_RMAX);
  SPQ_REQUIRE(r == 0 || (A && B && sl && zl && aq && bq), "spq_prepare_cpt: LoRA operands missing");
  SPQ_REQUIRE(N < (1 << 30) && K < (1 << 30), "spq_prepare_cpt: dimension too large");
  hipStream_t st = (hipStream_t)stream;
  CptArgs c;
  c.W = W; c.sw = sw; c.zw = zw; c.A = A; c.B = B; c.sl = sl; c.zl = zl; c.aq = aq; c.bq = bq; c.aqT = r > 0 ? aq_t : nullptr;
  c.w_eff = w_eff; c.N = (int)N; c.K = (int)K; c.r = (int)r;
  c.w_pc = w_per_channel; c.w_bits = w_bits; c.w_qtype = w_qtype; c.w_sym = w_symmetric;
  c.l_pc = l_per_channel; c.l_bits = l_bits; c.l_qtype = l_qtype; c.l_sym = l_symmetric;
  c.scaling = scaling;
  if (r > 0) {
    const int64_t total = (K + N) * r + (aq_t ? ((r + 63) / 64 * 64 - r) * K : 0);
    cpt_factors_kernel<<<(unsigned)std::min<int64_t>((total + 255) / 256, 2048), 256, 0, st>>>(c);
  }
  dim3 grid((unsigned)((K + 63) / 64), (unsigned)((N + 63) / 64));
  cpt_weff_kernel<<<grid, 256, 0, st>>>(c);
  int rc = check_launch("spq_prepare_cpt");
  if (rc || path == SPQ_PATH_F32) return rc;
  SPQ_REQUIRE(path == SPQ_PATH_F16X2 || path == SPQ_PATH_F16X3, "spq_prepare_cpt: unknown operand path %d", path);
  SPQ_REQUIRE(sx && w_prep && w_rowscale, "spq_prepare_cpt: limb buffers missing");
  // the limb split of W_eff: identity weight quantizer (32 bit), input scale folded in for SPQ_PATH_F16X2
  return spq_prepare_f16x2(w_eff, N, K, sx /* any valid scalar */, sx, 0, 32, SPQ_MINMAX, 1, nullptr, 0, nullptr, nullptr, 0, 0,
                           0, 1, 0.f, nullptr, nullptr, nullptr, 0, 0, 0, 1, sx, x_per_channel, w_prep, w_prep_bytes,
                           w_rowscale, nullptr, stream);
}

