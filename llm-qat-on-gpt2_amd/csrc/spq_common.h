// Shared device helpers and host-side error plumbing for libspq (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/spq.h"

namespace spq {

// ---- host: thread-local error message ----------------------------------------------------------
void set_error(const char* fmt, ...);
int check_launch(const char* what);

#define SPQ_REQUIRE(cond, ...)            \
  do {                                    \
    if (!(cond)) {                        \
      ::spq::set_error(__VA_ARGS__);      \
      return SPQ_ERR_INVALID;             \
    }                                     \
  } while (0)

__host__ __device__ static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// ---- device: the quantization arithmetic, op for op as the reference --------------------------------
// This translation unit set is compiled with -ffp-contract=off: every multiply, add and divide below is
// its own IEEE-754 round-to-nearest operation, like the ATen CPU kernels the reference runs on.

// torch.round == round-half-to-even == rintf under the default rounding mode.
// torch.clamp / min / max propagate NaN; fminf / fmaxf drop it.  clampf keeps a NaN input (one compare + select), so a NaN
// activation becomes a NaN level and a NaN output row, as in the reference.
__device__ __forceinline__ float clampf(float v, float lo, float hi) {
  const float r = fminf(fmaxf(v, lo), hi);
  return (v != v) ? v : r;
}
__device__ __forceinline__ float nan_min(float a, float b) { return (a != a || b != b) ? (a + b) : fminf(a, b); }
__device__ __forceinline__ float nan_max(float a, float b) { return (a != a || b != b) ? (a + b) : fmaxf(a, b); }

// quantization_methods.py:14-15 / :18-19 -- integer level (held in fp32)
template <bool SYM>
__device__ __forceinline__ float minmax_level(float x, float scale, float zp, float qlo, float qhi) {
  float q = SYM ? rintf(x / scale) : rintf(x / scale + zp);
  return clampf(q, qlo, qhi);
}
// The same level (symmetric form) without the IEEE division, for the streaming activation pass where the division's ~10 VALU
// instructions per element were the kernel's issue bound.  q = x * rs with rs = v_rcp_f32(scale) (1 ulp), r = rint(q).
//   |q - x/s| <= (2^-23 + 2^-24) |x/s| and |fl(x/s) - x/s| <= 2^-24 |x/s|, so q and the reference's quotient differ by less than
//   2^-22 |q|.  rint of the two can differ only if a tie k + 1/2 lies within that distance of q, i.e. only if
//   |q - r| + 2^-21 |q| >= 1/2 (|q - r| is exact; the test has a 2x margin over the bound and over its own rounding).
// `unsafe` is raised then -- and for NaN / Inf (x or rs: a zero, subnormal or NaN scale) since the comparison is written to fail
// on NaN -- and the caller recomputes the element with minmax_level<true>.  A scale above 2^126, whose reciprocal is subnormal,
// is the caller's to exclude (one compare per scale, not per element).  Returns r unclamped.
__device__ __forceinline__ float minmax_level_fast(float x, float rs, bool& unsafe) {
  const float q = x * rs;
  const float r = rintf(q);
  const float t = __builtin_fmaf(fabsf(q), 0x1p-21f, fabsf(q - r));
  unsafe |= !(t < 0.5f);
  return r;
}
// :16 / :20
template <bool SYM>
__device__ __forceinline__ float minmax_dequant(float q, float scale, float zp) {
  return SYM ? q * scale : (q - zp) * scale;
}

// fp32 log2 / exp2 via fp64 (correctly rounded in all but ~1e-7 of cases); ATen's CPU log2f/powf are
// <=1-ulp SLEEF kernels, so a last-bit difference is possible there, see DESIGN.md "log path".  Used for the calibration
// statistics and for level decisions near a rounding tie (log_level).
__device__ __forceinline__ float log2_rn(float v) { return (float)log2((double)v); }
__device__ __forceinline__ float exp2_rn(float v) { return (float)exp2((double)v); }

struct LogParams {
  float n2;     // 2*n (sym) or n (asym): the multiplier before rounding
  float qlo, qhi;
  float full;   // 2^b - 1
  float denom;  // sym: 2*n ; asym: n
};

// direct: part2's log quantizer (part2_cyclic_precision_training/quantization_methods.py:36-40) dequantises
// q/(2n) + 0.5 as is; part1's (quantization_methods.py:57,:64) sends it through * (2^b-1) / (2^b-1) first.
__host__ __device__ inline LogParams make_log_params(int bits, bool sym, bool direct = false) {
  LogParams p;
  if (sym) {
    float n = (float)((1 << (bits - 1)) - 1);
    p.n2 = n;  // applied as (c*2)*n like the reference, see log_level
    p.qlo = -n; p.qhi = n;
    p.denom = 2.0f * n;
  } else {
    float n = (float)((1u << bits) - 1u);
    p.n2 = n; p.qlo = 0.f; p.qhi = n; p.denom = n;
  }
  p.full = direct ? 0.f : (float)((1u << bits) - 1u);
  return p;
}

// quantization_methods.py:45-61 -> integer level, given log2|x|
template <bool SYM>
__device__ __forceinline__ float log_pre_round(float lg, float log_min, float log_range, const LogParams& p) {
  const float eps = 1e-5f;                                         // :35 (hard-coded)
  float ln = (lg - log_min) / fmaxf(log_range, eps);               // :49
  ln = clampf(ln, 0.f, 1.f);                                       // :50
  if (SYM) return ((ln - 0.5f) * 2.0f) * p.n2;                     // :54-55  centered * 2 * n_levels
  return ln * p.n2;                                                // :60
}

// The level is decided by a rounding, so log2 must be the correctly rounded one wherever the pre-round value sits near a
// tie; everywhere else the hardware log2 (v_log_f32, <= 1 ulp) gives the same level.  Fast path + checked fallback: the
// fp64 log2 runs for ~1e-4 of the elements instead of all of them (it was 70 % of the log fake-quant time).
template <bool SYM>
__device__ __forceinline__ float log_level(float x, float log_min, float log_range, const LogParams& p) {
  const float eps = 1e-5f;
  const float mag = fmaxf(fabsf(x), eps);                          // :45
  const float lg_fast = __builtin_amdgcn_logf(mag);                // :47, approximately
  // fast pre-round value: the division of :49 as a multiplication by v_rcp_f32(range) (the two IEEE divisions per element --
  // this one and the one in the error bound's slope -- were a fifth of the log fake-quant's instructions); only rint(pre)
  // is used, and the tie test below sends every element whose rounding could differ through the exact chain
  const float rr = __builtin_amdgcn_rcpf(fmaxf(log_range, eps));
  const float k2 = (SYM ? 2.0f : 1.0f) * p.n2;
  const float lnf = clampf((lg_fast - log_min) * rr, 0.f, 1.f);
  float pre = SYM ? ((lnf - 0.5f) * 2.0f) * p.n2 : lnf * p.n2;
  // |lg_fast - lg| <= 2 ulp, propagated through the slope of pre(lg); (lg - min) * rr against (lg - min) / range: <= 2.5 ulp of a
  // value that the clamp keeps within [0, 1]; plus the roundings of the chain itself
  const float slope = k2 * rr;
  const float err = 2.4e-7f * fmaxf(fabsf(lg_fast), 1.f) * slope + 3.0e-7f * k2 + 1e-6f * fmaxf(fabsf(pre), 1.f);
  const float tie_dist = 0.5f - fabsf(pre - rintf(pre));
  if (!(tie_dist > err)) pre = log_pre_round<SYM>(log2_rn(mag), log_min, log_range, p);   // also catches NaN
  return (x != x) ? x : clampf(rintf(pre), p.qlo, p.qhi);          // :55-56 / :60-61 (a NaN input stays NaN, as torch.clamp keeps it)
}
// :57,:64,:66 -> the normalised level: a function of the integer level alone (two IEEE divisions)
template <bool SYM>
__device__ __forceinline__ float log_qn(float q, const LogParams& p) {
  float qn;
  if (SYM) {
    qn = q / p.denom + 0.5f;                                       // :57
    if (p.full > 0.f) qn = (qn * p.full) / p.full;                 // :64 (part1 only)
  }
  else     qn = q / p.denom;                                       // :66
  return qn;
}
// :57,:64-74 -> dequantised value.  qn_lut (nullable): log_qn of every level qlo .. qhi, tabulated by the caller with log_qn
// itself (so the same numbers) -- the activation kernels keep it in LDS for widths of at most 8 bits, where the two divisions
// per ELEMENT were a third of the log fake-quant's instructions
template <bool SYM>
__device__ __forceinline__ float log_dequant(float x, float q, float log_min, float log_range,
                                             const LogParams& p, const float* qn_lut = nullptr) {
  const float qn = qn_lut ? qn_lut[(q == q) ? (int)(q - p.qlo) : 0] : log_qn<SYM>(q, p);   // (a NaN level: the sign below is NaN anyway)
  float x_hat = qn * log_range + log_min;                          // :68 (unclamped range)
  float mag = __builtin_amdgcn_exp2f(x_hat);                       // :70  v_exp_f32 (<= 1 ulp, like ATen's pow); |x_hat| < 64 here
  float sgn = (x > 0.f) ? 1.f : ((x < 0.f) ? -1.f : ((x == 0.f) ? 0.f : x));   // :43 torch.sign (NaN -> NaN)
  float out = mag * sgn;                                           // :72
  return (fabsf(x) < 1e-5f) ? 0.f : out;                           // :41,:74
}

// 64-lane butterfly reductions (wavefront = 64 on gfx950)
__device__ __forceinline__ float wave_min(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
// NaN-propagating twins for the calibration statistics (torch's min / max reductions propagate NaN)
__device__ __forceinline__ float wave_min_nan(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = nan_min(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ float wave_max_nan(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = nan_max(v, __shfl_xor(v, o, 64));
  return v;
}


// SwitchableLayerNorm row statistics (switchable_batchnorm.py:102-109): mean, then the mean of squared deviations as
// x.var(unbiased=False) defines it, over a row held in registers (NV float4 per lane, element c = (64 i + lane) * 4).
// Shared by layernorm_kernel and by the activation pass that applies the LayerNorm on the fly, so both produce the same bits.
// Extra iterations (c >= cols) add nothing: any NV that covers the row gives the same result.
// ln_row_load: the row's pieces into registers, every load issued before anything consumes one (branch-free: a guarded load
// made the compiler wait for each piece at the join, i.e. NV dependent memory round trips per row).  ln_row_reduce: the
// statistics of a loaded row.  ln_row_stats = both.
template <int NV>
__device__ __forceinline__ void ln_row_load(const float* __restrict__ xr, int cols, int lane, float4 (&v)[NV]) {
  const int last = (cols >> 2) - 1;                         // cols % 4 == 0
#pragma unroll
  for (int i = 0; i < NV; ++i) v[i] = *reinterpret_cast<const float4*>(xr + 4 * min(i * 64 + lane, last));
}
template <int NV>
__device__ __forceinline__ void ln_row_reduce(const float4 (&v)[NV], int cols, float eps, int lane, float& mean, float& den) {
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = (i * 64 + lane) * 4;
    if (c < cols) sum += (v[i].x + v[i].y) + (v[i].z + v[i].w);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
  mean = sum / (float)cols;
  float sq = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = (i * 64 + lane) * 4;
    if (c < cols) {
      const float dx = v[i].x - mean, dy = v[i].y - mean, dz = v[i].z - mean, dw = v[i].w - mean;
      sq += (dx * dx + dy * dy) + (dz * dz + dw * dw);
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) sq += __shfl_xor(sq, o, 64);
  den = sqrtf(sq / (float)cols + eps);           // torch.sqrt(var + eps)
}
template <int NV>
__device__ __forceinline__ void ln_row_stats(const float* __restrict__ xr, int cols, float eps, int lane, float4 (&v)[NV],
                                             float& mean, float& den) {
  ln_row_load<NV>(xr, cols, lane, v);
  ln_row_reduce<NV>(v, cols, eps, lane, mean, den);
}
// one normalised element: weight * ((x - mean) / den) + bias, each operation its own rounding (-ffp-contract=off)
__device__ __forceinline__ float ln_apply(float x, float mean, float den, float w, float b) { return w * ((x - mean) / den) + b; }

}  // namespace spq
