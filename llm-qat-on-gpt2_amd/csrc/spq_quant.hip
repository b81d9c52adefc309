// Calibration statistics, scale derivation and standalone quantize-dequantize kernels (HBM-bound scans).
// Compiled with -ffp-contract=off; see spq_common.h for the arithmetic.
#include <algorithm>

#include "spq_common.h"

namespace spq {

// =================================================================================================
// Statistics: x viewed as [outer, chan, inner].
//   rows kernel : one wavefront per (outer, chan) row of `inner` contiguous floats  -> part[outer*chan]
//   cols kernel : [rows, cols] matrix, per-column min/max over a slab of rows       -> part[slab][cols]
//   merge kernel: reduce partials over the leading axis, optional log2 transform, running update
// ABS=true reduces |x| (log domain: log2 is monotone, so min/max commute with it).
// =================================================================================================
constexpr int kStatsBlock = 256;
constexpr int kMaxSlabs = 256;

// torch's min / max reductions propagate NaN (a diverged activation must surface as a NaN scale, not a plausible finite one);
// v_min / v_max drop it.  The scan keeps a per-lane flag (one compare per element) and poisons its partial at the end; every
// combine after that is NaN-propagating.
template <bool ABS>
__device__ __forceinline__ void acc_minmax(float v, float& lo, float& hi, bool& nan) {
  if (ABS) v = fabsf(v);
  nan |= (v != v);
  lo = fminf(lo, v);
  hi = fmaxf(hi, v);
}
__device__ __forceinline__ void poison(bool nan, float& lo, float& hi) {
  if (nan) { lo = __builtin_nanf(""); hi = lo; }
}

template <bool ABS>
__global__ __launch_bounds__(kStatsBlock) void stats_rows_kernel(const float* __restrict__ x, int64_t nrows,
                                                                 int64_t len, float* __restrict__ pmin,
                                                                 float* __restrict__ pmax) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * (kStatsBlock / 64) + (threadIdx.x >> 6);
  if (row >= nrows) return;
  const float* p = x + row * len;
  float lo = INFINITY, hi = -INFINITY;
  bool nan = false;
  if ((len & 3) == 0 && aligned16(p)) {
    const float4* p4 = reinterpret_cast<const float4*>(p);
    const int64_t n4 = len >> 2;
    for (int64_t i = lane; i < n4; i += 64) {
      float4 v = p4[i];
      acc_minmax<ABS>(v.x, lo, hi, nan); acc_minmax<ABS>(v.y, lo, hi, nan);
      acc_minmax<ABS>(v.z, lo, hi, nan); acc_minmax<ABS>(v.w, lo, hi, nan);
    }
  } else {
    for (int64_t i = lane; i < len; i += 64) acc_minmax<ABS>(p[i], lo, hi, nan);
  }
  poison(nan, lo, hi);
  lo = wave_min_nan(lo);
  hi = wave_max_nan(hi);
  if (lane == 0) { pmin[row] = lo; pmax[row] = hi; }
}

// grid = (ceil(cols / (64*4)) or ceil(cols/64), slabs); block = 64 column lanes x 4 row lanes.
template <bool ABS, bool VEC4>
__global__ __launch_bounds__(kStatsBlock) void stats_cols_kernel(const float* __restrict__ x, int64_t rows,
                                                                 int64_t cols, int64_t rows_per_slab,
                                                                 float* __restrict__ pmin,
                                                                 float* __restrict__ pmax) {
  constexpr int W = VEC4 ? 4 : 1;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int64_t c0 = ((int64_t)blockIdx.x * 64 + tx) * W;
  const int64_t r0 = (int64_t)blockIdx.y * rows_per_slab;
  const int64_t r1 = min(rows, r0 + rows_per_slab);
  float lo[W], hi[W];
  bool nanw[W];
#pragma unroll
  for (int j = 0; j < W; ++j) { lo[j] = INFINITY; hi[j] = -INFINITY; nanw[j] = false; }
  if (c0 < cols) {
    int64_t r = r0 + ty;
    if (VEC4) {
      for (; r + 12 < r1; r += 16) {  // 4 independent 16-B loads in flight per lane
        float4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const float4*>(x + (r + 4 * u) * cols + c0);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          acc_minmax<ABS>(v[u].x, lo[0], hi[0], nanw[0]); acc_minmax<ABS>(v[u].y, lo[1 % W], hi[1 % W], nanw[1 % W]);
          acc_minmax<ABS>(v[u].z, lo[2 % W], hi[2 % W], nanw[2 % W]); acc_minmax<ABS>(v[u].w, lo[3 % W], hi[3 % W], nanw[3 % W]);
        }
      }
      for (; r < r1; r += 4) {
        float4 v = *reinterpret_cast<const float4*>(x + r * cols + c0);
        acc_minmax<ABS>(v.x, lo[0], hi[0], nanw[0]); acc_minmax<ABS>(v.y, lo[1 % W], hi[1 % W], nanw[1 % W]);
        acc_minmax<ABS>(v.z, lo[2 % W], hi[2 % W], nanw[2 % W]); acc_minmax<ABS>(v.w, lo[3 % W], hi[3 % W], nanw[3 % W]);
      }
    } else {
      for (; r < r1; r += 4) acc_minmax<ABS>(x[r * cols + c0], lo[0], hi[0], nanw[0]);
    }
  }
  __shared__ float s_lo[4][64 * W], s_hi[4][64 * W];
#pragma unroll
  for (int j = 0; j < W; ++j) { poison(nanw[j], lo[j], hi[j]); s_lo[ty][tx * W + j] = lo[j]; s_hi[ty][tx * W + j] = hi[j]; }
  __syncthreads();
  if (ty == 0 && c0 < cols) {
#pragma unroll
    for (int j = 0; j < W; ++j) {
      float a = s_lo[0][tx * W + j], b = s_hi[0][tx * W + j];
#pragma unroll
      for (int t = 1; t < 4; ++t) { a = nan_min(a, s_lo[t][tx * W + j]); b = nan_max(b, s_hi[t][tx * W + j]); }
      pmin[(int64_t)blockIdx.y * cols + c0 + j] = a;
      pmax[(int64_t)blockIdx.y * cols + c0 + j] = b;
    }
  }
}

// One contiguous chunk per block (per-tensor statistics).
template <bool ABS>
__global__ __launch_bounds__(kStatsBlock) void stats_flat_kernel(const float* __restrict__ x, int64_t total,
                                                                 int64_t chunk, float* __restrict__ pmin,
                                                                 float* __restrict__ pmax) {
  const int64_t b0 = (int64_t)blockIdx.x * chunk;
  const int64_t b1 = min(total, b0 + chunk);
  float lo = INFINITY, hi = -INFINITY;
  bool nan = false;
  if ((chunk & 3) == 0 && aligned16(x)) {
    const int64_t e4 = b0 + ((b1 - b0) & ~(int64_t)3);
    for (int64_t i = b0 + (int64_t)threadIdx.x * 4; i < e4; i += kStatsBlock * 4) {
      float4 v = *reinterpret_cast<const float4*>(x + i);
      acc_minmax<ABS>(v.x, lo, hi, nan); acc_minmax<ABS>(v.y, lo, hi, nan);
      acc_minmax<ABS>(v.z, lo, hi, nan); acc_minmax<ABS>(v.w, lo, hi, nan);
    }
    for (int64_t i = e4 + threadIdx.x; i < b1; i += kStatsBlock) acc_minmax<ABS>(x[i], lo, hi, nan);
  } else {
    for (int64_t i = b0 + threadIdx.x; i < b1; i += kStatsBlock) acc_minmax<ABS>(x[i], lo, hi, nan);
  }
  poison(nan, lo, hi);
  lo = wave_min_nan(lo); hi = wave_max_nan(hi);
  __shared__ float s_lo[kStatsBlock / 64], s_hi[kStatsBlock / 64];
  if ((threadIdx.x & 63) == 0) { s_lo[threadIdx.x >> 6] = lo; s_hi[threadIdx.x >> 6] = hi; }
  __syncthreads();
  if (threadIdx.x == 0) {
#pragma unroll
    for (int t = 1; t < kStatsBlock / 64; ++t) { lo = nan_min(lo, s_lo[t]); hi = nan_max(hi, s_hi[t]); }
    pmin[blockIdx.x] = lo; pmax[blockIdx.x] = hi;
  }
}

// "does ANY element exceed eps" of the log domain (quantization.py:177-179), over the partial maxima of all channels.
__global__ __launch_bounds__(1024) void stats_any_kernel(const float* __restrict__ pmax, int64_t n, float eps, int* __restrict__ flag) {
  float m0 = -INFINITY, m1 = -INFINITY, m2 = -INFINITY, m3 = -INFINITY;
  int64_t i = threadIdx.x;
  for (; i + 3 * 1024 < n; i += 4 * 1024) {            // four independent loads in flight per lane
    m0 = fmaxf(m0, pmax[i]); m1 = fmaxf(m1, pmax[i + 1024]); m2 = fmaxf(m2, pmax[i + 2048]); m3 = fmaxf(m3, pmax[i + 3072]);
  }
  for (; i < n; i += 1024) m0 = fmaxf(m0, pmax[i]);
  float g = wave_max(fmaxf(fmaxf(m0, m1), fmaxf(m2, m3)));
  __shared__ float sm[16];
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = g;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int t = 1; t < 16; ++t) g = fmaxf(g, sm[t]);
    *flag = (g > eps) ? 1 : 0;
  }
}

// part[S][chan] -> running update.  64 consecutive channels per workgroup (coalesced rows of the partial matrix), the four
// waves take every fourth slab, two independent accumulator pairs per lane, LDS combine.
__global__ __launch_bounds__(256) void stats_merge_kernel(const float* __restrict__ pmin,
                                                          const float* __restrict__ pmax, int64_t S,
                                                          int64_t chan, int log_domain, const int* __restrict__ any_flag,
                                                          float eps, float log_eps_fill, int first,
                                                          float* __restrict__ min_io,
                                                          float* __restrict__ max_io) {
  __shared__ float s_lo[4][64], s_hi[4][64];
  const int cl = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int64_t c = (int64_t)blockIdx.x * 64 + cl;
  const bool any = log_domain ? (*any_flag != 0) : true;
  float lo0 = INFINITY, hi0 = -INFINITY, lo1 = INFINITY, hi1 = -INFINITY;
  if (any && c < chan) {
    int64_t sl = w;
    for (; sl + 4 < S; sl += 8) {
      lo0 = nan_min(lo0, pmin[sl * chan + c]); hi0 = nan_max(hi0, pmax[sl * chan + c]);
      lo1 = nan_min(lo1, pmin[(sl + 4) * chan + c]); hi1 = nan_max(hi1, pmax[(sl + 4) * chan + c]);
    }
    for (; sl < S; sl += 4) { lo0 = nan_min(lo0, pmin[sl * chan + c]); hi0 = nan_max(hi0, pmax[sl * chan + c]); }
  }
  s_lo[w][cl] = nan_min(lo0, lo1); s_hi[w][cl] = nan_max(hi0, hi1);
  __syncthreads();
  if (w != 0 || c >= chan) return;
  float lo = nan_min(nan_min(s_lo[0][cl], s_lo[1][cl]), nan_min(s_lo[2][cl], s_lo[3][cl]));
  float hi = nan_max(nan_max(s_hi[0][cl], s_hi[1][cl]), nan_max(s_hi[2][cl], s_hi[3][cl]));
  if (!any) {
    if (first) { min_io[c] = log_eps_fill; max_io[c] = log_eps_fill; }  // :194-197
    return;                                                             // later batch: untouched
  }
  if (log_domain) {                                                     // :182-183
    lo = log2_rn(nan_max(lo, eps));                                     // torch.clamp(min=eps) keeps NaN
    hi = log2_rn(nan_max(hi, eps));
  }
  if (first) { min_io[c] = lo; max_io[c] = hi; }                        // :188-190 / :202-204
  else { min_io[c] = nan_min(min_io[c], lo); max_io[c] = nan_max(max_io[c], hi); }  // :192-193 / :206-207 (torch.minimum/maximum)
}

// SwitchableLayerNorm.forward (switchable_batchnorm.py:102-109).  One wave per row; the row lives in registers (NV float4 per
// lane), statistics are two-pass (mean, then mean of squared deviations) like the reference's x.mean / x.var(unbiased=False).
template <int NV>
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x, int64_t rows, int cols,
                                                        const float* __restrict__ w, const float* __restrict__ b, float eps,
                                                        float* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* xr = x + row * cols;
  float* orow = out + row * cols;
  float4 v[NV];
  float mean, den;
  ln_row_stats<NV>(xr, cols, eps, lane, v, mean, den);
  float4 wvs[NV], bvs[NV];                                   // weight / bias pieces: all loads before the arithmetic (see ln_row_load)
  const int last = (cols >> 2) - 1;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c4 = 4 * min(i * 64 + lane, last);
    wvs[i] = *reinterpret_cast<const float4*>(w + c4); bvs[i] = *reinterpret_cast<const float4*>(b + c4);
  }
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = (i * 64 + lane) * 4;
    if (c < cols) {
      const float4 wv = wvs[i], bv = bvs[i];
      float4 o;
      o.x = ln_apply(v[i].x, mean, den, wv.x, bv.x); o.y = ln_apply(v[i].y, mean, den, wv.y, bv.y);
      o.z = ln_apply(v[i].z, mean, den, wv.z, bv.z); o.w = ln_apply(v[i].w, mean, den, wv.w, bv.w);
      *reinterpret_cast<float4*>(orow + c) = o;
    }
  }
}

// {2^G, 2^-G} with max|x| * 2^G in [2^13, 2^14): the scale of the two-fp16-limb operand of an un-quantised tensor
// (the incoming gradient of the backward GEMM).  One block over the per-slab maxima of stats_flat_kernel<true>.
__global__ __launch_bounds__(256) void limb_scale_kernel(const float* __restrict__ pmax, int S, float* __restrict__ out2) {
  float m = 0.f;
  for (int i = threadIdx.x; i < S; i += 256) m = fmaxf(m, pmax[i]);
  m = wave_max(m);
  __shared__ float sm[4];
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    m = fmaxf(fmaxf(sm[0], sm[1]), fmaxf(sm[2], sm[3]));
    float p = 1.f;
    if (m > 0.f && m < INFINITY) {
      int e;
      (void)frexpf(m * (1.f + 0x1p-10f), &e);          // m = f * 2^e, f in [0.5, 1)
      e = 14 - e;
      e = e > 100 ? 100 : (e < -100 ? -100 : e);
      p = ldexpf(1.f, e);
    }
    out2[0] = p; out2[1] = 1.f / p;                      // exact: p is a power of two
  }
}

// =================================================================================================
// finish_calibration: running min/max -> scale, zero_point            quantization.py:110-127
// =================================================================================================
__global__ void finish_scale_kernel(const float* __restrict__ rmin, const float* __restrict__ rmax,
                                    int64_t len, int bits, int qtype, int symmetric, float eps,
                                    float* __restrict__ scale, float* __restrict__ zp) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= len) return;
  float lo = rmin[i], hi = rmax[i];
  if (qtype != SPQ_MINMAX) {                       // :111-117
    scale[i] = hi - lo;
    zp[i] = lo;
  } else if (symmetric) {                       // :119-123
    float amax = nan_max(nan_max(fabsf(lo), fabsf(hi)), eps);       // torch.max / torch.clamp keep NaN
    scale[i] = amax / (float)((1 << (bits - 1)) - 1);
    zp[i] = 0.f;
  } else {                                      // :124-127
    float rng = nan_max(hi - lo, eps);
    float s = rng / (float)((1u << bits) - 1u);
    scale[i] = s;
    zp[i] = rintf(-lo / s);
  }
}

// =================================================================================================
// Standalone fake-quant.  x viewed as [outer, chan, inner]; channel of flat index i = (i / inner) % chan.
// =================================================================================================
struct FQArgs {
  const float* x; const float* scale; const float* zp;
  float* out; void* levels;
  int64_t total, chan, inner;
  int per_channel; int bits;
  int log_direct;
};

template <typename L>
__device__ __forceinline__ void store_level(void* base, int64_t i, float q) {
  reinterpret_cast<L*>(base)[i] = (L)(int)q;
}

template <int QT, bool SYM, int LB>
__device__ __forceinline__ float fq_one(const FQArgs& a, int64_t i, float x, float s, float z, float qlo,
                                        float qhi, const LogParams& lp) {
  float q, o;
  if (QT == SPQ_MINMAX) {
    q = minmax_level<SYM>(x, s, z, qlo, qhi);
    o = minmax_dequant<SYM>(q, s, z);
  } else {
    q = log_level<SYM>(x, z, s, lp);
    o = log_dequant<SYM>(x, q, z, s, lp);
  }
  if (LB == 1) store_level<int8_t>(a.levels, i, q);
  if (LB == 2) store_level<int16_t>(a.levels, i, q);
  if (LB == 4) store_level<int32_t>(a.levels, i, q);
  return o;
}

// MODE 0: generic scalar; MODE 1: inner==1, chan%4==0 (scale varies fastest), float4;
// MODE 2: per-tensor or inner%4==0 (4 consecutive elements share a scale), float4.
template <int QT, bool SYM, int LB, int MODE, bool WRITE_OUT>
__global__ __launch_bounds__(256) void fakequant_kernel(FQArgs a) {
  float qlo, qhi;
  if (SYM) { qhi = (float)((1 << (a.bits - 1)) - 1); qlo = -qhi; }
  else { qlo = 0.f; qhi = (float)((1u << a.bits) - 1u); }
  const LogParams lp = make_log_params(a.bits, SYM, a.log_direct != 0);
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  if (MODE == 0) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < a.total; i += stride) {
      int64_t c = a.per_channel ? (i / a.inner) % a.chan : 0;
      float o = fq_one<QT, SYM, LB>(a, i, a.x[i], a.scale[c], a.zp[c], qlo, qhi, lp);
      if (WRITE_OUT) a.out[i] = o;
    }
  } else {
    const int64_t n4 = a.total >> 2;
    for (int64_t i4 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i4 < n4; i4 += stride) {
      const int64_t i = i4 << 2;
      float4 v = *reinterpret_cast<const float4*>(a.x + i);
      float4 s, z;
      if (MODE == 1) {
        int64_t c = i % a.chan;
        s = *reinterpret_cast<const float4*>(a.scale + c);
        z = *reinterpret_cast<const float4*>(a.zp + c);
      } else {
        int64_t c = a.per_channel ? (i / a.inner) % a.chan : 0;
        float s1 = a.scale[c], z1 = a.zp[c];
        s = make_float4(s1, s1, s1, s1); z = make_float4(z1, z1, z1, z1);
      }
      float4 o;
      o.x = fq_one<QT, SYM, LB>(a, i + 0, v.x, s.x, z.x, qlo, qhi, lp);
      o.y = fq_one<QT, SYM, LB>(a, i + 1, v.y, s.y, z.y, qlo, qhi, lp);
      o.z = fq_one<QT, SYM, LB>(a, i + 2, v.z, s.z, z.z, qlo, qhi, lp);
      o.w = fq_one<QT, SYM, LB>(a, i + 3, v.w, s.w, z.w, qlo, qhi, lp);
      if (WRITE_OUT) *reinterpret_cast<float4*>(a.out + i) = o;
    }
  }
}

// out[c, r] = out_scaling * FQ(x)[r, c]; x [rows, cols], scale per column or per tensor. 32x32 LDS tiles.
template <int QT, bool SYM>
__global__ __launch_bounds__(256) void fakequant_transposed_kernel(const float* __restrict__ x, int64_t rows,
                                                                   int64_t cols, const float* __restrict__ scale,
                                                                   const float* __restrict__ zp, int per_channel,
                                                                   int bits, int log_direct, float out_scaling,
                                                                   float* __restrict__ out) {
  __shared__ float tile[32][33];
  float qlo, qhi;
  if (SYM) { qhi = (float)((1 << (bits - 1)) - 1); qlo = -qhi; }
  else { qlo = 0.f; qhi = (float)((1u << bits) - 1u); }
  const LogParams lp = make_log_params(bits, SYM, log_direct != 0);
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  const int64_t c = (int64_t)blockIdx.x * 32 + tx;
  for (int j = ty; j < 32; j += 8) {
    const int64_t r = (int64_t)blockIdx.y * 32 + j;
    float o = 0.f;
    if (r < rows && c < cols) {
      const int64_t ch = per_channel ? c : 0;
      const float s = scale[ch], z = zp[ch], v = x[r * cols + c];
      if (QT == SPQ_MINMAX) o = minmax_dequant<SYM>(minmax_level<SYM>(v, s, z, qlo, qhi), s, z);
      else o = log_dequant<SYM>(v, log_level<SYM>(v, z, s, lp), z, s, lp);
      if (out_scaling != 1.0f) o = o * out_scaling;
    }
    tile[j][tx] = o;
  }
  __syncthreads();
  const int64_t r = (int64_t)blockIdx.y * 32 + tx;
  for (int j = ty; j < 32; j += 8) {
    const int64_t cc = (int64_t)blockIdx.x * 32 + j;
    if (r < rows && cc < cols) out[cc * rows + r] = tile[tx][j];
  }
}

}  // namespace spq

// =================================================================================================
// C ABI
// =================================================================================================
using namespace spq;

static int64_t ceil_div64(int64_t a, int64_t b) { return (a + b - 1) / b; }

extern "C" size_t spq_stats_workspace_bytes(int64_t outer, int64_t chan, int64_t inner, int per_channel) {
  if (outer <= 0 || chan <= 0 || inner <= 0) return 0;
  size_t n;
  if (!per_channel) n = (size_t)kMaxSlabs * 2;                               // flat: [slabs] x {min,max}
  else if (inner == 1) n = (size_t)kMaxSlabs * (size_t)chan * 2;             // cols: [slabs][chan]
  else if (outer == 1) n = (size_t)chan * 2;                                 // rows: [chan]
  else n = (size_t)outer * (size_t)chan * 2 + (size_t)kMaxSlabs * (size_t)chan * 3;  // rows then cols
  return n * sizeof(float) + 64;
}

extern "C" int spq_minmax_stats(const float* x, int64_t outer, int64_t chan, int64_t inner, int per_channel,
                                int log_domain, float eps, float log_eps_fill, int first_batch, float* min_io,
                                float* max_io, void* workspace, size_t workspace_bytes, spq_stream_t stream) {
  SPQ_REQUIRE(x && min_io && max_io && workspace, "spq_minmax_stats: null pointer");
  SPQ_REQUIRE(outer > 0 && chan > 0 && inner > 0, "spq_minmax_stats: empty tensor (%lld,%lld,%lld)",
              (long long)outer, (long long)chan, (long long)inner);
  if (workspace_bytes < spq_stats_workspace_bytes(outer, chan, inner, per_channel) || !aligned16(workspace)) {
    set_error("spq_minmax_stats: workspace too small or misaligned (%zu bytes)", workspace_bytes);
    return SPQ_ERR_WORKSPACE;
  }
  hipStream_t st = (hipStream_t)stream;
  float* ws = (float*)workspace;
  const bool ab = log_domain != 0;
  const int64_t total = outer * chan * inner;
  int64_t S = 0, C = 0;  // partial layout [S][C]
  float *pmin = nullptr, *pmax = nullptr;

  auto launch_cols = [&](const float* src, int64_t rows, int64_t cols, float* omin, float* omax, bool absmode) {
    // enough slabs to fill the chip, at least 16 rows each
    int64_t slabs = std::min<int64_t>(kMaxSlabs / 2, std::max<int64_t>(1, rows / 16));
    int64_t rps = ceil_div64(rows, slabs);
    rps = ceil_div64(rps, 4) * 4;
    slabs = ceil_div64(rows, rps);
    const bool v4 = (cols % 4 == 0) && aligned16(src);
    dim3 grid((unsigned)ceil_div64(cols, v4 ? 256 : 64), (unsigned)slabs);
    if (absmode) {
      if (v4) stats_cols_kernel<true, true><<<grid, kStatsBlock, 0, st>>>(src, rows, cols, rps, omin, omax);
      else stats_cols_kernel<true, false><<<grid, kStatsBlock, 0, st>>>(src, rows, cols, rps, omin, omax);
    } else {
      if (v4) stats_cols_kernel<false, true><<<grid, kStatsBlock, 0, st>>>(src, rows, cols, rps, omin, omax);
      else stats_cols_kernel<false, false><<<grid, kStatsBlock, 0, st>>>(src, rows, cols, rps, omin, omax);
    }
    return slabs;
  };

  if (!per_channel) {
    int64_t slabs = std::min<int64_t>(kMaxSlabs, std::max<int64_t>(1, total / 4096));
    int64_t chunk = ceil_div64(ceil_div64(total, slabs), 4) * 4;
    slabs = ceil_div64(total, chunk);
    pmin = ws; pmax = ws + kMaxSlabs;
    if (ab) stats_flat_kernel<true><<<(unsigned)slabs, kStatsBlock, 0, st>>>(x, total, chunk, pmin, pmax);
    else stats_flat_kernel<false><<<(unsigned)slabs, kStatsBlock, 0, st>>>(x, total, chunk, pmin, pmax);
    // partials are [slabs] for a single channel -> treat as S=slabs, C=1
    S = slabs; C = 1;
  } else if (inner == 1) {
    pmin = ws; pmax = ws + (size_t)kMaxSlabs * chan;
    S = launch_cols(x, outer, chan, pmin, pmax, ab);
    C = chan;
  } else {
    const int64_t nrows = outer * chan;
    float* rmin = ws; float* rmax = ws + nrows;
    const unsigned blocks = (unsigned)ceil_div64(nrows, kStatsBlock / 64);
    if (ab) stats_rows_kernel<true><<<blocks, kStatsBlock, 0, st>>>(x, nrows, inner, rmin, rmax);
    else stats_rows_kernel<false><<<blocks, kStatsBlock, 0, st>>>(x, nrows, inner, rmin, rmax);
    if (outer == 1) { pmin = rmin; pmax = rmax; S = 1; C = chan; }
    else {
      // second stage over [outer, chan]: min of the row minima, max of the row maxima.  The column kernel
      // always produces both statistics of its input, so it runs once per side and the unused half of each
      // result lands in a junk region of the workspace.
      const size_t slab = (size_t)kMaxSlabs * chan;
      float* keep_min = ws + 2 * nrows;
      float* junk = keep_min + slab;
      float* keep_max = junk + slab;
      S = launch_cols(rmin, outer, chan, keep_min, junk, false);
      launch_cols(rmax, outer, chan, junk, keep_max, false);
      pmin = keep_min; pmax = keep_max; C = chan;
    }
  }
  int rc = check_launch("spq_minmax_stats(partials)");
  if (rc) return rc;
  // the flag word lives at the very end of the workspace (spq_stats_workspace_bytes reserves 64 spare bytes)
  int* any_flag = reinterpret_cast<int*>((char*)workspace + (spq_stats_workspace_bytes(outer, chan, inner, per_channel) - 16) / 16 * 16);
  if (log_domain) stats_any_kernel<<<1, 1024, 0, st>>>(pmax, S * C, eps, any_flag);
  stats_merge_kernel<<<(unsigned)ceil_div64(C, 64), 256, 0, st>>>(pmin, pmax, S, C, log_domain, any_flag, eps, log_eps_fill,
                                                                   first_batch, min_io, max_io);
  return check_launch("spq_minmax_stats(merge)");
}

extern "C" int spq_layernorm(const float* x, int64_t rows, int64_t cols, const float* weight, const float* bias, float eps,
                             float* out, spq_stream_t stream) {
  SPQ_REQUIRE(x && weight && bias && out, "spq_layernorm: null pointer");
  SPQ_REQUIRE(rows > 0 && cols > 0, "spq_layernorm: empty tensor");
  if (cols % 4 != 0 || cols > 8192 || !aligned16(x) || !aligned16(out) || !aligned16(weight) || !aligned16(bias)) {
    set_error("spq_layernorm: needs cols %% 4 == 0, cols <= 8192 and 16-byte aligned tensors (cols = %lld)", (long long)cols);
    return SPQ_ERR_UNSUPPORTED;
  }
  hipStream_t st = (hipStream_t)stream;
  const unsigned grid = (unsigned)ceil_div64(rows, 4);
  const int nv = (int)ceil_div64(cols, 256);
  if (nv <= 2) layernorm_kernel<2><<<grid, 256, 0, st>>>(x, rows, (int)cols, weight, bias, eps, out);
  else if (nv <= 4) layernorm_kernel<4><<<grid, 256, 0, st>>>(x, rows, (int)cols, weight, bias, eps, out);
  else if (nv <= 8) layernorm_kernel<8><<<grid, 256, 0, st>>>(x, rows, (int)cols, weight, bias, eps, out);
  else if (nv <= 16) layernorm_kernel<16><<<grid, 256, 0, st>>>(x, rows, (int)cols, weight, bias, eps, out);
  else layernorm_kernel<32><<<grid, 256, 0, st>>>(x, rows, (int)cols, weight, bias, eps, out);
  return check_launch("spq_layernorm");
}

extern "C" int spq_dynamic_limb_scale(const float* x, int64_t n, float* scale_out2, void* workspace,
                                      size_t workspace_bytes, spq_stream_t stream) {
  SPQ_REQUIRE(x && scale_out2 && workspace && n > 0, "spq_dynamic_limb_scale: null pointer or empty tensor");
  constexpr int kSlabs = SPQ_LIMB_SCALE_WORKSPACE_BYTES / 8;     // {min, max} partial per slab
  if (workspace_bytes < SPQ_LIMB_SCALE_WORKSPACE_BYTES || !aligned16(workspace)) {
    set_error("spq_dynamic_limb_scale: workspace too small or misaligned (%zu bytes)", workspace_bytes);
    return SPQ_ERR_WORKSPACE;
  }
  hipStream_t st = (hipStream_t)stream;
  float* ws = (float*)workspace;
  int64_t slabs = std::min<int64_t>(kSlabs, std::max<int64_t>(1, n / 4096));
  const int64_t chunk = ceil_div64(ceil_div64(n, slabs), 4) * 4;
  slabs = ceil_div64(n, chunk);
  stats_flat_kernel<true><<<(unsigned)slabs, kStatsBlock, 0, st>>>(x, n, chunk, ws, ws + kSlabs);
  limb_scale_kernel<<<1, 256, 0, st>>>(ws + kSlabs, (int)slabs, scale_out2);
  return check_launch("spq_dynamic_limb_scale");
}

extern "C" int spq_finish_scale(const float* rmin, const float* rmax, int64_t len, int bits, int qtype,
                                int symmetric, float eps, float* scale_out, float* zp_out, spq_stream_t stream) {
  SPQ_REQUIRE(rmin && rmax && scale_out && zp_out, "spq_finish_scale: null pointer");
  SPQ_REQUIRE(len > 0, "spq_finish_scale: len must be positive");
  SPQ_REQUIRE(bits >= 1 && bits <= 24, "spq_finish_scale: bits %d outside [1,24]", bits);
  SPQ_REQUIRE(qtype >= SPQ_MINMAX && qtype <= SPQ_LOG_DIRECT, "spq_finish_scale: unknown quantizer type %d", qtype);
  finish_scale_kernel<<<(unsigned)ceil_div64(len, 256), 256, 0, (hipStream_t)stream>>>(
      rmin, rmax, len, bits, qtype, symmetric, eps, scale_out, zp_out);
  return check_launch("spq_finish_scale");
}

template <int QT, bool SYM, int LB>
static void launch_fq(const FQArgs& a, int mode, bool write_out, hipStream_t st) {
  const int64_t work = mode == 0 ? a.total : (a.total >> 2);
  const unsigned grid = (unsigned)std::min<int64_t>(ceil_div64(work, 256), 256 * 16);
#define SPQ_FQ(MODE)                                                                  \
  do {                                                                                \
    if (write_out) fakequant_kernel<QT, SYM, LB, MODE, true><<<grid, 256, 0, st>>>(a); \
    else fakequant_kernel<QT, SYM, LB, MODE, false><<<grid, 256, 0, st>>>(a);          \
  } while (0)
  if (mode == 0) SPQ_FQ(0);
  else if (mode == 1) SPQ_FQ(1);
  else SPQ_FQ(2);
#undef SPQ_FQ
}

extern "C" int spq_fakequant(const float* x, int64_t outer, int64_t chan, int64_t inner, const float* scale,
                             const float* zp, int per_channel, int bits, int qtype, int symmetric,
                             float* out_f32, void* out_levels, int levels_bytes, spq_stream_t stream) {
  SPQ_REQUIRE(x && scale && zp, "spq_fakequant: null pointer");
  SPQ_REQUIRE(out_f32 || out_levels, "spq_fakequant: no output requested");
  SPQ_REQUIRE(outer > 0 && chan > 0 && inner > 0, "spq_fakequant: empty tensor");
  SPQ_REQUIRE(bits >= 1 && bits <= 24, "spq_fakequant: bits %d outside [1,24]", bits);
  SPQ_REQUIRE(qtype >= SPQ_MINMAX && qtype <= SPQ_LOG_DIRECT, "spq_fakequant: unknown quantizer type %d", qtype);
  const int lb = out_levels ? levels_bytes : 0;
  SPQ_REQUIRE(lb == 0 || lb == 1 || lb == 2 || lb == 4, "spq_fakequant: levels_bytes must be 1, 2 or 4");
  const int maxlevel = symmetric ? (1 << (bits - 1)) - 1 : (1 << bits) - 1;
  SPQ_REQUIRE(lb == 0 || lb == 4 || maxlevel <= (lb == 1 ? 127 : 32767),
              "spq_fakequant: %d-bit %s levels do not fit int%d", bits, symmetric ? "symmetric" : "asymmetric",
              lb * 8);
  FQArgs a{x, scale, zp, out_f32, out_levels, outer * chan * inner, chan, inner, per_channel, bits,
           qtype == SPQ_LOG_DIRECT ? 1 : 0};
  int mode = 0;
  const bool al = aligned16(x) && (!out_f32 || aligned16(out_f32)) && (a.total % 4 == 0);
  if (al && per_channel && inner == 1 && chan % 4 == 0 && aligned16(scale) && aligned16(zp)) mode = 1;
  else if (al && (!per_channel || inner % 4 == 0)) mode = 2;
  hipStream_t st = (hipStream_t)stream;
  const bool wo = out_f32 != nullptr;
#define SPQ_DISPATCH_LB(QT, SYM)                              \
  do {                                                        \
    if (lb == 0) launch_fq<QT, SYM, 0>(a, mode, wo, st);      \
    else if (lb == 1) launch_fq<QT, SYM, 1>(a, mode, wo, st); \
    else if (lb == 2) launch_fq<QT, SYM, 2>(a, mode, wo, st); \
    else launch_fq<QT, SYM, 4>(a, mode, wo, st);              \
  } while (0)
  if (qtype == SPQ_MINMAX) { if (symmetric) SPQ_DISPATCH_LB(SPQ_MINMAX, true); else SPQ_DISPATCH_LB(SPQ_MINMAX, false); }
  else { if (symmetric) SPQ_DISPATCH_LB(SPQ_LOG, true); else SPQ_DISPATCH_LB(SPQ_LOG, false); }
#undef SPQ_DISPATCH_LB
  return check_launch("spq_fakequant");
}

extern "C" int spq_fakequant_transposed(const float* x, int64_t rows, int64_t cols, const float* scale,
                                        const float* zp, int per_channel, int bits, int qtype, int symmetric,
                                        float out_scaling, float* out_f32, spq_stream_t stream) {
  SPQ_REQUIRE(x && scale && zp && out_f32, "spq_fakequant_transposed: null pointer");
  SPQ_REQUIRE(rows > 0 && cols > 0, "spq_fakequant_transposed: empty tensor");
  SPQ_REQUIRE(bits >= 1 && bits <= 24, "spq_fakequant_transposed: bits %d outside [1,24]", bits);
  SPQ_REQUIRE(qtype >= SPQ_MINMAX && qtype <= SPQ_LOG_DIRECT, "spq_fakequant_transposed: unknown quantizer type %d", qtype);
  dim3 grid((unsigned)ceil_div64(cols, 32), (unsigned)ceil_div64(rows, 32));
  hipStream_t st = (hipStream_t)stream;
  if (qtype == SPQ_MINMAX) {
    if (symmetric) fakequant_transposed_kernel<SPQ_MINMAX, true><<<grid, 256, 0, st>>>(x, rows, cols, scale, zp, per_channel, bits, qtype == SPQ_LOG_DIRECT ? 1 : 0, out_scaling, out_f32);
    else fakequant_transposed_kernel<SPQ_MINMAX, false><<<grid, 256, 0, st>>>(x, rows, cols, scale, zp, per_channel, bits, qtype == SPQ_LOG_DIRECT ? 1 : 0, out_scaling, out_f32);
  } else {
    if (symmetric) fakequant_transposed_kernel<SPQ_LOG, true><<<grid, 256, 0, st>>>(x, rows, cols, scale, zp, per_channel, bits, qtype == SPQ_LOG_DIRECT ? 1 : 0, out_scaling, out_f32);
    else fakequant_transposed_kernel<SPQ_LOG, false><<<grid, 256, 0, st>>>(x, rows, cols, scale, zp, per_channel, bits, qtype == SPQ_LOG_DIRECT ? 1 : 0, out_scaling, out_f32);
  }
  return check_launch("spq_fakequant_transposed");
}
