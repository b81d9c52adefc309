// Proof of concept for DESIGN.md section 6 "another operand format" (NOT part of the library): the base contraction of the headline
// layer with FP6 (e2m3) operands on v_mfma_scale_f32_16x16x128_f8f6f4 --
//   activation levels q in [-7, 7]: exact in e2m3;
//   W' = FQ(W)[n,k] sx[k], scaled per row to a 25-bit integer w = round(W' 2^E[n]), as FIVE balanced radix-32 digits d_i, |d_i| <= 16,
//   stored as the e2m3 value d_i / 8 (e2m3 holds every multiple of 1/8 up to 2.0), the factor 8 * 32^i in the instruction's block scale:
//        y[m,n] = 2^-E[n] sum_i 2^(3 + 5 i) sum_k q[m,k] (d_i[n,k] / 8)          five MFMAs per 128-deep k block and 16 x 16 outputs
// against the production kernel's two f16 MFMAs per 32-deep k block (the same structure otherwise: 128 x 128 tiles, 4 waves, one stage
// buffer per workgroup, LDS-DMA copies, the production epilogue).  It checks the result against the f16-limb kernel and a double sum,
// and times both (no LoRA stages in either).
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -Wno-unused-value tools/fp6_gemm_probe/fp6_gemm_probe.hip -o tools/fp6_gemm_probe/fp6_gemm_probe
#include <stdarg.h>
#include <string.h>
#include <vector>
#include <algorithm>
#include <random>
#include "../../llm-qat-on-gpt2_amd/csrc/spq_f16x2.hip"
namespace spq {
void set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fprintf(stderr, "\n"); }
int check_launch(const char* what) { hipError_t e = hipGetLastError(); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", what, hipGetErrorString(e)); return -2; } return 0; }

typedef int v8i __attribute__((ext_vector_type(8)));
#ifndef FP6_NL
#define FP6_NL 5            // weight digits (limb planes)
#endif
#ifndef FP6_WGS
#define FP6_WGS 2           // workgroups per CU (72 KB of LDS each)
#endif
constexpr int F6_REC = 1536;                        // one MFMA operand tile: 16 rows x 128 k x 6 bit, = [64 lanes x 16 B | 64 lanes x 8 B]
constexpr int F6_PAIR = 2 * F6_REC;                 // two records interleaved: [rec0 16-B part | rec1 16-B part | rec0 8-B part | rec1 8-B part]
constexpr int F6_PLANE = 4 * F6_PAIR;               // 128 rows x 128 k of one operand plane: 12 KB
constexpr int F6_STAGE = (1 + FP6_NL) * F6_PLANE;   // A + NL weight planes: 72 KB

struct GemmFp6Args {
  const unsigned char* A6;      // [Mp/32 pairs][K/128][3072 B]
  const unsigned char* W6;      // [NL][Np/32 pairs][K/128][3072 B]
  const float *rowscale, *bias; // [Np], [N]
  float* y;
  int M, N, K;                  // K % 128 == 0
  int tiles_m, tiles_n;         // 128 x 128 tiles
};

__global__ __launch_bounds__(256, FP6_WGS) void gemm_fp6_t128_kernel(GemmFp6Args g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = w >> 1, wn = w & 1;
  const int l15 = lane & 15, q4 = lane >> 4;
  const int nwg = g.tiles_m * g.tiles_n;
  const int KB = g.K / 128;                                  // stages per tile
  const int gstride = (int)gridDim.x;
  auto tile_of = [&](int p, int& bm, int& bn) {              // the production kernel's XCD-aware band order
    const int q8 = nwg >> 3, r8 = nwg & 7, xcd = p & 7;
    const int wgid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (p >> 3);
    constexpr int GROUP_M = T128_GROUP_M;
    const int band = wgid / (GROUP_M * g.tiles_n);
    const int band_rows = min(GROUP_M, g.tiles_m - band * GROUP_M);
    const int in_band = wgid - band * GROUP_M * g.tiles_n;
    bm = (band * GROUP_M + in_band % band_rows) * 128;
    bn = (in_band / band_rows) * 128;
  };
  int p = blockIdx.x;
  if (p >= nwg) return;
  int bm, bn;
  tile_of(p, bm, bn);
  const int64_t plane_stride = (int64_t)(g.tiles_n * 4) * KB * F6_PAIR;   // bytes of one weight plane
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  const unsigned voff = (unsigned)lane * 16u;               // the ONE per-lane offset: everything else of a copy's address is wave-uniform
  // copies: a plane of a stage is 4 pairs x 3 KB = 12 pieces of 1 KB; wave w takes pieces 3w .. 3w+2 of every plane (scalar base +
  // per-lane offset, inline asm: the builtin form kept 36 address registers alive through the stage loop)
  auto glds = [&](const unsigned char* sbase, unsigned lds_addr) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(voff), "s"(sbase), "s"(lds_addr) : "memory");
  };
  auto issue = [&](int kb, int tbm, int tbn) {
    const unsigned char* a_src = g.A6 + ((int64_t)(tbm / 32) * KB + kb) * F6_PAIR;      // pair p of the tile: + p * KB * F6_PAIR
    const unsigned char* w_src = g.W6 + ((int64_t)(tbn / 32) * KB + kb) * F6_PAIR;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int piece = 3 * w + i, pair = piece / 3, sub = piece % 3;
      const int64_t off = (int64_t)pair * KB * F6_PAIR + sub * 1024;
      glds(a_src + off, lds0 + piece * 1024);
#pragma unroll
      for (int pl = 0; pl < FP6_NL; ++pl) glds(w_src + pl * plane_stride + off, lds0 + (1 + pl) * F6_PLANE + piece * 1024);
    }
  };
  // operand tile r (0..7) of a plane: 16-B part at pair base + (r & 1) KB + lane * 16, 8-B part at pair base + 2 KB + (r & 1) * 512 + lane * 8
  auto frag = [&](int plane_off, int r) -> v8i {
    const char* pb = smem + plane_off + (r >> 1) * F6_PAIR;
    const uint4 a = *reinterpret_cast<const uint4*>(pb + (r & 1) * 1024 + lane * 16);
    const uint2 b = *reinterpret_cast<const uint2*>(pb + 2048 + (r & 1) * 512 + lane * 8);
    v8i v; v[0] = (int)a.x; v[1] = (int)a.y; v[2] = (int)a.z; v[3] = (int)a.w; v[4] = (int)b.x; v[5] = (int)b.y; v[6] = 0; v[7] = 0;
    return v;
  };
  f32x4 acc[4][4];
  int prio_ctr = (int)(blockIdx.x / (gridDim.x / FP6_WGS > 0 ? gridDim.x / FP6_WGS : 1));
  int stage_ctr = 0;
  issue(0, bm, bn);
  while (true) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[i][j][e] = 0.f;
    const int pn = p + gstride;
    const bool more = pn < nwg;
    int nbm = 0, nbn = 0;
    if (more) tile_of(pn, nbm, nbn);
    for (int kb = 0; kb < KB; ++kb) {
      {                                                      // the CU's workgroups take the issue priorities in turn (as the production kernel)
        const int per = max(1, (KB * ((nwg + gstride - 1) / gstride) + 5) / 6);
        if (stage_ctr % per == 0) { if (((prio_ctr + stage_ctr / per) % FP6_WGS) == 0) __builtin_amdgcn_s_setprio(0); else __builtin_amdgcn_s_setprio(1); }
        ++stage_ctr;
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // (the compiler's own counts do not see the asm copies)
      __syncthreads();                                       // barrier: the stage has landed
      v8i fa[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) fa[t] = frag(0, 4 * wm + t);
      // the next plane's fragments are read under this plane's 16 MFMAs; two named fragment sets, planes unrolled by hand
      v8i b0[4], b1[4];
      auto load_plane = [&](v8i (&fb)[4], int pl) {
#pragma unroll
        for (int t = 0; t < 4; ++t) fb[t] = frag((1 + pl) * F6_PLANE, 4 * wn + t);
      };
      auto mfma_plane = [&](const v8i (&fb)[4], int pl) {
        const int sb = 127 + 3 + 5 * pl;                     // E8M0: the digit's weight 8 * 32^pl
#pragma unroll
        for (int tm = 0; tm < 4; ++tm)
#pragma unroll
          for (int tn = 0; tn < 4; ++tn)
            acc[tm][tn] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(fa[tm], fb[tn], acc[tm][tn], 2, 2, 0, 127, 0, sb);
        __builtin_amdgcn_sched_barrier(0);
      };
      load_plane(b0, 0);
      static_assert(FP6_NL == 5 || FP6_NL == 4 || FP6_NL == 6, "planes are unrolled by hand");
      load_plane(b1, 1); mfma_plane(b0, 0);
      load_plane(b0, 2); mfma_plane(b1, 1);
      load_plane(b1, 3); mfma_plane(b0, 2);
      if (FP6_NL >= 5) load_plane(b0, 4);
      mfma_plane(b1, 3);
      if (FP6_NL >= 6) load_plane(b1, 5);
      if (FP6_NL >= 5) mfma_plane(b0, 4);
      if (FP6_NL >= 6) mfma_plane(b1, 5);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory");   // every wave has read its fragments
      if (kb + 1 < KB) issue(kb + 1, bm, bn);
    }
    // the production epilogue: scale, bias, transpose through per-wave LDS slices (inside the free stage buffer), whole 128-B lines
    float4 ep_rs[2], ep_bv[2];
#pragma unroll
    for (int tn = 0; tn < 2; ++tn) {
      const int n = bn + wn * 64 + tn * 32 + (lane & 7) * 4;
      ep_rs[tn] = make_float4(0.f, 0.f, 0.f, 0.f); ep_bv[tn] = ep_rs[tn];
      if (n < g.N) { ep_rs[tn] = *reinterpret_cast<const float4*>(g.rowscale + n); if (g.bias) ep_bv[tn] = *reinterpret_cast<const float4*>(g.bias + n); }
    }
    {
      char* eb = smem + w * EPI_WAVE;
      const int c4 = (lane & 7) * 4;
      const bool interior = (bm + 128 <= g.M) && (bn + 128 <= g.N);
#pragma unroll
      for (int tn = 0; tn < 2; ++tn) {
        const int n = bn + wn * 64 + tn * 32 + c4;
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
            float* dst = g.y + (int64_t)m * g.N + n;
            if (interior) *reinterpret_cast<float4*>(dst) = o;
            else if (n < g.N && m < g.M) *reinterpret_cast<float4*>(dst) = o;
          }
        }
      }
    }
    if (!more) break;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory");
    issue(0, nbm, nbn);
    p = pn; bm = nbm; bn = nbn;
  }
}
}  // namespace spq
using namespace spq;

static int e2m3_of_eighths(int d) {         // d / 8 with |d| <= 16: exact in e2m3
  const int a = abs(d);
  int code;
  if (a < 8) code = a;                       // subnormal: mant / 8
  else if (a < 16) code = 0x08 | (a - 8);    // exp 1: 1 + mant / 8
  else code = 0x10;                          // 2.0
  return (d < 0 ? 0x20 : 0) | code;
}
static int e2m3_of_int(int q) {             // |q| <= 7
  static const int code[8] = {0x00, 0x08, 0x10, 0x14, 0x18, 0x1A, 0x1C, 0x1E};
  return (q < 0 ? 0x20 : 0) | code[abs(q)];
}
// pack codes[rows][K] (one byte each) into the pair-interleaved operand-tile layout
static std::vector<unsigned char> pack6(const std::vector<unsigned char>& codes, int rows, int K) {
  const int KB = K / 128, pairs = rows / 32;
  std::vector<unsigned char> out((size_t)pairs * KB * F6_PAIR, 0);
  for (int pr = 0; pr < pairs; ++pr)
    for (int kb = 0; kb < KB; ++kb) {
      unsigned char* base = out.data() + ((size_t)pr * KB + kb) * F6_PAIR;
      for (int rec = 0; rec < 2; ++rec)
        for (int l = 0; l < 64; ++l) {
          const int row = pr * 32 + rec * 16 + (l & 15), k0 = kb * 128 + 32 * (l >> 4);
          unsigned char bytes[24] = {0};
          for (int i = 0; i < 32; ++i) {
            const unsigned c = codes[(size_t)row * K + k0 + i];
            const int bit = 6 * i;
            bytes[bit >> 3] |= (unsigned char)(c << (bit & 7));
            if ((bit & 7) > 2) bytes[(bit >> 3) + 1] |= (unsigned char)(c >> (8 - (bit & 7)));
          }
          memcpy(base + rec * 1024 + l * 16, bytes, 16);
          memcpy(base + 2048 + rec * 512 + l * 8, bytes + 16, 8);
        }
    }
  return out;
}

int main(int argc, char** argv) {
  const int M = 8192, N = 3072, K = 768;
  std::mt19937 rng(1);
  std::uniform_int_distribution<int> lv(-7, 7); std::normal_distribution<float> nd(0.f, 0.02f);
  std::vector<int> q((size_t)M * K); for (auto& v : q) v = lv(rng);
  std::vector<float> Wp((size_t)N * K); for (auto& v : Wp) v = nd(rng) * (1.f + (float)(rng() % 7));
  // ---- f16-limb operands (production kernel) and FP6 digit operands from the same W'
  std::vector<_Float16> qx((size_t)M * K), whi((size_t)N * K), wlo((size_t)N * K);
  std::vector<float> rs16(N), rs6(N);
  std::vector<unsigned char> qc((size_t)M * K);
  std::vector<std::vector<unsigned char>> dc(FP6_NL, std::vector<unsigned char>((size_t)N * K));
  for (size_t i = 0; i < q.size(); ++i) { qx[i] = (_Float16)(float)q[i]; qc[i] = (unsigned char)e2m3_of_int(q[i]); }
  double digit_err = 0;
  for (int n = 0; n < N; ++n) {
    float mx = 0; for (int k = 0; k < K; ++k) mx = fmaxf(mx, fabsf(Wp[(size_t)n * K + k]));
    int ex; frexpf(mx, &ex);
    const float p16 = ldexpf(1.f, 14 - ex);            // production: row max in [2^13, 2^14)
    const int bits = 5 * FP6_NL - 1;                    // digits cover |w| < 2^bits (balanced radix 32)
    const double p6 = ldexp(1.0, bits - ex);            // row max in [2^(bits-1), 2^bits)
    rs16[n] = 1.f / p16; rs6[n] = (float)(1.0 / p6);
    for (int k = 0; k < K; ++k) {
      const float v = Wp[(size_t)n * K + k] * p16;
      const _Float16 h = (_Float16)v; whi[(size_t)n * K + k] = h; wlo[(size_t)n * K + k] = (_Float16)(v - (float)h);
      long long wi = llround((double)Wp[(size_t)n * K + k] * p6);
      digit_err = fmax(digit_err, fabs((double)wi / p6 - Wp[(size_t)n * K + k]) / (mx > 0 ? mx : 1));
      const int sgn = wi < 0 ? -1 : 1;                  // digits of |w| in [-15, 16] (so that 2^24 fits), negated for w < 0: |d| <= 16 either way
      wi = llabs(wi);
      for (int i = 0; i < FP6_NL; ++i) {
        int d = (int)(wi % 32); if (d > 16) d -= 32;
        wi = (wi - d) / 32;
        dc[i][(size_t)n * K + k] = (unsigned char)e2m3_of_eighths(sgn * d);
      }
      if (wi != 0) { printf("digit overflow\n"); return 1; }
    }
  }
  printf("digits: %d planes, worst |W' - digits| / rowmax = %.3g (f16 limbs: 2^-22 = %.3g of the element)\n", FP6_NL, digit_err, ldexp(1.0, -22));
  auto up = [&](const void* h, size_t bytes) { void* d; hipMalloc(&d, bytes); hipMemcpy(d, h, bytes, hipMemcpyHostToDevice); return d; };
  float *bias, *y16, *y6;
  hipMalloc(&bias, N * 4); hipMemset(bias, 0, N * 4);
  hipMalloc(&y16, (size_t)M * N * 4); hipMalloc(&y6, (size_t)M * N * 4);
  hipMemset(y16, 0, (size_t)M * N * 4); hipMemset(y6, 0, (size_t)M * N * 4);
  GemmF16Args g{};
  g.qx = (const _Float16*)up(qx.data(), qx.size() * 2); g.Whi = (const _Float16*)up(whi.data(), whi.size() * 2); g.Wlo = (const _Float16*)up(wlo.data(), wlo.size() * 2);
  g.thi = g.tlo = g.Bhi = g.Blo = nullptr; g.rowinv = nullptr;
  g.rowscale = (const float*)up(rs16.data(), N * 4); g.bias = bias; g.y = y16; g.M = M; g.N = N; g.Kp = K; g.Rp = 0;
  g.tiles_m = M / GM; g.tiles_n = N / GN; g.a_limbs = 1; g.split = 1;
  GemmFp6Args f{};
  { auto pa = pack6(qc, M, K); f.A6 = (const unsigned char*)up(pa.data(), pa.size()); }
  { std::vector<unsigned char> all; for (int i = 0; i < FP6_NL; ++i) { auto pw = pack6(dc[i], N, K); all.insert(all.end(), pw.begin(), pw.end()); }
    f.W6 = (const unsigned char*)up(all.data(), all.size()); }
  f.rowscale = (const float*)up(rs6.data(), N * 4); f.bias = bias; f.y = y6; f.M = M; f.N = N; f.K = K; f.tiles_m = M / 128; f.tiles_n = N / 128;
  auto k16 = gemm_f16x2_t128_kernel<1, 0>;
  hipFuncSetAttribute((const void*)k16, hipFuncAttributeMaxDynamicSharedMemorySize, T128_LDS);
  hipFuncSetAttribute((const void*)gemm_fp6_t128_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, F6_STAGE);
  const unsigned cus = gemm_grid(1 << 30);
  const unsigned grid16 = std::min<unsigned>(2 * g.tiles_m * g.tiles_n, T128_WGS * cus), grid6 = std::min<unsigned>(f.tiles_m * f.tiles_n, FP6_WGS * cus);
  k16<<<grid16, 256, T128_LDS>>>(g);
  gemm_fp6_t128_kernel<<<grid6, 256, F6_STAGE>>>(f);
  if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed: %s\n", hipGetErrorString(hipGetLastError())); return 1; }
  // ---- accuracy: both against a double sum on sampled rows, and against each other everywhere
  std::vector<float> h16((size_t)M * N), h6((size_t)M * N);
  hipMemcpy(h16.data(), y16, h16.size() * 4, hipMemcpyDeviceToHost); hipMemcpy(h6.data(), y6, h6.size() * 4, hipMemcpyDeviceToHost);
  double rms = 0; for (size_t i = 0; i < h16.size(); i += 97) rms += (double)h16[i] * h16[i]; rms = sqrt(rms / (h16.size() / 97));
  double e16 = 0, e6 = 0, e66 = 0;
  for (int m = 0; m < M; m += 257)
    for (int n = 0; n < N; ++n) {
      double s = 0; for (int k = 0; k < K; ++k) s += (double)q[(size_t)m * K + k] * (double)Wp[(size_t)n * K + k];
      const double bound = 1e-5 * fabs(s) + 1e-5 * rms;
      e16 = fmax(e16, fabs(h16[(size_t)m * N + n] - s) / bound); e6 = fmax(e6, fabs(h6[(size_t)m * N + n] - s) / bound);
    }
  for (size_t i = 0; i < h16.size(); ++i) e66 = fmax(e66, fabs((double)h16[i] - h6[i]) / (1e-5 * fabs((double)h16[i]) + 1e-5 * rms));
  printf("max err / (1e-5 |y| + 1e-5 rms): f16 limbs vs double %.3f, FP6 digits vs double %.3f (sampled rows); FP6 vs f16 limbs, every output %.3f\n", e16, e6, e66);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int rep = 0; rep < 3; ++rep) {
    float ms16, ms6;
    for (int i = 0; i < 10; ++i) k16<<<grid16, 256, T128_LDS>>>(g);
    hipEventRecord(a); for (int i = 0; i < 100; ++i) k16<<<grid16, 256, T128_LDS>>>(g); hipEventRecord(b); hipEventSynchronize(b); hipEventElapsedTime(&ms16, a, b);
    for (int i = 0; i < 10; ++i) gemm_fp6_t128_kernel<<<grid6, 256, F6_STAGE>>>(f);
    hipEventRecord(a); for (int i = 0; i < 100; ++i) gemm_fp6_t128_kernel<<<grid6, 256, F6_STAGE>>>(f); hipEventRecord(b); hipEventSynchronize(b); hipEventElapsedTime(&ms6, a, b);
    printf("base contraction 8192 x 768 x 3072 (no LoRA stages): f16 limbs (2 MFMA 16x16x32 per 32 k) %.1f us | FP6 digits (%d MFMA 16x16x128 per 128 k, %d workgroups per CU) %.1f us\n",
           ms16 * 10.f, FP6_NL, FP6_WGS, ms6 * 10.f);
  }
  return hipGetLastError() == hipSuccess ? 0 : 1;
}
