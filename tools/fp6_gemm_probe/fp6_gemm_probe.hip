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
#ifndef RING_ONLY
#include "../../llm-qat-on-gpt2_amd/csrc/spq_f16x2.hip"
namespace spq {
void set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fprintf(stderr, "\n"); }
int check_launch(const char* what) { hipError_t e = hipGetLastError(); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", what, hipGetErrorString(e)); return -2; } return 0; }
}
#else       // -DRING_ONLY: the ring kernel alone (diagnostic builds compile in seconds instead of minutes); no comparison kernels
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
namespace spq {
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int EPI_WAVE = 16 * 144, GM = 256, GN = 128;
#define T128_GROUP_M 8
#define T128_WGS 3
__device__ __forceinline__ void glds16(const void* gsrc, void* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc, (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}
unsigned gemm_grid(int ntiles) { int n = 0; (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, 0); if (n < 8) n = 256; return (unsigned)(ntiles < n ? ntiles : n); }
}
#endif
namespace spq {

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
  // LoRA-up stages (gemm_fp6_t128_kernel only; Rp = 0: none): the production kernel's fp16 limb operands, [Mp, Rp] / [Np, Rp], Rp % 64 == 0
  const _Float16 *thi, *tlo, *Bhi, *Blo; const float* rowinv; int Rp;
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
    {   // (wave-uniform by construction; say so to the compiler: the "s" constraint needs scalar registers)
      const unsigned long long a64 = (unsigned long long)sbase;
      const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a64), hi = __builtin_amdgcn_readfirstlane((unsigned)(a64 >> 32));
      sbase = (const unsigned char*)(((unsigned long long)hi << 32) | lo);
      lds_addr = __builtin_amdgcn_readfirstlane(lds_addr);
    }
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
#ifndef RING_ONLY
  // ---- LoRA-up stages: the production kernel's (fp16 limbs, 64-deep, [A 16 KB | B-hi 16 KB | B-lo 16 KB] with its XOR swizzle, v_mfma_f32_16x16x32_f16
  // into the SAME accumulators: the C/D layout does not depend on the operand type)
  const int nlb = g.Rp / 64;                                  // 64-deep blocks of the rank
  const int prow8 = lane >> 3, pchunk = lane & 7;
  auto issue_lora = [&](int j, int which, int tbm, int tbn) {
    const _Float16* A = (which ? g.tlo : g.thi) + (int64_t)tbm * g.Rp + j * 64;
    const _Float16* Bh = g.Bhi + (int64_t)tbn * g.Rp + j * 64;
    const _Float16* Bl = g.Blo + (int64_t)tbn * g.Rp + j * 64;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = (4 * w + i) * 8 + prow8, col = swz(row, pchunk) * 8;
      glds16(A + (row * g.Rp + col), smem + (4 * w + i) * 1024);
      if (!which) { glds16(Bh + (row * g.Rp + col), smem + 16384 + (4 * w + i) * 1024); glds16(Bl + (row * g.Rp + col), smem + 32768 + (4 * w + i) * 1024); }
    }
  };
  auto lora_stage = [&](bool two) {                          // one operand set live at a time (the stage is 2 of 8: registers matter more than overlap here)
    const int sx7 = (l15 >> 1) & 7;
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      const int koff = ((4 * s2 + q4) ^ sx7) * 16;
      f16x8 a[4], bb[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        a[t] = *reinterpret_cast<const f16x8*>(smem + (wm * 64 + l15) * 128 + t * 2048 + koff);
        bb[t] = *reinterpret_cast<const f16x8*>(smem + 16384 + (wn * 64 + l15) * 128 + t * 2048 + koff);
      }
#pragma unroll
      for (int tm = 0; tm < 4; ++tm)
#pragma unroll
        for (int tn = 0; tn < 4; ++tn) acc[tm][tn] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[tm], bb[tn], acc[tm][tn], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (two) {
#pragma unroll
        for (int t = 0; t < 4; ++t) bb[t] = *reinterpret_cast<const f16x8*>(smem + 32768 + (wn * 64 + l15) * 128 + t * 2048 + koff);
#pragma unroll
        for (int tm = 0; tm < 4; ++tm)
#pragma unroll
          for (int tn = 0; tn < 4; ++tn) acc[tm][tn] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[tm], bb[tn], acc[tm][tn], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  };
  if (nlb > 0) issue_lora(0, 0, bm, bn); else
#endif
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
#ifndef RING_ONLY
    for (int j = 0; j < nlb; ++j) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __syncthreads();
      lora_stage(true);                                      // thi x {Bhi, Blo}
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory");
      issue_lora(j, 1, bm, bn);                              // tlo; its B-hi is still in the buffer
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __syncthreads();
      lora_stage(false);                                     // tlo x Bhi
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory");
      if (j + 1 < nlb) issue_lora(j + 1, 0, bm, bn); else issue(0, bm, bn);
    }
    if (nlb > 0) {                                           // LoRA partial sums (units 2^e16[n] 2^g[m]) -> units of the digit sum: * 2^-g[m] * 2^(E6 - e16) = rowinv[m] * 2^10
#pragma unroll
      for (int tm = 0; tm < 4; ++tm) {
        const f32x4 riv = *reinterpret_cast<const f32x4*>(g.rowinv + bm + wm * 64 + tm * 16 + 4 * q4);
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int tn = 0; tn < 4; ++tn) acc[tm][tn][e] *= riv[e] * 1024.f;
      }
    }
#endif
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
      // one fragment set, plane after plane (two sets overlapped reads and MFMAs but cost 256 VGPRs and scratch: the CU's other workgroup covers the reads)
#pragma unroll
      for (int pl = 0; pl < FP6_NL; ++pl) { load_plane(b0, pl); mfma_plane(b0, pl); asm volatile("" ::: "memory"); }
      (void)b1;
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
#ifndef RING_ONLY
    if (nlb > 0) issue_lora(0, 0, nbm, nbn); else
#endif
    issue(0, nbm, nbn);
    p = pn; bm = nbm; bn = nbn;
  }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// The transplant with EIGHT waves per workgroup (2 x 4 of 64 x 32 outputs): same 128 x 128 tile, same 72-KB stage, two workgroups per
// CU = 16 waves per CU, so that one workgroup's eight waves can run the matrix pipe at two waves per SIMD while the other waits for
// its copies (with four waves per workgroup the computing workgroup is one wave per SIMD: 72 % of the pipe at best).
// ---------------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(512, 4) void gemm_fp6_w8_kernel(GemmFp6Args g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = w >> 2, wn = w & 3;
  const int l15 = lane & 15, q4 = lane >> 4;
  const int nwg = g.tiles_m * g.tiles_n;
  const int KB = g.K / 128;
  const int gstride = (int)gridDim.x;
  auto tile_of = [&](int p, int& bm, int& bn) {
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
  const int64_t plane_stride = (int64_t)(g.tiles_n * 4) * KB * F6_PAIR;
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  const unsigned voff = (unsigned)lane * 16u;
  auto glds = [&](const unsigned char* sbase, unsigned lds_addr) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(voff), "s"(sbase), "s"(lds_addr) : "memory");
  };
  // 72 pieces of 1 KB per stage (6 planes x 12), wave w takes pieces 9w .. 9w+8
  auto issue = [&](int kb, int tbm, int tbn) {
    const unsigned char* a_src = g.A6 + ((int64_t)(tbm / 32) * KB + kb) * F6_PAIR;
    const unsigned char* w_src = g.W6 + ((int64_t)(tbn / 32) * KB + kb) * F6_PAIR;
#pragma unroll
    for (int i = 0; i < 9; ++i) {
      const int idx = 9 * w + i, plane = idx / 12, piece = idx - plane * 12, pair = piece / 3, sub = piece - pair * 3;
      const int64_t off = (int64_t)pair * KB * F6_PAIR + sub * 1024;
      const unsigned char* src = plane == 0 ? a_src + off : w_src + (int64_t)(plane - 1) * plane_stride + off;
      glds(src, lds0 + (unsigned)(plane * F6_PLANE + piece * 1024));
    }
  };
  auto frag = [&](int plane_off, int r) -> v8i {
    const char* pb = smem + plane_off + (r >> 1) * F6_PAIR;
    const uint4 a = *reinterpret_cast<const uint4*>(pb + (r & 1) * 1024 + lane * 16);
    const uint2 b = *reinterpret_cast<const uint2*>(pb + 2048 + (r & 1) * 512 + lane * 8);
    v8i v; v[0] = (int)a.x; v[1] = (int)a.y; v[2] = (int)a.z; v[3] = (int)a.w; v[4] = (int)b.x; v[5] = (int)b.y; v[6] = 0; v[7] = 0;
    return v;
  };
  f32x4 acc[4][2];
  issue(0, bm, bn);
  while (true) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[i][j][e] = 0.f;
    const int pn = p + gstride;
    const bool more = pn < nwg;
    int nbm = 0, nbn = 0;
    if (more) tile_of(pn, nbm, nbn);
    for (int kb = 0; kb < KB; ++kb) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();                                       // the stage has landed
      v8i fa[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) fa[t] = frag(0, 4 * wm + t);
#pragma unroll
      for (int pl = 0; pl < FP6_NL; ++pl) {
        v8i fb[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) fb[t] = frag((1 + pl) * F6_PLANE, 2 * wn + t);
        const int sb = 127 + 3 + 5 * pl;
#pragma unroll
        for (int tm = 0; tm < 4; ++tm)
#pragma unroll
          for (int tn = 0; tn < 2; ++tn)
            acc[tm][tn] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(fa[tm], fb[tn], acc[tm][tn], 2, 2, 0, 127, 0, sb);
        __builtin_amdgcn_sched_barrier(0);                 // (keeps hipcc from hoisting every plane's reads: 88 B of scratch otherwise)
        asm volatile("" ::: "memory");
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory");   // every wave has read its fragments
      if (kb + 1 < KB) issue(kb + 1, bm, bn);
    }
    float4 ep_rs, ep_bv;
    {
      const int n = bn + wn * 32 + (lane & 7) * 4;
      ep_rs = make_float4(0.f, 0.f, 0.f, 0.f); ep_bv = ep_rs;
      if (n < g.N) { ep_rs = *reinterpret_cast<const float4*>(g.rowscale + n); if (g.bias) ep_bv = *reinterpret_cast<const float4*>(g.bias + n); }
    }
    {
      char* eb = smem + w * EPI_WAVE;
      const int c4 = (lane & 7) * 4;
      const int n = bn + wn * 32 + c4;
#pragma unroll
      for (int tm = 0; tm < 4; ++tm) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          *reinterpret_cast<float*>(eb + (4 * q4 + e) * 144 + l15 * 4) = acc[tm][0][e];
          *reinterpret_cast<float*>(eb + (4 * q4 + e) * 144 + (16 + l15) * 4) = acc[tm][1][e];
        }
#pragma unroll
        for (int it = 0; it < 2; ++it) {
          const int r16 = it * 8 + (lane >> 3);
          const float4 v = *reinterpret_cast<const float4*>(eb + r16 * 144 + c4 * 4);
          const int m = bm + wm * 64 + tm * 16 + r16;
          float4 o;
          o.x = v.x * ep_rs.x + ep_bv.x; o.y = v.y * ep_rs.y + ep_bv.y; o.z = v.z * ep_rs.z + ep_bv.z; o.w = v.w * ep_rs.w + ep_bv.w;
          if (n < g.N && m < g.M) *reinterpret_cast<float4*>(g.y + (int64_t)m * g.N + n) = o;
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

// ---------------------------------------------------------------------------------------------------------------------------------
// The transplant on 128 x 64 tiles: 42 KB of LDS per workgroup (A 12 KB + five planes of 64 rows, 6 KB each), so THREE workgroups
// share a CU as in the production kernel (12 waves: three per SIMD); 4 waves of 64 x 32 outputs.  More copy bytes (774 MB against 663)
// for more overlap.
// ---------------------------------------------------------------------------------------------------------------------------------
constexpr int N64_PLANE = 2 * F6_PAIR;                            // 64 rows x 128 k: 6 KB
constexpr int N64_STAGE = F6_PLANE + FP6_NL * N64_PLANE;          // 42 KB
__global__ __launch_bounds__(256, 3) void gemm_fp6_n64_kernel(GemmFp6Args g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = w >> 1, wn = w & 1;
  const int l15 = lane & 15, q4 = lane >> 4;
  const int tiles_n = g.tiles_n * 2;                              // 64-column tiles
  const int nwg = g.tiles_m * tiles_n;
  const int KB = g.K / 128;
  const int gstride = (int)gridDim.x;
  auto tile_of = [&](int p, int& bm, int& bn) {
    const int q8 = nwg >> 3, r8 = nwg & 7, xcd = p & 7;
    const int wgid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (p >> 3);
    constexpr int GROUP_M = T128_GROUP_M;
    const int band = wgid / (GROUP_M * tiles_n);
    const int band_rows = min(GROUP_M, g.tiles_m - band * GROUP_M);
    const int in_band = wgid - band * GROUP_M * tiles_n;
    bm = (band * GROUP_M + in_band % band_rows) * 128;
    bn = (in_band / band_rows) * 64;
  };
  int p = blockIdx.x;
  if (p >= nwg) return;
  int bm, bn;
  tile_of(p, bm, bn);
  const int64_t plane_stride = (int64_t)(g.tiles_n * 4) * KB * F6_PAIR;
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  const unsigned voff = (unsigned)lane * 16u;
  auto glds = [&](const unsigned char* sbase, unsigned lds_addr) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(voff), "s"(sbase), "s"(lds_addr) : "memory");
  };
  // 42 pieces of 1 KB per stage: A 12 (4 pairs x 3), each plane 6 (2 pairs x 3); piece idx = w + 4 i
  auto issue = [&](int kb, int tbm, int tbn) {
    const unsigned char* a_src = g.A6 + ((int64_t)(tbm / 32) * KB + kb) * F6_PAIR;
    const unsigned char* w_src = g.W6 + ((int64_t)(tbn / 32) * KB + kb) * F6_PAIR;
#pragma unroll
    for (int i = 0; i < 11; ++i) {
      const int idx = w + 4 * i;
      if (idx >= 42) break;
      const unsigned char* src; unsigned dst;
      if (idx < 12) { const int pair = idx / 3, sub = idx - pair * 3; src = a_src + (int64_t)pair * KB * F6_PAIR + sub * 1024; dst = (unsigned)(idx * 1024); }
      else {
        const int j = idx - 12, plane = j / 6, piece = j - plane * 6, pair = piece / 3, sub = piece - pair * 3;
        src = w_src + (int64_t)plane * plane_stride + (int64_t)pair * KB * F6_PAIR + sub * 1024;
        dst = (unsigned)(F6_PLANE + plane * N64_PLANE + piece * 1024);
      }
      glds(src, lds0 + dst);
    }
  };
  auto frag = [&](int plane_off, int r) -> v8i {
    const char* pb = smem + plane_off + (r >> 1) * F6_PAIR;
    const uint4 a = *reinterpret_cast<const uint4*>(pb + (r & 1) * 1024 + lane * 16);
    const uint2 b = *reinterpret_cast<const uint2*>(pb + 2048 + (r & 1) * 512 + lane * 8);
    v8i v; v[0] = (int)a.x; v[1] = (int)a.y; v[2] = (int)a.z; v[3] = (int)a.w; v[4] = (int)b.x; v[5] = (int)b.y; v[6] = 0; v[7] = 0;
    return v;
  };
  f32x4 acc[4][2];
  int prio_ctr = (int)(blockIdx.x / (gridDim.x / 3 > 0 ? gridDim.x / 3 : 1));
  int stage_ctr = 0;
  issue(0, bm, bn);
  while (true) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[i][j][e] = 0.f;
    const int pn = p + gstride;
    const bool more = pn < nwg;
    int nbm = 0, nbn = 0;
    if (more) tile_of(pn, nbm, nbn);
    for (int kb = 0; kb < KB; ++kb) {
      {                                                      // the CU's three workgroups take the issue priorities in turn (as the production kernel)
        const int per = max(1, (KB * ((nwg + gstride - 1) / gstride) + 5) / 6);
        if (stage_ctr % per == 0) { const int pr = (prio_ctr + stage_ctr / per) % 3; if (pr == 0) __builtin_amdgcn_s_setprio(0); else if (pr == 1) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(2); }
        ++stage_ctr;
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      v8i fa[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) fa[t] = frag(0, 4 * wm + t);
#pragma unroll
      for (int pl = 0; pl < FP6_NL; ++pl) {
        v8i fb[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) fb[t] = frag(F6_PLANE + pl * N64_PLANE, 2 * wn + t);
        const int sb = 127 + 3 + 5 * pl;
#pragma unroll
        for (int tm = 0; tm < 4; ++tm)
#pragma unroll
          for (int tn = 0; tn < 2; ++tn)
            acc[tm][tn] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(fa[tm], fb[tn], acc[tm][tn], 2, 2, 0, 127, 0, sb);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory");
      if (kb + 1 < KB) issue(kb + 1, bm, bn);
    }
    float4 ep_rs, ep_bv;
    {
      const int n = bn + wn * 32 + (lane & 7) * 4;
      ep_rs = make_float4(0.f, 0.f, 0.f, 0.f); ep_bv = ep_rs;
      if (n < g.N) { ep_rs = *reinterpret_cast<const float4*>(g.rowscale + n); if (g.bias) ep_bv = *reinterpret_cast<const float4*>(g.bias + n); }
    }
    {
      char* eb = smem + w * EPI_WAVE;
      const int c4 = (lane & 7) * 4;
      const int n = bn + wn * 32 + c4;
#pragma unroll
      for (int tm = 0; tm < 4; ++tm) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          *reinterpret_cast<float*>(eb + (4 * q4 + e) * 144 + l15 * 4) = acc[tm][0][e];
          *reinterpret_cast<float*>(eb + (4 * q4 + e) * 144 + (16 + l15) * 4) = acc[tm][1][e];
        }
#pragma unroll
        for (int it = 0; it < 2; ++it) {
          const int r16 = it * 8 + (lane >> 3);
          const float4 v = *reinterpret_cast<const float4*>(eb + r16 * 144 + c4 * 4);
          const int m = bm + wm * 64 + tm * 16 + r16;
          float4 o;
          o.x = v.x * ep_rs.x + ep_bv.x; o.y = v.y * ep_rs.y + ep_bv.y; o.z = v.z * ep_rs.z + ep_bv.z; o.w = v.w * ep_rs.w + ep_bv.w;
          if (n < g.N && m < g.M) *reinterpret_cast<float4*>(g.y + (int64_t)m * g.N + n) = o;
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

// ---------------------------------------------------------------------------------------------------------------------------------
// The transplant with its 72-KB stage cut into TWO HALF STAGES that double-buffer each other: X = {A, planes 0, 1} (36 KB),
// Y = {planes 2, 3, 4} (36 KB).  While a workgroup multiplies X of k block kb, the copies of Y(kb) are in flight; while it multiplies
// Y(kb), those of X(kb + 1) -- counted vmcnt, the A fragments stay in registers across both halves.  Same 128 x 128 tile, 4 waves,
// two workgroups per CU, same bytes (663 MB) as the transplant.
// ---------------------------------------------------------------------------------------------------------------------------------
constexpr int HS_HALF = 3 * F6_PLANE;                             // 36 KB
__global__ __launch_bounds__(256, 2) void gemm_fp6_hs_kernel(GemmFp6Args g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = w >> 1, wn = w & 1;
  const int l15 = lane & 15, q4 = lane >> 4;
  const int nwg = g.tiles_m * g.tiles_n;
  const int KB = g.K / 128;
  const int gstride = (int)gridDim.x;
  auto tile_of = [&](int p, int& bm, int& bn) {
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
  const int64_t plane_stride = (int64_t)(g.tiles_n * 4) * KB * F6_PAIR;
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  const unsigned voff = (unsigned)lane * 16u;
  auto glds = [&](const unsigned char* sbase, unsigned lds_addr) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(voff), "s"(sbase), "s"(lds_addr) : "memory");
  };
  // a half stage = three 12-KB planes = 36 pieces of 1 KB; wave w takes pieces 3w..3w+2 of each plane (9 copies per wave and half)
  auto issue_half = [&](int half, int kb, int tbm, int tbn) {       // half 0: A + planes 0, 1; half 1: planes 2, 3, 4
    const unsigned char* a_src = g.A6 + ((int64_t)(tbm / 32 + w) * KB + kb) * F6_PAIR;      // pair w of the tile's rows
    const unsigned char* w_src = g.W6 + ((int64_t)(tbn / 32 + w) * KB + kb) * F6_PAIR;
#pragma unroll
    for (int j = 0; j < 3; ++j) {                                     // plane j of the half
      const unsigned char* src = (half == 0 && j == 0) ? a_src : w_src + (int64_t)(half == 0 ? j - 1 : 2 + j) * plane_stride;
      const unsigned dst = lds0 + (unsigned)(half * HS_HALF + j * F6_PLANE + 3 * w * 1024);
      glds(src, dst); glds(src + 1024, dst + 1024u); glds(src + 2048, dst + 2048u);
    }
  };
  auto frag = [&](int off, int r) -> v8i {
    const char* pb = smem + off + (r >> 1) * F6_PAIR;
    const uint4 a = *reinterpret_cast<const uint4*>(pb + (r & 1) * 1024 + lane * 16);
    const uint2 b = *reinterpret_cast<const uint2*>(pb + 2048 + (r & 1) * 512 + lane * 8);
    v8i v; v[0] = (int)a.x; v[1] = (int)a.y; v[2] = (int)a.z; v[3] = (int)a.w; v[4] = (int)b.x; v[5] = (int)b.y; v[6] = 0; v[7] = 0;
    return v;
  };
  f32x4 acc[4][4];
  v8i fa[4], fb[4];
  auto plane = [&](int off, int pl) {
#pragma unroll
    for (int t = 0; t < 4; ++t) fb[t] = frag(off, 4 * wn + t);
    const int sb = 127 + 3 + 5 * pl;
#pragma unroll
    for (int tm = 0; tm < 4; ++tm)
#pragma unroll
      for (int tn = 0; tn < 4; ++tn)
        acc[tm][tn] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(fa[tm], fb[tn], acc[tm][tn], 2, 2, 0, 127, 0, sb);
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("" ::: "memory");
  };
  issue_half(0, 0, bm, bn);
  issue_half(1, 0, bm, bn);
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
#pragma unroll 1
    for (int kb = 0; kb < KB; ++kb) {
      const bool last = kb + 1 == KB;
      asm volatile("s_waitcnt vmcnt(9)" ::: "memory");         // X(kb) has landed; the nine copies of Y(kb) may still fly
      __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory");
#pragma unroll
      for (int t = 0; t < 4; ++t) fa[t] = frag(0, 4 * wm + t);
      plane(F6_PLANE, 0); plane(2 * F6_PLANE, 1);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory");   // every wave is done with X
      if (!last) issue_half(0, kb + 1, bm, bn);
      if (!last) asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // Y(kb) has landed
      __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory");
      plane(HS_HALF, 2); plane(HS_HALF + F6_PLANE, 3); plane(HS_HALF + 2 * F6_PLANE, 4);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory");   // every wave is done with Y
      if (!last) issue_half(1, kb + 1, bm, bn);
      else if (more) issue_half(1, 0, nbm, nbn);                // the next tile's Y half under the epilogue (its X half after it: the slices live there)
    }
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
            if (n < g.N && m < g.M) *reinterpret_cast<float4*>(g.y + (int64_t)m * g.N + n) = o;
          }
        }
      }
    }
    if (!more) break;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory");     // the slices are done with
    // the epilogue's stores are ordinary VMEM operations: they sit in the same in-order vmcnt as the Y copies issued before them
    issue_half(0, 0, nbm, nbn);
    p = pn; bm = nbm; bn = nbn;
  }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// The production kernel's recipe on FP6 operands: 128 x 128 tiles, 4 waves, ONE 36-KB buffer through which the half stages X = {A, planes
// 0, 1} and Y = {planes 2, 3, 4} of every k block pass in turn (wait + barrier, multiply, barrier, issue the next half), the A fragments kept
// in registers from X to Y -- so that THREE workgroups share a CU and fall out of step, as the f16-limb kernel's three do.
// ---------------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 3) void gemm_fp6_h3_kernel(GemmFp6Args g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = w >> 1, wn = w & 1;
  const int l15 = lane & 15, q4 = lane >> 4;
  const int nwg = g.tiles_m * g.tiles_n;
  const int KB = g.K / 128;
  const int gstride = (int)gridDim.x;
  auto tile_of = [&](int p, int& bm, int& bn) {
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
  const int64_t plane_stride = (int64_t)(g.tiles_n * 4) * KB * F6_PAIR;
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  const unsigned voff = (unsigned)lane * 16u;
  auto glds = [&](const unsigned char* sbase, unsigned lds_addr) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(voff), "s"(sbase), "s"(lds_addr) : "memory");
  };
  // a half stage = three 12-KB planes = 36 pieces of 1 KB; wave w copies the three pieces of pair w of each plane.  SIX running source pointers
  // (A and the five planes, pair w of the current tile), + 3 KB per k block: per-copy 64-bit address arithmetic made hipcc spill 344 B per lane.
  const unsigned char* src6[6];
  auto set_tile = [&](int tbm, int tbn) {
    src6[0] = g.A6 + ((int64_t)(tbm / 32 + w) * KB) * F6_PAIR;
#pragma unroll
    for (int pl = 0; pl < 5; ++pl) src6[1 + pl] = g.W6 + (int64_t)pl * plane_stride + ((int64_t)(tbn / 32 + w) * KB) * F6_PAIR;
  };
  auto issue_x = [&]() {                                       // half X of the NEXT k block of the running pointers: A, planes 0, 1
#pragma unroll
    for (int j = 0; j < 3; ++j) { const unsigned dst = lds0 + (unsigned)(j * F6_PLANE + 3 * w * 1024); glds(src6[j], dst); glds(src6[j] + 1024, dst + 1024u); glds(src6[j] + 2048, dst + 2048u); }
  };
  auto issue_y = [&]() {                                       // half Y of the same k block: planes 2, 3, 4; then the pointers advance
#pragma unroll
    for (int j = 0; j < 3; ++j) { const unsigned dst = lds0 + (unsigned)(j * F6_PLANE + 3 * w * 1024); glds(src6[3 + j], dst); glds(src6[3 + j] + 1024, dst + 1024u); glds(src6[3 + j] + 2048, dst + 2048u); }
#pragma unroll
    for (int j = 0; j < 6; ++j) src6[j] += F6_PAIR;
  };
  // ---- hand-allocated fragment registers: one asm block per plane.  v[120:143] hold the four 6-register weight operands of the plane (fixed, declared
  // clobbered in every block, so hipcc keeps out of them), the sixteen accumulators are tied operands (in place), the four activation operands inputs.
  typedef int v6i __attribute__((ext_vector_type(6)));
  v6i fa[4];
  f32x4 acc[4][4];
  const unsigned a16_0 = lds0 + (unsigned)(2 * wn * F6_PAIR + lane * 16), a8_0 = lds0 + (unsigned)(2 * wn * F6_PAIR + 2048 + lane * 8);
  const int scale_a = 127;
#define F6_MFMA(tm, tn, breg) "v_mfma_scale_f32_16x16x128_f8f6f4 %[c" #tm #tn "], %[a" #tm "], " breg ", %[c" #tm #tn "], %[sa], %[sb] op_sel_hi:[0,0,0] cbsz:2 blgp:2\n\t"
#define F6_ROW0(tm) F6_MFMA(tm, 0, "v[120:125]") F6_MFMA(tm, 1, "v[126:131]") F6_MFMA(tm, 2, "v[132:137]") F6_MFMA(tm, 3, "v[138:143]")
#define F6_ROW1(tm) F6_MFMA(tm, 0, "v[144:149]") F6_MFMA(tm, 1, "v[150:155]") F6_MFMA(tm, 2, "v[156:161]") F6_MFMA(tm, 3, "v[162:167]")
#define F6_CLOB0 "v120", "v121", "v122", "v123", "v124", "v125", "v126", "v127", "v128", "v129", "v130", "v131", "v132", "v133", "v134", "v135", "v136", "v137", "v138", "v139", "v140", "v141", "v142", "v143"
#define F6_CLOB1 "v144", "v145", "v146", "v147", "v148", "v149", "v150", "v151", "v152", "v153", "v154", "v155", "v156", "v157", "v158", "v159", "v160", "v161", "v162", "v163", "v164", "v165", "v166", "v167"
  // TWO fixed fragment sets (v[120:143], v[144:167]): a plane's eight reads are issued one plane ahead of its sixteen MFMAs; both sets are declared
  // clobbered by every block, so hipcc never allocates them and their contents survive from block to block.
  auto rd = [&](int set, int off) {
    const unsigned a16 = a16_0 + (unsigned)off, a8 = a8_0 + (unsigned)off;
    if (set == 0)
      asm volatile("ds_read_b128 v[120:123], %[p16]\n\tds_read_b64 v[124:125], %[p8]\n\tds_read_b128 v[126:129], %[p16] offset:1024\n\tds_read_b64 v[130:131], %[p8] offset:512\n\t"
                   "ds_read_b128 v[132:135], %[p16] offset:3072\n\tds_read_b64 v[136:137], %[p8] offset:3072\n\tds_read_b128 v[138:141], %[p16] offset:4096\n\tds_read_b64 v[142:143], %[p8] offset:3584"
                   :: [p16] "v"(a16), [p8] "v"(a8) : "memory", F6_CLOB0, F6_CLOB1);
    else
      asm volatile("ds_read_b128 v[144:147], %[p16]\n\tds_read_b64 v[148:149], %[p8]\n\tds_read_b128 v[150:153], %[p16] offset:1024\n\tds_read_b64 v[154:155], %[p8] offset:512\n\t"
                   "ds_read_b128 v[156:159], %[p16] offset:3072\n\tds_read_b64 v[160:161], %[p8] offset:3072\n\tds_read_b128 v[162:165], %[p16] offset:4096\n\tds_read_b64 v[166:167], %[p8] offset:3584"
                   :: [p16] "v"(a16), [p8] "v"(a8) : "memory", F6_CLOB0, F6_CLOB1);
  };
#define F6_MM(ROWS, WAIT) asm volatile("s_waitcnt lgkmcnt(" #WAIT ")\n\t" ROWS(0) ROWS(1) ROWS(2) ROWS(3) \
        : [c00] "+v"(acc[0][0]), [c01] "+v"(acc[0][1]), [c02] "+v"(acc[0][2]), [c03] "+v"(acc[0][3]), [c10] "+v"(acc[1][0]), [c11] "+v"(acc[1][1]), [c12] "+v"(acc[1][2]), [c13] "+v"(acc[1][3]), \
          [c20] "+v"(acc[2][0]), [c21] "+v"(acc[2][1]), [c22] "+v"(acc[2][2]), [c23] "+v"(acc[2][3]), [c30] "+v"(acc[3][0]), [c31] "+v"(acc[3][1]), [c32] "+v"(acc[3][2]), [c33] "+v"(acc[3][3]) \
        : [a0] "v"(fa[0]), [a1] "v"(fa[1]), [a2] "v"(fa[2]), [a3] "v"(fa[3]), [sa] "v"(scale_a), [sb] "v"(sb) : "memory", F6_CLOB0, F6_CLOB1)
  auto mm = [&](int set, int pl, bool more_in_flight) {        // more_in_flight: the OTHER set's eight reads were issued after this set's
    const int sb = 127 + 3 + 5 * pl;
    if (set == 0) { if (more_in_flight) F6_MM(F6_ROW0, 8); else F6_MM(F6_ROW0, 0); }
    else { if (more_in_flight) F6_MM(F6_ROW1, 8); else F6_MM(F6_ROW1, 0); }
  };
  auto load_a = [&]() {                                        // the activation operands of the wave's four row tiles (plane 0 of the X half)
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int r = 4 * wm + t;
      const char* pb = smem + (r >> 1) * F6_PAIR;
      const uint4 a = *reinterpret_cast<const uint4*>(pb + (r & 1) * 1024 + lane * 16);
      const uint2 b2 = *reinterpret_cast<const uint2*>(pb + 2048 + (r & 1) * 512 + lane * 8);
      fa[t][0] = (int)a.x; fa[t][1] = (int)a.y; fa[t][2] = (int)a.z; fa[t][3] = (int)a.w; fa[t][4] = (int)b2.x; fa[t][5] = (int)b2.y;
    }
  };
  set_tile(bm, bn);
  issue_x();
  int prio_ctr = (int)(blockIdx.x / (gridDim.x / 3 > 0 ? gridDim.x / 3 : 1));
  int stage_ctr = 0;
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
#pragma unroll 1
    for (int kb = 0; kb < KB; ++kb) {
      {                                                      // the CU's three workgroups take the issue priorities in turn (as the production kernel)
        const int per = max(1, (KB * ((nwg + gstride - 1) / gstride) + 5) / 6);
        if (stage_ctr % per == 0) { const int pr = (prio_ctr + stage_ctr / per) % 3; if (pr == 0) __builtin_amdgcn_s_setprio(0); else if (pr == 1) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(2); }
        ++stage_ctr;
      }
      const bool last = kb + 1 == KB;
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // X(kb) has landed
      __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory");
      load_a();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // (the activation operands: compiler-managed reads; the counted waits below start from zero)
      rd(0, F6_PLANE); rd(1, 2 * F6_PLANE); mm(0, 0, true); mm(1, 1, false);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory");   // every wave is done with X
      issue_y();
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // Y(kb) has landed
      __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory");
      rd(0, 0); rd(1, F6_PLANE); mm(0, 2, true); rd(0, 2 * F6_PLANE); mm(1, 3, true); mm(0, 4, false);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory");   // every wave is done with Y
      if (!last) issue_x();
    }
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
            if (n < g.N && m < g.M) *reinterpret_cast<float4*>(g.y + (int64_t)m * g.N + n) = o;
          }
        }
      }
    }
    if (!more) break;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory");     // the slices are done with
    // the epilogue's stores are ordinary VMEM operations: they sit in the same in-order vmcnt as the Y copies issued before them
    set_tile(nbm, nbn);
    issue_x();
    p = pn; bm = nbm; bn = nbn;
  }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// The half-stage double buffer again (X = {A, planes 0, 1} and Y = {planes 2, 3, 4} in two 36-KB buffers, the copies of one in flight under the MFMAs of
// the other, two workgroups per CU) -- with the hand-allocated plane blocks of the kernel above instead of hipcc's fragment handling.
// ---------------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 2) void gemm_fp6_hs2_kernel(GemmFp6Args g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = w >> 1, wn = w & 1;
  const int l15 = lane & 15, q4 = lane >> 4;
  const int nwg = g.tiles_m * g.tiles_n;
  const int KB = g.K / 128;
  const int gstride = (int)gridDim.x;
  auto tile_of = [&](int p, int& bm, int& bn) {
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
  const int64_t plane_stride = (int64_t)(g.tiles_n * 4) * KB * F6_PAIR;
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  const unsigned voff = (unsigned)lane * 16u;
  auto glds = [&](const unsigned char* sbase, unsigned lds_addr) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(voff), "s"(sbase), "s"(lds_addr) : "memory");
  };
  // a half stage = three 12-KB planes = 36 pieces of 1 KB; wave w copies the three pieces of pair w of each plane.  SIX running source pointers
  // (A and the five planes, pair w of the current tile), + 3 KB per k block: per-copy 64-bit address arithmetic made hipcc spill 344 B per lane.
  const unsigned char* src6[6];
  auto set_tile = [&](int tbm, int tbn) {
    src6[0] = g.A6 + ((int64_t)(tbm / 32 + w) * KB) * F6_PAIR;
#pragma unroll
    for (int pl = 0; pl < 5; ++pl) src6[1 + pl] = g.W6 + (int64_t)pl * plane_stride + ((int64_t)(tbn / 32 + w) * KB) * F6_PAIR;
  };
  auto issue_x = [&]() {                                       // half X of the NEXT k block of the running pointers: A, planes 0, 1
#pragma unroll
    for (int j = 0; j < 3; ++j) { const unsigned dst = lds0 + (unsigned)(j * F6_PLANE + 3 * w * 1024); glds(src6[j], dst); glds(src6[j] + 1024, dst + 1024u); glds(src6[j] + 2048, dst + 2048u); }
#pragma unroll
    for (int j = 0; j < 3; ++j) src6[j] += F6_PAIR;
  };
  auto issue_y = [&]() {                                       // half Y of the same k block: planes 2, 3, 4; then the pointers advance
#pragma unroll
    for (int j = 0; j < 3; ++j) { const unsigned dst = lds0 + (unsigned)(HS_HALF + j * F6_PLANE + 3 * w * 1024); glds(src6[3 + j], dst); glds(src6[3 + j] + 1024, dst + 1024u); glds(src6[3 + j] + 2048, dst + 2048u); }
#pragma unroll
    for (int j = 3; j < 6; ++j) src6[j] += F6_PAIR;
  };
  // ---- hand-allocated fragment registers: one asm block per plane.  v[120:143] hold the four 6-register weight operands of the plane (fixed, declared
  // clobbered in every block, so hipcc keeps out of them), the sixteen accumulators are tied operands (in place), the four activation operands inputs.
  typedef int v6i __attribute__((ext_vector_type(6)));
  v6i fa[4];
  f32x4 acc[4][4];
  const unsigned a16_0 = lds0 + (unsigned)(2 * wn * F6_PAIR + lane * 16), a8_0 = lds0 + (unsigned)(2 * wn * F6_PAIR + 2048 + lane * 8);
  const int scale_a = 127;
#define F6_MFMA(tm, tn, breg) "v_mfma_scale_f32_16x16x128_f8f6f4 %[c" #tm #tn "], %[a" #tm "], " breg ", %[c" #tm #tn "], %[sa], %[sb] op_sel_hi:[0,0,0] cbsz:2 blgp:2\n\t"
#define F6_ROW0(tm) F6_MFMA(tm, 0, "v[120:125]") F6_MFMA(tm, 1, "v[126:131]") F6_MFMA(tm, 2, "v[132:137]") F6_MFMA(tm, 3, "v[138:143]")
#define F6_ROW1(tm) F6_MFMA(tm, 0, "v[144:149]") F6_MFMA(tm, 1, "v[150:155]") F6_MFMA(tm, 2, "v[156:161]") F6_MFMA(tm, 3, "v[162:167]")
#define F6_CLOB0 "v120", "v121", "v122", "v123", "v124", "v125", "v126", "v127", "v128", "v129", "v130", "v131", "v132", "v133", "v134", "v135", "v136", "v137", "v138", "v139", "v140", "v141", "v142", "v143"
#define F6_CLOB1 "v144", "v145", "v146", "v147", "v148", "v149", "v150", "v151", "v152", "v153", "v154", "v155", "v156", "v157", "v158", "v159", "v160", "v161", "v162", "v163", "v164", "v165", "v166", "v167"
  // TWO fixed fragment sets (v[120:143], v[144:167]): a plane's eight reads are issued one plane ahead of its sixteen MFMAs; both sets are declared
  // clobbered by every block, so hipcc never allocates them and their contents survive from block to block.
  auto rd = [&](int set, int off) {
    const unsigned a16 = a16_0 + (unsigned)off, a8 = a8_0 + (unsigned)off;
    if (set == 0)
      asm volatile("ds_read_b128 v[120:123], %[p16]\n\tds_read_b64 v[124:125], %[p8]\n\tds_read_b128 v[126:129], %[p16] offset:1024\n\tds_read_b64 v[130:131], %[p8] offset:512\n\t"
                   "ds_read_b128 v[132:135], %[p16] offset:3072\n\tds_read_b64 v[136:137], %[p8] offset:3072\n\tds_read_b128 v[138:141], %[p16] offset:4096\n\tds_read_b64 v[142:143], %[p8] offset:3584"
                   :: [p16] "v"(a16), [p8] "v"(a8) : "memory", F6_CLOB0, F6_CLOB1);
    else
      asm volatile("ds_read_b128 v[144:147], %[p16]\n\tds_read_b64 v[148:149], %[p8]\n\tds_read_b128 v[150:153], %[p16] offset:1024\n\tds_read_b64 v[154:155], %[p8] offset:512\n\t"
                   "ds_read_b128 v[156:159], %[p16] offset:3072\n\tds_read_b64 v[160:161], %[p8] offset:3072\n\tds_read_b128 v[162:165], %[p16] offset:4096\n\tds_read_b64 v[166:167], %[p8] offset:3584"
                   :: [p16] "v"(a16), [p8] "v"(a8) : "memory", F6_CLOB0, F6_CLOB1);
  };
#define F6_MM(ROWS, WAIT) asm volatile("s_waitcnt lgkmcnt(" #WAIT ")\n\t" ROWS(0) ROWS(1) ROWS(2) ROWS(3) \
        : [c00] "+v"(acc[0][0]), [c01] "+v"(acc[0][1]), [c02] "+v"(acc[0][2]), [c03] "+v"(acc[0][3]), [c10] "+v"(acc[1][0]), [c11] "+v"(acc[1][1]), [c12] "+v"(acc[1][2]), [c13] "+v"(acc[1][3]), \
          [c20] "+v"(acc[2][0]), [c21] "+v"(acc[2][1]), [c22] "+v"(acc[2][2]), [c23] "+v"(acc[2][3]), [c30] "+v"(acc[3][0]), [c31] "+v"(acc[3][1]), [c32] "+v"(acc[3][2]), [c33] "+v"(acc[3][3]) \
        : [a0] "v"(fa[0]), [a1] "v"(fa[1]), [a2] "v"(fa[2]), [a3] "v"(fa[3]), [sa] "v"(scale_a), [sb] "v"(sb) : "memory", F6_CLOB0, F6_CLOB1)
  auto mm = [&](int set, int pl, bool more_in_flight) {        // more_in_flight: the OTHER set's eight reads were issued after this set's
    const int sb = 127 + 3 + 5 * pl;
    if (set == 0) { if (more_in_flight) F6_MM(F6_ROW0, 8); else F6_MM(F6_ROW0, 0); }
    else { if (more_in_flight) F6_MM(F6_ROW1, 8); else F6_MM(F6_ROW1, 0); }
  };
  auto load_a = [&]() {                                        // the activation operands of the wave's four row tiles (plane 0 of the X half)
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int r = 4 * wm + t;
      const char* pb = smem + (r >> 1) * F6_PAIR;
      const uint4 a = *reinterpret_cast<const uint4*>(pb + (r & 1) * 1024 + lane * 16);
      const uint2 b2 = *reinterpret_cast<const uint2*>(pb + 2048 + (r & 1) * 512 + lane * 8);
      fa[t][0] = (int)a.x; fa[t][1] = (int)a.y; fa[t][2] = (int)a.z; fa[t][3] = (int)a.w; fa[t][4] = (int)b2.x; fa[t][5] = (int)b2.y;
    }
  };
  set_tile(bm, bn);
  issue_x();
  issue_y();
  int prio_ctr = (int)(blockIdx.x / (gridDim.x / 2 > 0 ? gridDim.x / 2 : 1));
  int stage_ctr = 0;
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
#pragma unroll 1
    for (int kb = 0; kb < KB; ++kb) {
      {                                                      // the CU's three workgroups take the issue priorities in turn (as the production kernel)
        const int per = max(1, (KB * ((nwg + gstride - 1) / gstride) + 5) / 6);
        if (stage_ctr % per == 0) { const int pr = (prio_ctr + stage_ctr / per) % 2; if (pr == 0) __builtin_amdgcn_s_setprio(0); else __builtin_amdgcn_s_setprio(1); }
        ++stage_ctr;
      }
      const bool last = kb + 1 == KB;
      asm volatile("s_waitcnt vmcnt(9)" ::: "memory");         // X(kb) has landed; the nine copies of Y(kb) may still fly
      __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory");
      load_a();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      rd(0, F6_PLANE); rd(1, 2 * F6_PLANE); mm(0, 0, true); mm(1, 1, false);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory");   // every wave is done with X
      if (!last) issue_x();
      if (!last) asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // Y(kb) has landed
      __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory");
      rd(0, HS_HALF); rd(1, HS_HALF + F6_PLANE); mm(0, 2, true); rd(0, HS_HALF + 2 * F6_PLANE); mm(1, 3, true); mm(0, 4, false);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory");   // every wave is done with Y
      if (!last) issue_y();
    }
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
            if (n < g.N && m < g.M) *reinterpret_cast<float4*>(g.y + (int64_t)m * g.N + n) = o;
          }
        }
      }
    }
    if (!more) break;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory");     // the slices are done with
    // the epilogue's stores are ordinary VMEM operations: they sit in the same in-order vmcnt as the Y copies issued before them
    set_tile(nbm, nbn);
    issue_x();
    issue_y();
    p = pn; bm = nbm; bn = nbn;
  }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// The double-buffered form: 256 x 128 tiles, ONE workgroup of 8 waves per CU (4 x 2 of 64 x 64 outputs), the activation fragments straight
// from global memory into registers (one k block ahead), the five digit planes of a k block (60 KB) in one of TWO LDS buffers: every
// wave issues its share of the next k block's plane copies (60 pieces of 1 KB: 7 or 8 per wave) at the top of a k block and then
// multiplies the current one -- one vmcnt(0) + one barrier per k block, no counters.  The next tile's first k block is in flight
// under the epilogue.
// ---------------------------------------------------------------------------------------------------------------------------------
constexpr int DB_BUF = 5 * F6_PLANE;                              // 60 KB
constexpr int DB_EPI_OFF = 2 * DB_BUF;
constexpr int DB_LDS = DB_EPI_OFF + 8 * EPI_WAVE;                 // 138 KB
__global__ __launch_bounds__(512, 2) void gemm_fp6_db_kernel(GemmFp6Args g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = w >> 1, wn = w & 1;
  const int l15 = lane & 15, q4 = lane >> 4;
  const int tiles_m = g.tiles_m / 2;
  const int nwg = tiles_m * g.tiles_n;
  const int KB = g.K / 128;
  const int gstride = (int)gridDim.x;
  auto tile_of = [&](int p, int& bm, int& bn) {
    const int q8 = nwg >> 3, r8 = nwg & 7, xcd = p & 7;
    const int wgid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (p >> 3);
    constexpr int GROUP_M = 4;
    const int band = wgid / (GROUP_M * g.tiles_n);
    const int band_rows = min(GROUP_M, tiles_m - band * GROUP_M);
    const int in_band = wgid - band * GROUP_M * g.tiles_n;
    bm = (band * GROUP_M + in_band % band_rows) * 256;
    bn = (in_band / band_rows) * 128;
  };
  const int my_tiles = (nwg - (int)blockIdx.x + gstride - 1) / gstride;
  const int64_t plane_stride = (int64_t)(g.tiles_n * 4) * KB * F6_PAIR;
  const unsigned voff = (unsigned)lane * 16u;
  auto glds = [&](const unsigned char* sbase, unsigned lds_addr) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(voff), "s"(sbase), "s"(lds_addr) : "memory");
  };
  // Running source pointers: this wave's (up to) eight plane pieces and its four activation records of the CURRENT tile, + 3 KB per k block
  // (per-copy 64-bit address arithmetic made hipcc spill scalar registers by the dozen).
  const unsigned char* psrc[8]; unsigned pdst[8];
  const unsigned char* asrc16[4]; const unsigned char* asrc8[4];
  auto set_tile = [&](int tbm, int tbn) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int idx = min(w + 8 * i, 59), plane = idx / 12, piece = idx - plane * 12, pair = piece / 3, sub = piece - pair * 3;
      psrc[i] = g.W6 + (int64_t)plane * plane_stride + ((int64_t)(tbn / 32 + pair) * KB) * F6_PAIR + sub * 1024;
      pdst[i] = (unsigned)(plane * F6_PLANE + piece * 1024);
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int rec = 4 * wm + t;
      const unsigned char* pb = g.A6 + ((int64_t)(tbm / 32 + (rec >> 1)) * KB) * F6_PAIR;
      asrc16[t] = pb + (rec & 1) * 1024 + lane * 16; asrc8[t] = pb + 2048 + (rec & 1) * 512 + lane * 8;
    }
  };
  typedef int v6i __attribute__((ext_vector_type(6)));
  auto issue_next = [&](int buf, v6i (&fa)[4]) {                 // the next k block of the current pointers: plane copies + activation fragments; then advance
#pragma unroll
    for (int i = 0; i < 8; ++i) if (w + 8 * i < 60) glds(psrc[i], lds0 + (unsigned)(buf * DB_BUF) + pdst[i]);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const uint4 a = *reinterpret_cast<const uint4*>(asrc16[t]);
      const uint2 b2 = *reinterpret_cast<const uint2*>(asrc8[t]);
      fa[t][0] = (int)a.x; fa[t][1] = (int)a.y; fa[t][2] = (int)a.z; fa[t][3] = (int)a.w; fa[t][4] = (int)b2.x; fa[t][5] = (int)b2.y;
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) psrc[i] += F6_PAIR;
#pragma unroll
    for (int t = 0; t < 4; ++t) { asrc16[t] += F6_PAIR; asrc8[t] += F6_PAIR; }
  };
  f32x4 acc[4][4];
  v6i fa[4], fan[4];
  const unsigned a16_0 = lds0 + (unsigned)(2 * wn * F6_PAIR + lane * 16), a8_0 = lds0 + (unsigned)(2 * wn * F6_PAIR + 2048 + lane * 8);
  const int scale_a = 127;
  auto rd = [&](int set, unsigned off) {                           // the hand-allocated plane blocks of gemm_fp6_h3_kernel
    const unsigned a16 = a16_0 + off, a8 = a8_0 + off;
    if (set == 0)
      asm volatile("ds_read_b128 v[120:123], %[p16]\n\tds_read_b64 v[124:125], %[p8]\n\tds_read_b128 v[126:129], %[p16] offset:1024\n\tds_read_b64 v[130:131], %[p8] offset:512\n\t"
                   "ds_read_b128 v[132:135], %[p16] offset:3072\n\tds_read_b64 v[136:137], %[p8] offset:3072\n\tds_read_b128 v[138:141], %[p16] offset:4096\n\tds_read_b64 v[142:143], %[p8] offset:3584"
                   :: [p16] "v"(a16), [p8] "v"(a8) : "memory", F6_CLOB0, F6_CLOB1);
    else
      asm volatile("ds_read_b128 v[144:147], %[p16]\n\tds_read_b64 v[148:149], %[p8]\n\tds_read_b128 v[150:153], %[p16] offset:1024\n\tds_read_b64 v[154:155], %[p8] offset:512\n\t"
                   "ds_read_b128 v[156:159], %[p16] offset:3072\n\tds_read_b64 v[160:161], %[p8] offset:3072\n\tds_read_b128 v[162:165], %[p16] offset:4096\n\tds_read_b64 v[166:167], %[p8] offset:3584"
                   :: [p16] "v"(a16), [p8] "v"(a8) : "memory", F6_CLOB0, F6_CLOB1);
  };
  auto mm = [&](int set, int pl, bool more_in_flight) {
    const int sb = 127 + 3 + 5 * pl;
    if (set == 0) { if (more_in_flight) F6_MM(F6_ROW0, 8); else F6_MM(F6_ROW0, 0); }
    else { if (more_in_flight) F6_MM(F6_ROW1, 8); else F6_MM(F6_ROW1, 0); }
  };
  int p = blockIdx.x, bm, bn;
  tile_of(p, bm, bn);
  set_tile(bm, bn);
  int buf = 0;
  issue_next(0, fan);
  for (int ti = 0; ti < my_tiles; ++ti) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[i][j][e] = 0.f;
    const int pn = p + gstride;
    const bool more = ti + 1 < my_tiles;
    int nbm = bm, nbn = bn;
    if (more) tile_of(pn, nbm, nbn);
#pragma unroll 1
    for (int kb = 0; kb < KB; ++kb) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // my plane copies of this k block and my A fragments are here
      __syncthreads();                                          // ... and everyone's; everyone is done with the other buffer
#pragma unroll
      for (int t = 0; t < 4; ++t) fa[t] = fan[t];               // (24 register copies per k block instead of two code copies of the whole block)
      const bool last = kb + 1 == KB;
      if (last && more) set_tile(nbm, nbn);
      if (!last || more) issue_next(buf ^ 1, fan);
      {
        const unsigned bo = (unsigned)(buf * DB_BUF);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        rd(0, bo); rd(1, bo + F6_PLANE); mm(0, 0, true); rd(0, bo + 2 * F6_PLANE); mm(1, 1, true); rd(1, bo + 3 * F6_PLANE); mm(0, 2, true);
        rd(0, bo + 4 * F6_PLANE); mm(1, 3, true); mm(0, 4, false);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
      buf ^= 1;
    }
    float4 ep_rs[2], ep_bv[2];
#pragma unroll
    for (int tn = 0; tn < 2; ++tn) {
      const int n = bn + wn * 64 + tn * 32 + (lane & 7) * 4;
      ep_rs[tn] = make_float4(0.f, 0.f, 0.f, 0.f); ep_bv[tn] = ep_rs[tn];
      if (n < g.N) { ep_rs[tn] = *reinterpret_cast<const float4*>(g.rowscale + n); if (g.bias) ep_bv[tn] = *reinterpret_cast<const float4*>(g.bias + n); }
    }
    char* eb = smem + DB_EPI_OFF + w * EPI_WAVE;
    const int c4 = (lane & 7) * 4;
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
        for (int it2 = 0; it2 < 2; ++it2) {
          const int r16 = it2 * 8 + (lane >> 3);
          const float4 v = *reinterpret_cast<const float4*>(eb + r16 * 144 + c4 * 4);
          const int m = bm + wm * 64 + tm * 16 + r16;
          float4 o;
          o.x = v.x * rs.x + bv.x; o.y = v.y * rs.y + bv.y; o.z = v.z * rs.z + bv.z; o.w = v.w * rs.w + bv.w;
          if (n < g.N && m < g.M) *reinterpret_cast<float4*>(g.y + (int64_t)m * g.N + n) = o;
        }
      }
    }
    p = pn; bm = nbm; bn = nbn;
  }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// The ring form: 256 x 128 tiles, 12 waves -- waves 0..7 compute (4 x 2 of 64 x 64 outputs), waves 8..11 only copy.  A 128-deep k
// block is seven ITEMS of 12 KB (A rows 0..127, A rows 128..255, the five digit planes); the LDS holds a ring of RING_NS such slots.
// Loader wave j copies pieces 3j..3j+2 of every item (scalar base + lane offset, LDS-DMA), confirms an item RING_LOOK items later
// with a counted vmcnt and adds 1 to FULL[slot]; a compute wave polls FULL[slot] >= 4 (gen + 1), reads its fragments, and adds 1 to
// FREE[slot] once they are in registers (every compute wave releases every item, also the A half it does not read); the loader
// polls FREE[slot] >= 8 gen before it refills.  Counters are monotonic; every spin is bounded (g.err is set, the result is then wrong
// but the grid drains).  Persistent: tile p, p + grid, ...; the loader runs ahead into the next tile during the epilogue.
// ---------------------------------------------------------------------------------------------------------------------------------
#ifndef RING_NS
#define RING_NS 11
#endif
#ifndef RING_ONESET
#define RING_ONESET 0
#endif
#ifndef RING_HYST
#define RING_HYST 3
#endif
#ifndef RING_B64
#define RING_B64 0
#endif
#ifndef RING_AREG      // 1: the activation fragments go global -> registers in the compute waves; the ring carries the digit planes only
#define RING_AREG 0
#endif
#ifndef RING_SLEEP     // s_sleep argument inside a waiting loop (0: tight polling)
#define RING_SLEEP 0
#endif
#ifndef RING_DIAG     // timing probes (wrong results): 1 no copies, 2 no MFMAs, 4 no fragment reads, 8 loader waves at top issue priority, 16 no stores, 32 no waiting on either side
#define RING_DIAG 0
#endif
#ifndef RING_LOOK
#define RING_LOOK 4
#endif
constexpr int RING_ITEM = F6_PLANE;                              // 12 KB
constexpr int RING_EPI_OFF = RING_NS * RING_ITEM;                // 8 epilogue slices
constexpr int RING_CNT_OFF = RING_EPI_OFF + 8 * EPI_WAVE;        // FULL[NS], FREE[NS]: 16 B apart
constexpr int RING_LDS = RING_CNT_OFF + 64;
static_assert(RING_LDS <= 160 * 1024, "LDS");
static_assert(FP6_NL == 5, "the ring kernel is written for five digit planes");
struct GemmFp6RingArgs { GemmFp6Args f; int* err; unsigned long long* stamps; };

__global__ __launch_bounds__(768, 1) void gemm_fp6_ring_kernel(GemmFp6RingArgs ga) {
  const GemmFp6Args& g = ga.f;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tiles_m = g.tiles_m / 2;                              // 256-row tiles
  const int nwg = tiles_m * g.tiles_n;
  const int KB = g.K / 128;
  const int gstride = (int)gridDim.x;
  auto tile_of = [&](int p, int& bm, int& bn) {
    const int q8 = nwg >> 3, r8 = nwg & 7, xcd = p & 7;
    const int wgid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (p >> 3);
    constexpr int GROUP_M = 4;
    const int band = wgid / (GROUP_M * g.tiles_n);
    const int band_rows = min(GROUP_M, tiles_m - band * GROUP_M);
    const int in_band = wgid - band * GROUP_M * g.tiles_n;
    bm = (band * GROUP_M + in_band % band_rows) * 256;
    bn = (in_band / band_rows) * 128;
  };
  // counters: zero them before anyone polls
  // progress words: FULLV[4] = items each loader wave has landed (in order), FREEV[8] = items each compute wave has released (in
  // order).  A reader takes the minimum of a whole line with ONE 16-B read (two for FREEV) and remembers it: it polls again only when
  // it needs an item beyond what it already knows -- about once per k block, not once per item.
  if (tid < 16) *reinterpret_cast<int*>(smem + RING_CNT_OFF + tid * 4) = 0;
  if (RING_DIAG & 64) for (int i = tid; i < RING_EPI_OFF / 4; i += 768) reinterpret_cast<int*>(smem)[i] = 0;   // zero operands: the data-dependence of the clock
  __syncthreads();
  const unsigned fullv = lds0 + RING_CNT_OFF, freev = lds0 + RING_CNT_OFF + 16;
  bool timed_out = false;
  auto min_line = [&](unsigned addr, int words) -> int {        // min of 4 or 8 progress words
    uint4 v, v2;
    if (words == 8) { asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:16\n\ts_waitcnt lgkmcnt(0)" : "=&v"(v), "=&v"(v2) : "v"(addr) : "memory"); }
    else { asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory"); v2 = v; }
    const unsigned m = min(min(min(v.x, v.y), min(v.z, v.w)), min(min(v2.x, v2.y), min(v2.z, v2.w)));
    return (int)__builtin_amdgcn_readfirstlane(m);
  };
  auto wait_for = [&](int& known, int need, unsigned addr, int words) {      // until min(line) >= need; bounded
    int spins = 0;
    while (known < need && !timed_out && !(RING_DIAG & 32)) {
      known = min_line(addr, words);
      if (known < need) { if (RING_SLEEP) __builtin_amdgcn_s_sleep(RING_SLEEP); if (++spins > (1 << 20)) timed_out = true; }
    }
  };
  auto publish = [&](unsigned addr, int value) { if (lane == 0) asm volatile("ds_write_b32 %0, %1" :: "v"(addr), "v"(value) : "memory"); };
  const int my_tiles = (nwg - (int)blockIdx.x + gstride - 1) / gstride;   // tiles this workgroup walks
  constexpr int IPK = RING_AREG ? 5 : 7;                          // items per k block (RING_AREG: the A fragments go global -> registers, not through the ring)
  const int total_items = my_tiles * KB * IPK;

  if (w >= 8) {
    // ================================================== loader ==================================================
    const int lw = w - 8;
    if (RING_DIAG & 8) __builtin_amdgcn_s_setprio(3);
    const unsigned voff = (unsigned)lane * 16u;
    const int64_t plane_stride = (int64_t)(g.tiles_n * 4) * KB * F6_PAIR;
    int p = blockIdx.x, kb = 0, it = 0, bm, bn;
    tile_of(p, bm, bn);
    auto glds = [&](const unsigned char* sbase, unsigned lds_addr) {
      asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(voff), "s"(sbase), "s"(lds_addr) : "memory");
    };
    // Seven running source pointers (A rows 0..127 / 128..255 and the five planes of the tile's columns), + 3 KB per k block: the
    // loader's scalar stream shares its SIMD with two MFMA-streaming waves and a per-item address computation (64-bit multiplies)
    // cost ~250 cycles per item -- stamped: the compute waves spent a third of their time waiting for items even with NO copies.
    int released = 0;
    const unsigned char* src[7];
    auto set_tile = [&]() {
      if (!RING_AREG) {
        src[5] = g.A6 + ((int64_t)(bm / 32 + lw) * KB) * F6_PAIR;
        src[6] = g.A6 + ((int64_t)(bm / 32 + 4 + lw) * KB) * F6_PAIR;
      }
#pragma unroll
      for (int pl = 0; pl < 5; ++pl) src[pl] = g.W6 + (int64_t)pl * plane_stride + ((int64_t)(bn / 32 + lw) * KB) * F6_PAIR;
    };
    set_tile();
    int gi = 0, slot = 0;
    for (int ti = 0; ti < my_tiles; ++ti) {
      for (kb = 0; kb < KB; ++kb) {
#pragma unroll
        for (int it7 = 0; it7 < IPK; ++it7) {
          it = RING_AREG ? it7 : (it7 < 2 ? 5 + it7 : it7 - 2);               // item order: (A half 0, A half 1,) planes 0..4
          if (gi >= RING_NS && released < gi - RING_NS + 1) {                     // every compute wave must be done with the item this slot held
            released = min_line(freev, 8);
            if (released < gi - RING_NS + 1) {                                     // blocked: confirm everything issued so far instead of idling behind the look-ahead
              asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
              publish(fullv + 4u * (unsigned)lw, gi);
              wait_for(released, gi - RING_NS + 1, freev, 8);
            }
          }
          const unsigned dst = lds0 + (unsigned)slot * RING_ITEM + (unsigned)(3 * lw) * 1024u;
          if (!(RING_DIAG & 1)) { glds(src[it], dst); glds(src[it] + 1024, dst + 1024u); glds(src[it] + 2048, dst + 2048u); }
          src[it] += F6_PAIR;
          if (gi >= RING_LOOK) {                                                    // item gi - LOOK has landed once only the younger 3 LOOK copies are in flight
            asm volatile("s_waitcnt vmcnt(%0)" :: "n"(3 * RING_LOOK) : "memory");
            publish(fullv + 4u * (unsigned)lw, gi - RING_LOOK + 1);
          }
          ++gi; slot = slot == RING_NS - 1 ? 0 : slot + 1;
        }
      }
      p += gstride;
      if (p < nwg) { tile_of(p, bm, bn); set_tile(); }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");               // the last LOOK items
    publish(fullv + 4u * (unsigned)lw, total_items);
    if (timed_out && lane == 0) atomicAdd(ga.err, 1);
    return;
  }
  // ================================================== compute ==================================================
  const int wm = w >> 1, wn = w & 1;
  const int l15 = lane & 15, q4 = lane >> 4;
  const int my_half = wm >> 1;                                     // which A item this wave reads
  auto frag = [&](unsigned item_base, int r) -> v8i {
    const char* pb = smem + item_base + (r >> 1) * F6_PAIR;
#if RING_B64      // three 8-B reads instead of 16 B + 8 B: does hipcc place them into the operand tuple without copies?
    const uint2 a0 = *reinterpret_cast<const uint2*>(pb + (r & 1) * 1024 + lane * 16);
    const uint2 a1 = *reinterpret_cast<const uint2*>(pb + (r & 1) * 1024 + lane * 16 + 8);
    const uint2 b = *reinterpret_cast<const uint2*>(pb + 2048 + (r & 1) * 512 + lane * 8);
    v8i v; v[0] = (int)a0.x; v[1] = (int)a0.y; v[2] = (int)a1.x; v[3] = (int)a1.y; v[4] = (int)b.x; v[5] = (int)b.y; v[6] = 0; v[7] = 0;
    return v;
#else
    const uint4 a = *reinterpret_cast<const uint4*>(pb + (r & 1) * 1024 + lane * 16);
    const uint2 b = *reinterpret_cast<const uint2*>(pb + 2048 + (r & 1) * 512 + lane * 8);
    v8i v; v[0] = (int)a.x; v[1] = (int)a.y; v[2] = (int)a.z; v[3] = (int)a.w; v[4] = (int)b.x; v[5] = (int)b.y; v[6] = 0; v[7] = 0;
    return v;
#endif
  };
  f32x4 acc[4][4];
  int gi = 0, ready = 0, dummy = 0;
  unsigned long long st[6] = {0, 0, 0, 0, 0, 0};      // RING_DIAG & 256: cycles in [0] waiting for items, [1] fragment reads landing, [2] MFMA issue, [3] epilogue, [4] total
#define RSTAMP() ((RING_DIAG & 256) ? __builtin_readcyclecounter() : 0ull)
  const unsigned long long t_begin = RSTAMP();
  v8i kc[8];
  for (int i = 0; i < 8; ++i) for (int e = 0; e < 8; ++e) kc[i][e] = 0x08208208 + lane + i;
  int p = blockIdx.x;
  for (int ti = 0; ti < my_tiles; ++ti, p += gstride) {
    int bm, bn;
    tile_of(p, bm, bn);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[i][j][e] = 0.f;
    for (int kb = 0; kb < KB; ++kb) {
      v8i fa[4], b0[4], b1[4];
      if (RING_DIAG & 4) { for (int t = 0; t < 4; ++t) for (int e = 0; e < 8; ++e) { fa[t][e] = lane + e; b0[t][e] = lane * 3 + e; b1[t][e] = lane * 5 + e; } }
      if (RING_AREG) {                                             // A fragments: straight from global memory (operand-tile order: 16 B + 8 B per lane)
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const int rec = 4 * wm + t;
          const unsigned char* pb = g.A6 + ((int64_t)(bm / 32 + (rec >> 1)) * KB + kb) * F6_PAIR;
          const uint4 a = *reinterpret_cast<const uint4*>(pb + (rec & 1) * 1024 + lane * 16);
          const uint2 b = *reinterpret_cast<const uint2*>(pb + 2048 + (rec & 1) * 512 + lane * 8);
          fa[t][0] = (int)a.x; fa[t][1] = (int)a.y; fa[t][2] = (int)a.z; fa[t][3] = (int)a.w; fa[t][4] = (int)b.x; fa[t][5] = (int)b.y; fa[t][6] = 0; fa[t][7] = 0;
        }
      } else {       // my A item (the other half is released implicitly by the progress word of the first plane)
        const int slot = (gi + my_half) % RING_NS;
        const unsigned long long s0 = RSTAMP();
        wait_for(ready, gi + my_half + 1, fullv, 4);
        st[0] += RSTAMP() - s0;
#pragma unroll
        for (int t = 0; t < 4; ++t) if (!(RING_DIAG & 4)) fa[t] = frag((unsigned)slot * RING_ITEM, 4 * (wm & 1) + t);
      }
      auto load_plane = [&](v8i (&fb)[4], int pl) {               // wait for the plane, read it; publish the previous item once ITS reads are in registers
        const int item = gi + (RING_AREG ? 0 : 2) + pl, slot = item % RING_NS;
        const unsigned long long s0 = RSTAMP();
        wait_for(ready, item + 1, fullv, 4);
        const unsigned long long s1 = RSTAMP();
        st[0] += s1 - s0;
#pragma unroll
        for (int t = 0; t < 4; ++t) if (!(RING_DIAG & 4)) fb[t] = frag((unsigned)slot * RING_ITEM, 4 * wn + t);
        asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");        // everything older than this plane's eight reads has landed
        publish(freev + 4u * (unsigned)w, item);                   // items 0 .. item - 1 are released
        if (RING_DIAG & 256) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); st[1] += RSTAMP() - s1; }
      };
      auto mfma_plane = [&](const v8i (&fb)[4], int pl) {
        const int sb = 127 + 3 + 5 * pl;
        if (RING_DIAG & 2) return;
        const unsigned long long m0 = RSTAMP();
        if (RING_DIAG & 128) {                                     // probe: the MFMAs on CONSTANT registers, the loaded fragments only folded into a VALU sum
          int x = 0;
#pragma unroll
          for (int t = 0; t < 4; ++t) x ^= fb[t][0] ^ fb[t][5] ^ fa[t][0];
          dummy ^= x;
#pragma unroll
          for (int tm = 0; tm < 4; ++tm)
#pragma unroll
            for (int tn = 0; tn < 4; ++tn)
              acc[tm][tn] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(kc[tm], kc[4 + tn], acc[tm][tn], 2, 2, 0, 127, 0, sb);
          __builtin_amdgcn_sched_barrier(0);
          return;
        }
#pragma unroll
        for (int tm = 0; tm < 4; ++tm)
#pragma unroll
          for (int tn = 0; tn < 4; ++tn)
            acc[tm][tn] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(fa[tm], fb[tn], acc[tm][tn], 2, 2, 0, 127, 0, sb);
        __builtin_amdgcn_sched_barrier(0);
        if (RING_DIAG & 256) st[2] += RSTAMP() - m0;
      };
#if RING_ONESET          // one fragment set: 24 registers fewer; the SIMD's other compute wave covers the read latency
      for (int pl = 0; pl < 5; ++pl) { load_plane(b0, pl); mfma_plane(b0, pl); }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      publish(freev + 4u * (unsigned)w, gi + IPK);
#else
      load_plane(b0, 0);
      load_plane(b1, 1); mfma_plane(b0, 0);
      load_plane(b0, 2); mfma_plane(b1, 1);
      load_plane(b1, 3); mfma_plane(b0, 2);
      load_plane(b0, 4); mfma_plane(b1, 3);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      publish(freev + 4u * (unsigned)w, gi + IPK);
      mfma_plane(b0, 4);
#endif
      gi += IPK;
    }
    // epilogue (the production one) through this wave's own slice
    const unsigned long long e0 = RSTAMP();
    float4 ep_rs[2], ep_bv[2];
#pragma unroll
    for (int tn = 0; tn < 2; ++tn) {
      const int n = bn + wn * 64 + tn * 32 + (lane & 7) * 4;
      ep_rs[tn] = make_float4(0.f, 0.f, 0.f, 0.f); ep_bv[tn] = ep_rs[tn];
      if (n < g.N) { ep_rs[tn] = *reinterpret_cast<const float4*>(g.rowscale + n); if (g.bias) ep_bv[tn] = *reinterpret_cast<const float4*>(g.bias + n); }
    }
    char* eb = smem + RING_EPI_OFF + w * EPI_WAVE;
    const int c4 = (lane & 7) * 4;
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
        for (int it2 = 0; it2 < 2; ++it2) {
          const int r16 = it2 * 8 + (lane >> 3);
          const float4 v = *reinterpret_cast<const float4*>(eb + r16 * 144 + c4 * 4);
          const int m = bm + wm * 64 + tm * 16 + r16;
          float4 o;
          o.x = v.x * rs.x + bv.x; o.y = v.y * rs.y + bv.y; o.z = v.z * rs.z + bv.z; o.w = v.w * rs.w + bv.w;
          if (n < g.N && m < g.M && (!(RING_DIAG & 16) || o.x == 12345.f)) *reinterpret_cast<float4*>(g.y + (int64_t)m * g.N + n) = o;
        }
      }
    }
    st[3] += RSTAMP() - e0;
  }
  if ((RING_DIAG & 256) && lane == 0 && ga.stamps) {
    st[4] = RSTAMP() - t_begin;
    for (int i = 0; i < 5; ++i) ga.stamps[((size_t)blockIdx.x * 8 + w) * 8 + i] = st[i];
  }
  if ((timed_out || dummy == 0x7fffffff) && lane == 0) atomicAdd(ga.err, 1);
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

#ifdef RING_ONLY
int main() {      // timing of the ring kernel alone on random operand bytes (every 6-bit code is a number)
  const int M = 8192, N = 3072, K = 768, KB = K / 128;
  const size_t a_bytes = (size_t)(M / 32) * KB * F6_PAIR, w_bytes = (size_t)FP6_NL * (N / 32) * KB * F6_PAIR;
  unsigned char *dA, *dW; float *rs, *bias, *y; int* derr;
  hipMalloc(&dA, a_bytes); hipMalloc(&dW, w_bytes); hipMalloc(&rs, N * 4); hipMalloc(&bias, N * 4); hipMalloc(&y, (size_t)M * N * 4); hipMalloc(&derr, 4);
  { std::vector<unsigned char> h(std::max(a_bytes, w_bytes)); unsigned x = 12345; for (auto& v : h) { x = x * 1664525u + 1013904223u; v = (unsigned char)(x >> 24); }
    hipMemcpy(dA, h.data(), a_bytes, hipMemcpyHostToDevice); hipMemcpy(dW, h.data(), w_bytes, hipMemcpyHostToDevice); }
  hipMemset(rs, 0, N * 4); hipMemset(bias, 0, N * 4); hipMemset(derr, 0, 4);
  if (getenv("FP6_ZERO")) { hipMemset(dA, 0, a_bytes); hipMemset(dW, 0, w_bytes); printf("(all-zero operands)\n"); }      // how much of the time is the data-dependent clock?
  GemmFp6RingArgs fr{}; fr.f.A6 = dA; fr.f.W6 = dW; fr.f.rowscale = rs; fr.f.bias = bias; fr.f.y = y; fr.f.M = M; fr.f.N = N; fr.f.K = K;
  fr.f.tiles_m = M / 128; fr.f.tiles_n = N / 128; fr.err = derr;
  unsigned long long* dst = nullptr; hipMalloc(&dst, 256 * 8 * 8 * 8); hipMemset(dst, 0, 256 * 8 * 8 * 8); fr.stamps = dst;
  hipFuncSetAttribute((const void*)gemm_fp6_ring_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, RING_LDS);
  const unsigned gridr = std::min<unsigned>((M / 256) * (N / 128), gemm_grid(1 << 30));
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int rep = 0; rep < 3; ++rep) {
    float ms;
    for (int i = 0; i < 20; ++i) gemm_fp6_ring_kernel<<<gridr, 768, RING_LDS>>>(fr);
    hipEventRecord(a); for (int i = 0; i < 200; ++i) gemm_fp6_ring_kernel<<<gridr, 768, RING_LDS>>>(fr); hipEventRecord(b); hipEventSynchronize(b); hipEventElapsedTime(&ms, a, b);
    int herr = 0; hipMemcpy(&herr, derr, 4, hipMemcpyDeviceToHost);
    printf("ring NS=%d LOOK=%d DIAG=%d: %.1f us (%d time-outs)\n", RING_NS, RING_LOOK, RING_DIAG, ms * 5.f, herr);
  }
  {   // the double-buffered 256 x 128 form
    hipFuncSetAttribute((const void*)gemm_fp6_db_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, DB_LDS);
    const unsigned gd = std::min<unsigned>((M / 256) * (N / 128), gemm_grid(1 << 30));
    for (int rep = 0; rep < 3; ++rep) {
      float ms;
      for (int i = 0; i < 20; ++i) gemm_fp6_db_kernel<<<gd, 512, DB_LDS>>>(fr.f);
      hipEventRecord(a); for (int i = 0; i < 200; ++i) gemm_fp6_db_kernel<<<gd, 512, DB_LDS>>>(fr.f); hipEventRecord(b); hipEventSynchronize(b); hipEventElapsedTime(&ms, a, b);
      printf("double-buffered 256 x 128 form (8 waves, A fragments from global memory): %.1f us\n", ms * 5.f);
    }
  }
  {   // double-buffered half stages, hand-allocated plane blocks, two workgroups per CU
    hipFuncSetAttribute((const void*)gemm_fp6_hs2_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, F6_STAGE);
    const unsigned g2 = std::min<unsigned>((M / 128) * (N / 128), 2 * gemm_grid(1 << 30));
    for (int rep = 0; rep < 3; ++rep) {
      float ms;
      for (int i = 0; i < 20; ++i) gemm_fp6_hs2_kernel<<<g2, 256, F6_STAGE>>>(fr.f);
      hipEventRecord(a); for (int i = 0; i < 200; ++i) gemm_fp6_hs2_kernel<<<g2, 256, F6_STAGE>>>(fr.f); hipEventRecord(b); hipEventSynchronize(b); hipEventElapsedTime(&ms, a, b);
      printf("double-buffered half stages with hand-allocated plane blocks (two workgroups per CU): %.1f us\n", ms * 5.f);
    }
  }
  {   // the production recipe on FP6 operands: half stages through one 36-KB buffer, three workgroups per CU
    hipFuncSetAttribute((const void*)gemm_fp6_h3_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, HS_HALF);
    const unsigned g3 = std::min<unsigned>((M / 128) * (N / 128), 3 * gemm_grid(1 << 30));
    for (int rep = 0; rep < 3; ++rep) {
      float ms;
      for (int i = 0; i < 20; ++i) gemm_fp6_h3_kernel<<<g3, 256, HS_HALF>>>(fr.f);
      hipEventRecord(a); for (int i = 0; i < 200; ++i) gemm_fp6_h3_kernel<<<g3, 256, HS_HALF>>>(fr.f); hipEventRecord(b); hipEventSynchronize(b); hipEventElapsedTime(&ms, a, b);
      printf("half stages through one 36-KB buffer, three workgroups per CU (the production recipe): %.1f us\n", ms * 5.f);
    }
  }
  {   // the half-stage double-buffered transplant
    hipFuncSetAttribute((const void*)gemm_fp6_hs_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, F6_STAGE);
    const unsigned gh = std::min<unsigned>((M / 128) * (N / 128), 2 * gemm_grid(1 << 30));
    for (int rep = 0; rep < 3; ++rep) {
      float ms;
      for (int i = 0; i < 20; ++i) gemm_fp6_hs_kernel<<<gh, 256, F6_STAGE>>>(fr.f);
      hipEventRecord(a); for (int i = 0; i < 200; ++i) gemm_fp6_hs_kernel<<<gh, 256, F6_STAGE>>>(fr.f); hipEventRecord(b); hipEventSynchronize(b); hipEventElapsedTime(&ms, a, b);
      printf("half-stage double-buffered transplant (128 x 128 tiles, two workgroups per CU): %.1f us\n", ms * 5.f);
    }
  }
  {   // the 128 x 64 transplant, three workgroups per CU
    hipFuncSetAttribute((const void*)gemm_fp6_n64_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, N64_STAGE);
    const unsigned g6 = std::min<unsigned>((M / 128) * (N / 64), 3 * gemm_grid(1 << 30));
    for (int rep = 0; rep < 3; ++rep) {
      float ms;
      for (int i = 0; i < 20; ++i) gemm_fp6_n64_kernel<<<g6, 256, N64_STAGE>>>(fr.f);
      hipEventRecord(a); for (int i = 0; i < 200; ++i) gemm_fp6_n64_kernel<<<g6, 256, N64_STAGE>>>(fr.f); hipEventRecord(b); hipEventSynchronize(b); hipEventElapsedTime(&ms, a, b);
      printf("128 x 64 transplant (three workgroups per CU): %.1f us\n", ms * 5.f);
    }
  }
  {   // the eight-wave transplant on the same operands
    hipFuncSetAttribute((const void*)gemm_fp6_w8_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, F6_STAGE);
    const unsigned g8 = std::min<unsigned>((M / 128) * (N / 128), 2 * gemm_grid(1 << 30));
    for (int rep = 0; rep < 3; ++rep) {
      float ms;
      for (int i = 0; i < 20; ++i) gemm_fp6_w8_kernel<<<g8, 512, F6_STAGE>>>(fr.f);
      hipEventRecord(a); for (int i = 0; i < 200; ++i) gemm_fp6_w8_kernel<<<g8, 512, F6_STAGE>>>(fr.f); hipEventRecord(b); hipEventSynchronize(b); hipEventElapsedTime(&ms, a, b);
      printf("eight-wave transplant (128 x 128 tiles, two workgroups per CU): %.1f us\n", ms * 5.f);
    }
  }
  if (RING_DIAG & 256) {
    std::vector<unsigned long long> h(256 * 8 * 8); hipMemcpy(h.data(), dst, h.size() * 8, hipMemcpyDeviceToHost);
    double sum[5] = {0, 0, 0, 0, 0};
    for (unsigned b = 0; b < gridr; ++b) for (int w = 0; w < 8; ++w) for (int i = 0; i < 5; ++i) sum[i] += (double)h[((size_t)b * 8 + w) * 8 + i];
    const double nw = gridr * 8.0;
    printf("compute waves, counter cycles per wave (18 k blocks, 3 tiles): waiting for items %.0f | fragment reads landing %.0f | MFMA issue %.0f | epilogue %.0f | total %.0f\n",
           sum[0] / nw, sum[1] / nw, sum[2] / nw, sum[3] / nw, sum[4] / nw);
  }
  return hipGetLastError() == hipSuccess ? 0 : 1;
}
#else
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
  // ---- LoRA-up operands, as the production activation pass / preparation make them: t 2^g[m] = thi + tlo, B' 2^e16[n] = Bhi + Blo
  const int R = 64;
  std::vector<float> tt((size_t)M * R), Bp((size_t)N * R), rowinv(M);
  std::vector<_Float16> thi((size_t)M * R), tlo((size_t)M * R), bhi((size_t)N * R), blo((size_t)N * R);
  { std::normal_distribution<float> n1(0.f, 1.f);
    for (auto& v : tt) v = n1(rng);
    for (auto& v : Bp) v = n1(rng) * 0.01f; }
  for (int m = 0; m < M; ++m) {
    float mx = 0; for (int j = 0; j < R; ++j) mx = fmaxf(mx, fabsf(tt[(size_t)m * R + j]));
    int ex; frexpf(mx, &ex); const float pg = ldexpf(1.f, 14 - ex); rowinv[m] = 1.f / pg;
    for (int j = 0; j < R; ++j) { const float v = tt[(size_t)m * R + j] * pg; const _Float16 h = (_Float16)v; thi[(size_t)m * R + j] = h; tlo[(size_t)m * R + j] = (_Float16)(v - (float)h); }
  }
  for (int n = 0; n < N; ++n)
    for (int j = 0; j < R; ++j) { const float v = Bp[(size_t)n * R + j] / rs16[n]; const _Float16 h = (_Float16)v; bhi[(size_t)n * R + j] = h; blo[(size_t)n * R + j] = (_Float16)(v - (float)h); }
  auto up = [&](const void* h, size_t bytes) { void* d; hipMalloc(&d, bytes); hipMemcpy(d, h, bytes, hipMemcpyHostToDevice); return d; };
  float *bias, *y16, *y6;
  hipMalloc(&bias, N * 4); hipMemset(bias, 0, N * 4);
  hipMalloc(&y16, (size_t)M * N * 4); hipMalloc(&y6, (size_t)M * N * 4);
  hipMemset(y16, 0, (size_t)M * N * 4); hipMemset(y6, 0, (size_t)M * N * 4);
  GemmF16Args g{};
  g.qx = (const _Float16*)up(qx.data(), qx.size() * 2); g.Whi = (const _Float16*)up(whi.data(), whi.size() * 2); g.Wlo = (const _Float16*)up(wlo.data(), wlo.size() * 2);
  g.thi = (const _Float16*)up(thi.data(), thi.size() * 2); g.tlo = (const _Float16*)up(tlo.data(), tlo.size() * 2);
  g.Bhi = (const _Float16*)up(bhi.data(), bhi.size() * 2); g.Blo = (const _Float16*)up(blo.data(), blo.size() * 2); g.rowinv = (const float*)up(rowinv.data(), M * 4);
  g.rowscale = (const float*)up(rs16.data(), N * 4); g.bias = bias; g.y = y16; g.M = M; g.N = N; g.Kp = K; g.Rp = R;
  g.tiles_m = M / GM; g.tiles_n = N / GN; g.a_limbs = 1; g.split = 1;
  GemmFp6Args f{};
  { auto pa = pack6(qc, M, K); f.A6 = (const unsigned char*)up(pa.data(), pa.size()); }
  { std::vector<unsigned char> all; for (int i = 0; i < FP6_NL; ++i) { auto pw = pack6(dc[i], N, K); all.insert(all.end(), pw.begin(), pw.end()); }
    f.W6 = (const unsigned char*)up(all.data(), all.size()); }
  f.rowscale = (const float*)up(rs6.data(), N * 4); f.bias = bias; f.y = y6; f.M = M; f.N = N; f.K = K; f.tiles_m = M / 128; f.tiles_n = N / 128;
  f.thi = g.thi; f.tlo = g.tlo; f.Bhi = g.Bhi; f.Blo = g.Blo; f.rowinv = g.rowinv; f.Rp = 0;
  float* y6L; hipMalloc(&y6L, (size_t)M * N * 4); hipMemset(y6L, 0, (size_t)M * N * 4);
  GemmFp6Args fL = f; fL.Rp = R; fL.y = y6L;       // the same kernel WITH the LoRA-up stages: the like-for-like partner of the production kernel
  auto k16 = gemm_f16x2_t128_kernel<1, 0>;
  hipFuncSetAttribute((const void*)k16, hipFuncAttributeMaxDynamicSharedMemorySize, T128_LDS);
  hipFuncSetAttribute((const void*)gemm_fp6_t128_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, F6_STAGE);
  const unsigned cus = gemm_grid(1 << 30);
  const unsigned grid16 = std::min<unsigned>(2 * g.tiles_m * g.tiles_n, T128_WGS * cus), grid6 = std::min<unsigned>(f.tiles_m * f.tiles_n, FP6_WGS * cus);
  k16<<<grid16, 256, T128_LDS>>>(g);
  gemm_fp6_t128_kernel<<<grid6, 256, F6_STAGE>>>(f);
  gemm_fp6_t128_kernel<<<grid6, 256, F6_STAGE>>>(fL);
  if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed: %s\n", hipGetErrorString(hipGetLastError())); return 1; }
  // ---- accuracy: both against a double sum on sampled rows, and against each other everywhere
  std::vector<float> h16((size_t)M * N), h6((size_t)M * N), h6L((size_t)M * N);
  hipMemcpy(h16.data(), y16, h16.size() * 4, hipMemcpyDeviceToHost); hipMemcpy(h6.data(), y6, h6.size() * 4, hipMemcpyDeviceToHost); hipMemcpy(h6L.data(), y6L, h6L.size() * 4, hipMemcpyDeviceToHost);
  double rms = 0; for (size_t i = 0; i < h16.size(); i += 97) rms += (double)h16[i] * h16[i]; rms = sqrt(rms / (h16.size() / 97));
  double e16 = 0, e6 = 0, e66 = 0;
  for (int m = 0; m < M; m += 257)
    for (int n = 0; n < N; ++n) {
      double s = 0; for (int k = 0; k < K; ++k) s += (double)q[(size_t)m * K + k] * (double)Wp[(size_t)n * K + k];
      for (int j = 0; j < R; ++j) s += (double)tt[(size_t)m * R + j] * (double)Bp[(size_t)n * R + j];
      const double bound = 1e-5 * fabs(s) + 1e-5 * rms;
      e16 = fmax(e16, fabs(h16[(size_t)m * N + n] - s) / bound); e6 = fmax(e6, fabs(h6L[(size_t)m * N + n] - s) / bound);
    }
  for (size_t i = 0; i < h16.size(); ++i) e66 = fmax(e66, fabs((double)h16[i] - h6L[i]) / (1e-5 * fabs((double)h16[i]) + 1e-5 * rms));
  printf("WITH the LoRA-up stages in both kernels -- max err / (1e-5 |y| + 1e-5 rms): f16 limbs vs double %.3f, FP6 digits vs double %.3f (sampled rows); FP6 vs f16 limbs, every output %.3f\n", e16, e6, e66);
  // ---- the ring form
  float* y6r; hipMalloc(&y6r, (size_t)M * N * 4); hipMemset(y6r, 0, (size_t)M * N * 4);
  int* derr; hipMalloc(&derr, 4); hipMemset(derr, 0, 4);
  GemmFp6RingArgs fr{f, derr, nullptr}; fr.f.y = y6r;
  hipFuncSetAttribute((const void*)gemm_fp6_ring_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, RING_LDS);
  const unsigned gridr = std::min<unsigned>((M / 256) * (N / 128), cus);
  gemm_fp6_ring_kernel<<<gridr, 768, RING_LDS>>>(fr);
  if (hipDeviceSynchronize() != hipSuccess) { printf("ring kernel failed: %s\n", hipGetErrorString(hipGetLastError())); return 1; }
  { int herr = 0; hipMemcpy(&herr, derr, 4, hipMemcpyDeviceToHost);
    std::vector<float> hr((size_t)M * N); hipMemcpy(hr.data(), y6r, hr.size() * 4, hipMemcpyDeviceToHost);
    size_t diff = 0; for (size_t i = 0; i < hr.size(); ++i) diff += hr[i] != h6[i];
    printf("ring kernel: %d waves timed out; %zu of %zu outputs differ from the transplant kernel's (same products, same order per output)\n", herr, diff, hr.size()); }
  {   // the double-buffered 256 x 128 form: same products per output in the same order -> bit-identical
    float* yd; hipMalloc(&yd, (size_t)M * N * 4); hipMemset(yd, 0, (size_t)M * N * 4);
    GemmFp6Args fd = f; fd.y = yd;
    hipFuncSetAttribute((const void*)gemm_fp6_db_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, DB_LDS);
    const unsigned gd = std::min<unsigned>((M / 256) * (N / 128), cus);
    gemm_fp6_db_kernel<<<gd, 512, DB_LDS>>>(fd);
    if (hipDeviceSynchronize() != hipSuccess) { printf("db kernel failed\n"); return 1; }
    std::vector<float> hd((size_t)M * N); hipMemcpy(hd.data(), yd, hd.size() * 4, hipMemcpyDeviceToHost);
    size_t diff = 0; for (size_t i = 0; i < hd.size(); ++i) diff += hd[i] != h6[i];
    printf("double-buffered 256 x 128 form: %zu of %zu outputs differ from the transplant's\n", diff, hd.size());
    hipEvent_t a2, b2; hipEventCreate(&a2); hipEventCreate(&b2); float ms;
    for (int i = 0; i < 10; ++i) gemm_fp6_db_kernel<<<gd, 512, DB_LDS>>>(fd);
    hipEventRecord(a2); for (int i = 0; i < 100; ++i) gemm_fp6_db_kernel<<<gd, 512, DB_LDS>>>(fd); hipEventRecord(b2); hipEventSynchronize(b2); hipEventElapsedTime(&ms, a2, b2);
    printf("double-buffered 256 x 128 form: %.1f us\n", ms * 10.f);
  }
  {   // double-buffered half stages with hand-allocated plane blocks -> bit-identical
    float* y2; hipMalloc(&y2, (size_t)M * N * 4); hipMemset(y2, 0, (size_t)M * N * 4);
    GemmFp6Args f2 = f; f2.y = y2;
    hipFuncSetAttribute((const void*)gemm_fp6_hs2_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, F6_STAGE);
    gemm_fp6_hs2_kernel<<<grid6, 256, F6_STAGE>>>(f2);
    if (hipDeviceSynchronize() != hipSuccess) { printf("hs2 kernel failed\n"); return 1; }
    std::vector<float> h2v((size_t)M * N); hipMemcpy(h2v.data(), y2, h2v.size() * 4, hipMemcpyDeviceToHost);
    size_t diff = 0; for (size_t i = 0; i < h2v.size(); ++i) diff += h2v[i] != h6[i];
    printf("double-buffered half stages with hand-allocated plane blocks: %zu of %zu outputs differ from the transplant's\n", diff, h2v.size());
  }
  {   // the production recipe on FP6 operands -> bit-identical
    float* y3; hipMalloc(&y3, (size_t)M * N * 4); hipMemset(y3, 0, (size_t)M * N * 4);
    GemmFp6Args f3 = f; f3.y = y3;
    hipFuncSetAttribute((const void*)gemm_fp6_h3_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, HS_HALF);
    const unsigned g3 = std::min<unsigned>((M / 128) * (N / 128), 3 * cus);
    gemm_fp6_h3_kernel<<<g3, 256, HS_HALF>>>(f3);
    if (hipDeviceSynchronize() != hipSuccess) { printf("h3 kernel failed\n"); return 1; }
    std::vector<float> h3v((size_t)M * N); hipMemcpy(h3v.data(), y3, h3v.size() * 4, hipMemcpyDeviceToHost);
    size_t diff = 0; for (size_t i = 0; i < h3v.size(); ++i) diff += h3v[i] != h6[i];
    printf("half stages through one buffer, three workgroups per CU: %zu of %zu outputs differ from the transplant's\n", diff, h3v.size());
  }
  {   // the half-stage double-buffered transplant: same products per output in the same order -> bit-identical
    float* yh; hipMalloc(&yh, (size_t)M * N * 4); hipMemset(yh, 0, (size_t)M * N * 4);
    GemmFp6Args fh = f; fh.y = yh;
    hipFuncSetAttribute((const void*)gemm_fp6_hs_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, F6_STAGE);
    gemm_fp6_hs_kernel<<<grid6, 256, F6_STAGE>>>(fh);
    if (hipDeviceSynchronize() != hipSuccess) { printf("hs kernel failed\n"); return 1; }
    std::vector<float> hh((size_t)M * N); hipMemcpy(hh.data(), yh, hh.size() * 4, hipMemcpyDeviceToHost);
    size_t diff = 0; for (size_t i = 0; i < hh.size(); ++i) diff += hh[i] != h6[i];
    printf("half-stage double-buffered transplant: %zu of %zu outputs differ from the transplant's\n", diff, hh.size());
    hipEvent_t a2, b2; hipEventCreate(&a2); hipEventCreate(&b2); float ms;
    for (int i = 0; i < 10; ++i) gemm_fp6_hs_kernel<<<grid6, 256, F6_STAGE>>>(fh);
    hipEventRecord(a2); for (int i = 0; i < 100; ++i) gemm_fp6_hs_kernel<<<grid6, 256, F6_STAGE>>>(fh); hipEventRecord(b2); hipEventSynchronize(b2); hipEventElapsedTime(&ms, a2, b2);
    printf("half-stage double-buffered transplant: %.1f us\n", ms * 10.f);
  }
  {   // the 128 x 64 transplant: same products per output in the same order -> bit-identical to the 128 x 128 transplant
    float* y64; hipMalloc(&y64, (size_t)M * N * 4); hipMemset(y64, 0, (size_t)M * N * 4);
    GemmFp6Args f64 = f; f64.y = y64;
    hipFuncSetAttribute((const void*)gemm_fp6_n64_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, N64_STAGE);
    const unsigned g64 = std::min<unsigned>((M / 128) * (N / 64), 3 * cus);
    gemm_fp6_n64_kernel<<<g64, 256, N64_STAGE>>>(f64);
    if (hipDeviceSynchronize() != hipSuccess) { printf("n64 kernel failed\n"); return 1; }
    std::vector<float> hn((size_t)M * N); hipMemcpy(hn.data(), y64, hn.size() * 4, hipMemcpyDeviceToHost);
    size_t diff = 0; for (size_t i = 0; i < hn.size(); ++i) diff += hn[i] != h6[i];
    printf("128 x 64 transplant: %zu of %zu outputs differ from the 128 x 128 transplant's\n", diff, hn.size());
    hipEvent_t a2, b2; hipEventCreate(&a2); hipEventCreate(&b2); float ms;
    for (int i = 0; i < 10; ++i) gemm_fp6_n64_kernel<<<g64, 256, N64_STAGE>>>(f64);
    hipEventRecord(a2); for (int i = 0; i < 100; ++i) gemm_fp6_n64_kernel<<<g64, 256, N64_STAGE>>>(f64); hipEventRecord(b2); hipEventSynchronize(b2); hipEventElapsedTime(&ms, a2, b2);
    printf("128 x 64 transplant (three workgroups per CU): %.1f us\n", ms * 10.f);
  }
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int rep = 0; rep < 3; ++rep) {
    float ms16, ms6;
    for (int i = 0; i < 10; ++i) k16<<<grid16, 256, T128_LDS>>>(g);
    hipEventRecord(a); for (int i = 0; i < 100; ++i) k16<<<grid16, 256, T128_LDS>>>(g); hipEventRecord(b); hipEventSynchronize(b); hipEventElapsedTime(&ms16, a, b);
    for (int i = 0; i < 10; ++i) gemm_fp6_t128_kernel<<<grid6, 256, F6_STAGE>>>(fL);
    hipEventRecord(a); for (int i = 0; i < 100; ++i) gemm_fp6_t128_kernel<<<grid6, 256, F6_STAGE>>>(fL); hipEventRecord(b); hipEventSynchronize(b); hipEventElapsedTime(&ms6, a, b);
    float msr;
    for (int i = 0; i < 10; ++i) gemm_fp6_ring_kernel<<<gridr, 768, RING_LDS>>>(fr);
    hipEventRecord(a); for (int i = 0; i < 100; ++i) gemm_fp6_ring_kernel<<<gridr, 768, RING_LDS>>>(fr); hipEventRecord(b); hipEventSynchronize(b); hipEventElapsedTime(&msr, a, b);
    printf("FP6 ring kernel (256 x 128 tiles, 8 compute + 4 loader waves, %d slots, look-ahead %d): %.1f us\n", RING_NS, RING_LOOK, msr * 10.f);
    printf("whole contraction 8192 x 768 x 3072 + LoRA-up r = 64 (the headline): production kernel, f16 limbs (2 MFMA 16x16x32 per 32 k) %.1f us | FP6 digits (%d MFMA 16x16x128 per 128 k, %d workgroups per CU) + the same f16 LoRA stages %.1f us\n",
           ms16 * 10.f, FP6_NL, FP6_WGS, ms6 * 10.f);
  }
  return hipGetLastError() == hipSuccess ? 0 : 1;
}
#endif
