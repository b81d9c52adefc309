// What does an MFMA stream with interleaved LDS fragment reads cost per MFMA on gfx950?  (kernel tuning probe)
//   hipcc -O3 --offload-arch=gfx950 tools/contraction_experiments/mfma_stream_probe.hip -o tools/mfma_probe
// Variants (template V): 0 = 8 MFMAs per pair, no reads; 1 = + 2 asm ds_read_b128 per pair, counted wait (window 3);
// 2 = as 1 with a uniform branch in front of the 4 lo MFMAs; 3 = as 1 but 24 hi then 24 lo per stage (dependency distance 24)
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int V, int BAR>
__global__ __launch_bounds__(768) void probe(float* out, unsigned long long* cyc, int stages, int two) {
  __shared__ __attribute__((aligned(16))) char smem[65536];
  const int lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 65536 / 4; i += blockDim.x) reinterpret_cast<float*>(smem)[i] = 0.001f * (i & 255);
  __syncthreads();
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  const unsigned aa = lds0 + lane * 16, ba = lds0 + 16384 + lane * 16;
  f32x4 acc[4][6];
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 6; ++j) for (int e = 0; e < 4; ++e) acc[i][j][e] = 0.f;
  if (threadIdx.x >= 512) {                                  // a third wave per SIMD that only takes part in the barriers
    if (BAR == 1) for (int s = 0; s < stages; ++s) __builtin_amdgcn_s_barrier();
    if (BAR == 2) for (int s = 0; s < 2 * stages + 1; ++s) __builtin_amdgcn_s_barrier();
    return;
  }
  if (BAR == 2 && threadIdx.x >= 256) __builtin_amdgcn_s_barrier();     // group 1 half a stage behind
  const unsigned long long t0 = __builtin_readcyclecounter();
#define RD(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off) : "memory")
#pragma clang loop unroll(disable)
  for (int s = 0; s < stages; ++s) {
    if (BAR) { __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); }
    f16x8 fa[4], bh[3], bl[3];
    for (int t = 0; t < 4; ++t) RD(fa[t], aa, t * 1024);
    for (int j = 0; j < 2; ++j) { RD(bh[j], ba, j * 1024); RD(bl[j], ba, 12288 + j * 1024); }
    if (V == 3) {
      for (int j = 2; j < 3; ++j) { RD(bh[j], ba, j * 1024); RD(bl[j], ba, 12288 + j * 1024); }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int tn = 0; tn < 6; ++tn)
#pragma unroll
        for (int tm = 0; tm < 4; ++tm) acc[tm][tn] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[tm], bh[tn % 3], acc[tm][tn], 0, 0, 0);
#pragma unroll
      for (int tn = 0; tn < 6; ++tn)
#pragma unroll
        for (int tm = 0; tm < 4; ++tm) acc[tm][tn] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[tm], bl[tn % 3], acc[tm][tn], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      continue;
    }
#pragma unroll
    for (int tn = 0; tn < 6; ++tn) {
      if (BAR == 2 && tn == 3) { __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); }
      if (V >= 1 && tn + 2 < 6) { RD(bh[(tn + 2) % 3], ba, (tn + 2) * 1024); RD(bl[(tn + 2) % 3], ba, 12288 + (tn + 2) * 1024); }
      const int ahead = (tn + 2 < 6 ? tn + 2 : 5) - tn;
      if (V == 0 && tn > 0) { }
      else if (V >= 1 && ahead >= 2) asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
      else if (V >= 1 && ahead == 1) asm volatile("s_waitcnt lgkmcnt(2)" ::: "memory");
      else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int tm = 0; tm < 4; ++tm) acc[tm][tn] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[tm], bh[tn % 3], acc[tm][tn], 0, 0, 0);
      if (V != 2 || two) {
#pragma unroll
        for (int tm = 0; tm < 4; ++tm) acc[tm][tn] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[tm], bl[tn % 3], acc[tm][tn], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  asm volatile("s_nop 0" :: "v"(acc[3][5][0]), "v"(acc[0][0][0]) : "memory");
  const unsigned long long t1 = __builtin_readcyclecounter();
  if (BAR == 2 && threadIdx.x < 256) __builtin_amdgcn_s_barrier();
  float sum = 0.f;
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 6; ++j) for (int e = 0; e < 4; ++e) sum += acc[i][j][e];
  out[blockIdx.x * blockDim.x + threadIdx.x] = sum;
  if (lane == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}
template <int V, int BAR> void run(int threads, const char* what) {
  float* out; unsigned long long* cyc; hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 256 * 12 * 8);
  const int stages = 200;
  for (int rep = 0; rep < 2; ++rep) probe<V, BAR><<<256, threads>>>(out, cyc, stages, 1);
  hipDeviceSynchronize();
  static unsigned long long h[256 * 12]; hipMemcpy(h, cyc, sizeof h, hipMemcpyDeviceToHost);
  double s = 0; const int wpb = threads / 64, cw = wpb > 8 ? 8 : wpb; const int nw = 256 * cw; for (int b = 0; b < 256; ++b) for (int i = 0; i < cw; ++i) s += (double)h[b * wpb + i];
  printf("%-58s compute waves/SIMD %d, barrier %d, threads %d: %.1f cycles per stage of 48 MFMAs per wave = %.1f per MFMA\n", what, (threads > 512 ? 512 : threads) / 256, BAR, threads, s / nw / stages, s / nw / stages / 48);
  hipFree(out); hipFree(cyc);
}
int main() {
  run<1, 0>(512, "V1 free running");
  run<1, 1>(512, "V1 + s_barrier per stage (waves in step)");
  run<1, 2>(512, "V1 + two barriers per stage, groups half a stage apart");
  run<1, 2>(768, "V1 + two barriers, staggered, + a barrier-only third wave");
  run<1, 1>(768, "V1 + s_barrier per stage + a barrier-only third wave");
  return hipGetLastError() == hipSuccess ? 0 : 1;
}
