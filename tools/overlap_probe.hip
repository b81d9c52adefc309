// Probe: does a wave's global->LDS traffic overlap with other waves' MFMAs on the same CU?  (tuning aid only)
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/overlap_probe.hip -o tools/overlap_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// MODE bits: 1 = compute waves run MFMAs; 2 = copy waves run.  KIND: 0 glds, 1 global_load->VGPR->ds_write, 2 global_load->VGPR only
template <int MODE, int KIND, int COPY_FIRST = 0, int NCOMP = 8, int PRIO = 0>
__global__ __launch_bounds__(768, 3) void probe(const _Float16* src, float* sink, int stages, unsigned long long* stamps) {
  const unsigned long long t_start = __builtin_amdgcn_s_memrealtime();
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w0 = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int w = COPY_FIRST ? (w0 < 4 ? w0 + 8 : w0 - 4) : w0;   // logical role: 0..7 compute, 8..11 copy
  if (w >= 8) {
    if (!(MODE & 2)) return;
    if (PRIO) __builtin_amdgcn_s_setprio(3);
    const int j = w - 8;
    const char* base = (const char*)src + (size_t)(blockIdx.x & 31) * 65536;   // 64 KB per stage per block; 2 MB x 4 regions: L2 resident
    float4 acc4 = make_float4(0, 0, 0, 0);
    for (int s = 0; s < stages; ++s) {
      if (stamps && blockIdx.x == 0 && lane == 0) stamps[1024 + w0 * 64 + s] = __builtin_amdgcn_s_memrealtime();
      const char* sp = base + (size_t)(s & 3) * 65536 * 32;
      if (KIND == 0) {
#pragma unroll
        for (int i = 0; i < 16; ++i)
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(sp + (j * 16 + i) * 1024 + lane * 16),
                                           (__attribute__((address_space(3))) void*)(smem + (s & 1) * 65536 + (j * 16 + i) * 1024), 16, 0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      } else {
        float4 v[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = *reinterpret_cast<const float4*>(sp + (j * 16 + i) * 1024 + lane * 16);
        if (KIND == 1) {
#pragma unroll
          for (int i = 0; i < 16; ++i) *reinterpret_cast<float4*>(smem + (s & 1) * 65536 + (j * 16 + i) * 1024 + lane * 16) = v[i];
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        } else {
#pragma unroll
          for (int i = 0; i < 16; ++i) { acc4.x += v[i].x; acc4.y += v[i].w; }
        }
      }
    }
    if (acc4.x == 12345.f) sink[tid] = acc4.x + acc4.y;
    return;
  }
  if (!(MODE & 1) || w >= NCOMP) return;
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
  f16x8 a = (f16x8)(_Float16)1.f, b = (f16x8)(_Float16)0.5f;
  asm volatile("" : "+v"(a), "+v"(b));
  for (int s = 0; s < stages; ++s) {
    if (stamps && blockIdx.x == 0 && lane == 0) stamps[1024 + w0 * 64 + s] = __builtin_amdgcn_s_memrealtime();
#pragma unroll
    for (int r = 0; r < 8; ++r)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[i], 0, 0, 0);
  }
  float t = 0; for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) t += acc[i][e];
  if (t == 12345.f) sink[tid] = t;
}

template <int MODE, int KIND, int CF = 0, int NC = 8, int PR = 0> float run(const _Float16* src, float* sink, int stages) {
  hipFuncSetAttribute((const void*)probe<MODE, KIND, CF, NC, PR>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int i = 0; i < 3; ++i) probe<MODE, KIND, CF, NC, PR><<<256, 768, 131072>>>(src, sink, stages, nullptr);
  hipEventRecord(a);
  for (int i = 0; i < 20; ++i) probe<MODE, KIND, CF, NC, PR><<<256, 768, 131072>>>(src, sink, stages, nullptr);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b); return ms / 20 * 1e3f;
}

int main() {
  _Float16* src; float* sink;
  hipMalloc(&src, (size_t)65536 * 256 * 16); hipMemset(src, 0, (size_t)65536 * 256 * 16); hipMalloc(&sink, 4096);
  const int stages = 42;   // 42 stages x (32 MFMA per compute wave | 64 KB per CU)
  printf("mfma alone: 8 waves %.1f | 4 waves (1/SIMD) %.1f us\n", run<1, 0>(src, sink, stages), run<1, 0, 0, 4>(src, sink, stages));
  printf("glds  L2-resident:  copy alone %.1f | both %.1f | both, copy prio3 %.1f | both, copy waves oldest %.1f | both, 4 compute waves %.1f us\n",
         run<2, 0>(src, sink, stages), run<3, 0>(src, sink, stages), run<3, 0, 0, 8, 1>(src, sink, stages), run<3, 0, 1>(src, sink, stages), run<3, 0, 0, 4>(src, sink, stages));
  printf("load->VGPR:         copy alone %.1f | both %.1f | both, copy prio3 %.1f | both, copy waves oldest %.1f | both, 4 compute waves %.1f us\n",
         run<2, 2>(src, sink, stages), run<3, 2>(src, sink, stages), run<3, 2, 0, 8, 1>(src, sink, stages), run<3, 2, 1>(src, sink, stages), run<3, 2, 0, 4>(src, sink, stages));
  unsigned long long* st; hipMalloc(&st, 2048 * 8);
  auto show = [&](const char* name, auto launch) {
    hipMemset(st, 0, 2048 * 8);
    for (int i = 0; i < 3; ++i) launch();
    hipDeviceSynchronize();
    static unsigned long long h[2048]; hipMemcpy(h, st, sizeof(h), hipMemcpyDeviceToHost);
    unsigned long long t0 = ~0ull; for (int w = 0; w < 12; ++w) if (h[1024 + w * 64]) t0 = h[1024 + w * 64] < t0 ? h[1024 + w * 64] : t0;
    printf("%s: time (us) at which each physical wave starts stage 0 / 10 / 20 / 30 / 41\n", name);
    for (int w = 0; w < 12; ++w) { if (!h[1024 + w * 64]) continue; printf("  wave %2d:", w); for (int s : {0, 10, 20, 30, 41}) printf(" %6.1f", (h[1024 + w * 64 + s] - t0) / 100.0); printf("\n"); }
  };
  hipFuncSetAttribute((const void*)probe<1, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
  hipFuncSetAttribute((const void*)probe<3, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
  hipFuncSetAttribute((const void*)probe<3, 0, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
  hipFuncSetAttribute((const void*)probe<3, 0, 0, 8, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
  show("mfma alone", [&] { probe<1, 0><<<256, 768, 131072>>>(src, sink, stages, st); });
  show("both (glds), copy = waves 8-11", [&] { probe<3, 0><<<256, 768, 131072>>>(src, sink, stages, st); });
  show("both (glds), copy = waves 0-3", [&] { probe<3, 0, 1><<<256, 768, 131072>>>(src, sink, stages, st); });
  show("both (glds), copy = waves 8-11 at prio 3", [&] { probe<3, 0, 0, 8, 1><<<256, 768, 131072>>>(src, sink, stages, st); });
  return 0;
}
