// Microbenchmark: does the fp32-input MFMA (v_mfma_f32_16x16x4_f32) execute beside fp32 VALU work of ANOTHER wave on
// the same SIMD, or do the two share the SIMD's fp32 lanes?  Workgroups of 512 threads (2 waves per SIMD), one per CU:
//   mode 0: both waves of a SIMD run the VALU loop          mode 1: both run the MFMA loop
//   mode 2: waves 0-3 VALU, waves 4-7 MFMA (one of each per SIMD)
//   mode 3: each wave alternates MFMA and VALU in one stream (8 fma per MFMA)
// Reports wall time per mode for the same per-wave instruction counts.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float v4f __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void valu_block(float (&a)[8], float c, float d) {
#pragma unroll
  for (int r = 0; r < 8; ++r)
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = __builtin_fmaf(a[i], c, d);      // 64 v_fma_f32
}
__device__ __forceinline__ void mfma_block(v4f (&acc)[4], float x, float y) {
#pragma unroll
  for (int r = 0; r < 2; ++r)
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, acc[i], 0, 0, 0);   // 8 MFMA
}

template <int MODE>
__global__ __launch_bounds__(512) void k(float* out, int iters) {
  float a[8];
  v4f acc[4];
  for (int i = 0; i < 8; ++i) a[i] = threadIdx.x * 1e-3f + i;
  for (int i = 0; i < 4; ++i) acc[i] = v4f{0.f, 0.f, 0.f, 0.f};
  const float c = 1.0001f, d = 0.9999f;
  const float x = threadIdx.x * 1e-4f, y = 1.f - x;
  const int w = threadIdx.x >> 6;
  const bool do_valu = MODE == 0 || (MODE == 2 && w < 4);
  const bool do_mfma = MODE == 1 || (MODE == 2 && w >= 4);
  long long t0 = clock64();
  if (MODE == 3) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        acc[r & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, acc[r & 3], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 8; ++i) a[i] = __builtin_fmaf(a[i], c, d);
      }
    }
  } else if (do_valu) {
    for (int it = 0; it < iters; ++it) valu_block(a, c, d);
  } else if (do_mfma) {
    for (int it = 0; it < iters; ++it) mfma_block(acc, x, y);
  }
  long long t1 = clock64();
  float s = 0;
  for (int i = 0; i < 8; ++i) s += a[i];
  for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) out[(1 << 20) + w] = (float)(t1 - t0);
}

template <int MODE>
void run(const char* name) {
  float* d;
  hipMalloc(&d, (1 << 22) + 4096);
  const int iters = 4000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), 0, 0, d, 10);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), 0, 0, d, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  float cyc[8];
  hipMemcpy(cyc, d + (1 << 20), 32, hipMemcpyDeviceToHost);
  printf("%-44s wall %.3f ms   wave cycles: w0 %.0f  w4 %.0f  (per iteration: %.1f / %.1f)\n", name, ms, cyc[0], cyc[4],
         cyc[0] / iters, cyc[4] / iters);
  hipFree(d);
}
int main() {
  // per iteration: VALU wave = 64 v_fma (2 cycles each at full rate = 128 SIMD cycles); MFMA wave = 8 MFMA = 256 SIMD cycles
  run<0>("0: 2 waves/SIMD, both VALU (64 fma/iter)");
  run<1>("1: 2 waves/SIMD, both MFMA (8 mfma/iter)");
  run<2>("2: 1 VALU wave + 1 MFMA wave per SIMD");
  run<3>("3: one stream: 8 x (1 mfma + 8 fma) per iter");
  return 0;
}
