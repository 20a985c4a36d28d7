// Microbenchmark: issue rate of v_fma_f32 vs v_pk_fma_f32 vs v_add_f32 on gfx950 at 1/2/4 waves per SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float v2f __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ void k(float* out, int iters) {
  float a[8]; v2f p[8];
  for (int i = 0; i < 8; ++i) { a[i] = threadIdx.x * 1e-3f + i; p[i] = v2f{a[i], a[i] + 1}; }
  const float c = 1.0001f, d = 0.9999f; const v2f pc = {c, d}, pd = {d, c};
  long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 8; ++r) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (MODE == 0) a[i] = __builtin_fmaf(a[i], c, d);
        else if (MODE == 1) p[i] = __builtin_elementwise_fma(p[i], pc, pd);
        else a[i] = a[i] + c;
      }
    }
  }
  long long t1 = clock64();
  float s = 0; for (int i = 0; i < 8; ++i) s += a[i] + p[i].x + p[i].y;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = (float)(t1 - t0);
}
template <int MODE> void run(const char* name, int threads) {
  float* d; hipMalloc(&d, 1 << 24);
  const int iters = 2000; // 64 instr per iter
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 0, 0, d, 10);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 0, 0, d, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  float cyc; hipMemcpy(&cyc, d, 4, hipMemcpyDeviceToHost);
  double instr_per_wave = (double)iters * 64;
  int waves_per_simd = threads / 256;
  printf("%-12s waves/SIMD=%d  wave-cycles/instr=%.2f  (SIMD cycles per instr = %.2f)  wall %.3f ms\n", name,
         waves_per_simd ? waves_per_simd : 1, cyc / instr_per_wave, cyc / instr_per_wave / (waves_per_simd ? waves_per_simd : 1), ms);
  hipFree(d);
}
int main() {
  for (int th : {256, 512, 1024}) {
    run<0>("v_fma_f32", th); run<1>("v_pk_fma_f32", th); run<2>("v_add_f32", th);
  }
  return 0;
}
