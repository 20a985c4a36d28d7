// Microbenchmark: issue rate of v_fma_f64 on gfx950 at 1/2/4 waves per SIMD (wall-clock based).
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(double* out, int iters) {
  double a[8];
  for (int i = 0; i < 8; ++i) a[i] = threadIdx.x * 1e-3 + i;
  const double c = 1.0001, d = 0.9999;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 8; ++r) {
#pragma unroll
      for (int i = 0; i < 8; ++i) a[i] = __builtin_fma(a[i], c, d);
    }
  }
  double s = 0; for (int i = 0; i < 8; ++i) s += a[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main() {
  double* d; hipMalloc(&d, 1 << 24);
  for (int th : {256, 512, 1024}) {
    const int iters = 2000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(k, dim3(256), dim3(th), 0, 0, d, iters);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(256), dim3(th), 0, 0, d, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double instr_per_simd = (double)iters * 64 * (th / 256);
    printf("v_fma_f64 waves/SIMD=%d: %.3f ms -> %.2f ns per wave-instruction per SIMD\n", th / 256, ms, ms * 1e6 / instr_per_simd);
  }
  return 0;
}
