// Keeps `wgs` workgroups of 512 threads resident for ~`cycles` clock cycles each (one per CU in practice): a stand-in
// for a communication kernel that occupies a few CUs while a compute kernel runs (tools/queue_bench.py).
#include <hip/hip_runtime.h>
__global__ __launch_bounds__(512) void spin_kernel(long long cycles, int* sink) {
  const long long t0 = clock64();
  int v = 0;
  while (clock64() - t0 < cycles) v += 1;
  if (v == -1) *sink = v;
}
extern "C" void spin_launch(int wgs, long long cycles, void* stream) {
  static int* sink = nullptr;
  if (!sink) hipMalloc(&sink, 4);
  hipLaunchKernelGGL(spin_kernel, dim3(wgs), dim3(512), 0, (hipStream_t)stream, cycles, sink);
}
