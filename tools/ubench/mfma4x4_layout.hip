// Lane maps of v_mfma_f32_4x4x1_16b_f32 (16 blocks of 4x4, K = 1), checked with exact integer data:
// which lane supplies A[i] / B[j] of block b and where D[i][j] lands.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float v4f __attribute__((ext_vector_type(4)));
__global__ void k(float* out) {
  const int l = threadIdx.x;
  // A value encodes (lane) as 1 + l; B value encodes 1000 * (1 + l): D = a * b identifies both source lanes
  const float a = (float)(1 + l), b = (float)(1 + l) * 128.f;
  v4f c = {0.f, 0.f, 0.f, 0.f};
  c = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 0, 0, 0);
  for (int r = 0; r < 4; ++r) out[l * 4 + r] = c[r];
}
int main() {
  float* d; hipMalloc(&d, 64 * 4 * 4);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  float h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  int ok = 1;
  for (int l = 0; l < 64; ++l)
    for (int r = 0; r < 4; ++r) {
      const int v = (int)(h[l * 4 + r] / 128.f + 0.5f);          // = (1 + la) * (1 + lb)
      // hypothesis: D[i = r][j = l & 3] of block l >> 2 = A(lane 4 (l >> 2) + r) * B(lane l)
      const int la = 4 * (l >> 2) + r, lb = l;
      if (v != (1 + la) * (1 + lb)) { ok = 0; if (l < 8) printf("lane %d reg %d: got %d want %d\n", l, r, v, (1 + la) * (1 + lb)); }
    }
  printf("4x4x1 layout hypothesis (A: lane 4b+i, B: lane 4b+j, D[i][j]: lane 4b+j reg i): %s\n", ok ? "CONFIRMED" : "WRONG");
  return 0;
}
