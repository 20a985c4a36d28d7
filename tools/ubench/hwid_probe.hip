// Which hardware slots do the workgroups of a two-per-CU launch get?  Prints, per CU, the blocks it ran with their HW_ID
// fields and start times (development probe for the stagger experiment of sosfilt_clip.hip).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <map>
#include <vector>
__global__ __launch_bounds__(256, 2) void probe(unsigned* out, int spin) {
  extern __shared__ float lds[];
  unsigned hw, xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  float acc = 0.f;
  for (int i = 0; i < spin; ++i) { lds[threadIdx.x] = acc; acc += lds[(threadIdx.x + 1) & 255] * 1.0001f + 1.f; }
  unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0) {
    out[blockIdx.x * 6 + 0] = hw; out[blockIdx.x * 6 + 1] = xcc;
    out[blockIdx.x * 6 + 2] = (unsigned)t0; out[blockIdx.x * 6 + 3] = (unsigned)(t0 >> 32);
    out[blockIdx.x * 6 + 4] = (unsigned)(t1 - t0); out[blockIdx.x * 6 + 5] = __float_as_uint(acc);
  }
}
int main() {
  const int nb = 1024;
  unsigned* d; hipMalloc(&d, nb * 6 * 4);
  hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 74 * 1024);
  hipLaunchKernelGGL(probe, dim3(nb), dim3(256), 74 * 1024, 0, d, 20000);
  hipDeviceSynchronize();
  std::vector<unsigned> h(nb * 6);
  hipMemcpy(h.data(), d, nb * 6 * 4, hipMemcpyDeviceToHost);
  unsigned long long tmin = ~0ull;
  for (int b = 0; b < nb; ++b) { unsigned long long t = ((unsigned long long)h[b * 6 + 3] << 32) | h[b * 6 + 2]; if (t < tmin) tmin = t; }
  std::map<unsigned, std::vector<int>> cu;
  for (int b = 0; b < nb; ++b) {
    const unsigned hw = h[b * 6], xcc = h[b * 6 + 1] & 0xf;
    const unsigned cu_id = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
    cu[(xcc << 12) | (se << 8) | (sh << 4) | cu_id].push_back(b);
  }
  printf("distinct CUs seen: %zu\n", cu.size());
  int shown = 0;
  for (auto& kv : cu) {
    if (shown++ >= 12) break;
    printf("xcc %u se %u sh %u cu %2u :", kv.first >> 12, (kv.first >> 8) & 7, (kv.first >> 4) & 1, kv.first & 15);
    for (int b : kv.second) {
      const unsigned hw = h[b * 6];
      unsigned long long t = ((unsigned long long)h[b * 6 + 3] << 32) | h[b * 6 + 2];
      printf("  [blk %4d wave %u simd %u tg %2u start %6.1f us dur %5.1f us]", b, hw & 15, (hw >> 4) & 3, (hw >> 16) & 15, (t - tmin) / 100.0,
             h[b * 6 + 4] / 100.0);
    }
    printf("\n");
  }
  // how do the pairs differ?
  int same_tg = 0, pairs = 0, first_two_parity = 0;
  for (auto& kv : cu) {
    if (kv.second.size() < 2) continue;
    // the two blocks that started first on this CU
    std::vector<std::pair<unsigned long long, int>> v;
    for (int b : kv.second) v.push_back({((unsigned long long)h[b * 6 + 3] << 32) | h[b * 6 + 2], b});
    std::sort(v.begin(), v.end());
    const unsigned tg0 = (h[v[0].second * 6] >> 16) & 15, tg1 = (h[v[1].second * 6] >> 16) & 15;
    ++pairs; if (tg0 == tg1) ++same_tg; if ((tg0 & 1) != (tg1 & 1)) ++first_two_parity;
  }
  printf("CUs with >= 2 blocks: %d; first two co-resident blocks with the same TG_ID: %d; with TG_ID of different parity: %d\n", pairs, same_tg,
         first_two_parity);
  return 0;
}
