// Welch partial periodograms for nperseg = nfft = 4096 (BASELINE config C5: one-hour streams, 50 % overlap), one
// WAVE per segment: scipy.signal.welch as called by compute_psd_welch (sygnals/core/dsp.py:495-560).
//
// A 4096-sample real segment is the 2048-point complex sequence z[m] = x[2m] + i x[2m+1]; its transform is built from
// two 1024-point wave FFTs (wave_fft.h) of the even and odd elements of z -- lane l loads x[4 (64 a + l) .. + 3] as one
// 16-byte word: the first two floats feed the even transform, the last two the odd one -- followed, in registers, by
// the radix-2 combine and the real-input split on mirror pairs: the lane that owns bin k of the 1024-point transforms
// also owns bin 1024 - k, which is everything the bins {k, 1024 - k, 1024 + k, 2048 - k} of the segment need.
// |X|^2 is accumulated per bin in registers over the segments of a wave (segment g, g + n_waves, ...); the per-wave sums
// go to the work buffer and are combined in float64, fixed order, by welch_final_kernel (fft_generic.hip).
// No workgroup barrier inside the segment loop: the waves of a workgroup only share the constant tables.
#include "wave_fft.h"

namespace syg {
namespace {

constexpr int WW = 4;                    // waves per workgroup
constexpr int NSEG = 4096;
constexpr int MBIN = 2048;               // one-sided bins 0 .. 2048

struct WelchLds {
  float2 sc[WW][wfft::SC_COMPLEX];
  float2 tw2l[wfft::TW2_COMPLEX];
  float2 tw1l[wfft::TW1_COMPLEX];
  float2 t2048[8][64];                   // W_2048^k of (lane, unit j, pair d), index j * 4 + d
  float2 t4096[8][64];                   // W_4096^k
  float win[NSEG];
};

// E = zk + conj(zm), O = -i (zk - conj(zm));  X[k] = (E + w O) / 2,  X[N/2 - k] = conj(E - w O) / 2 (powers only)
__device__ __forceinline__ void split_pow(float2 zk, float2 zm, float2 w, float& pk, float& pm) {
  const float2 E = make_float2(zk.x + zm.x, zk.y - zm.y);
  const float2 O = make_float2(zk.y + zm.y, zm.x - zk.x);
  const float2 wO = cmul(w, O);
  const float ax = E.x + wO.x, ay = E.y + wO.y, bx = E.x - wO.x, by = E.y - wO.y;
  pk += fmaf(ax, ax, ay * ay);           // (the factor 1/4 is applied once, by the combine kernel)
  pm += fmaf(bx, bx, by * by);
}

template <int DETREND>
__global__ __launch_bounds__(WW * 64) void welch_wave_kernel(const float* __restrict__ x, int64_t ldx, int step,
                                                             int64_t nseg, const float* __restrict__ win,
                                                             const float2* __restrict__ tw4096, int nblk,
                                                             float* __restrict__ partial) {
  __shared__ __attribute__((aligned(16))) WelchLds L;
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  wfft::Lane lc;
  wfft::init_lane(lc, lane);
  wfft::init_tables(L.tw2l, L.tw1l, tw4096, 4096, tid, WW * 64);
  for (int i = tid; i < NSEG; i += WW * 64) L.win[i] = win[i];
  for (int i = tid; i < 8 * 64; i += WW * 64) {
    const int q = i >> 6, l = i & 63;
    const int k = wfft::bin_of(l, q >> 2, q & 3);
    L.t2048[q][l] = tw4096[2 * k];
    L.t4096[q][l] = tw4096[k];
  }
  __syncthreads();

  const int64_t b = blockIdx.y;
  const float* xr = x + b * ldx;
  const int g = blockIdx.x * WW + w;                    // this wave's slot among the nblk partial sums of the stream
  float acc[8][4];
  float a512 = 0.f, a1536 = 0.f;
#pragma unroll
  for (int q = 0; q < 8; ++q)
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[q][r] = 0.f;

  for (int64_t sg = g; sg < nseg; sg += nblk) {
    const float4* sp = reinterpret_cast<const float4*>(xr + sg * (int64_t)step);
    float4 raw[16];
#pragma unroll
    for (int a = 0; a < 16; ++a) raw[a] = sp[64 * a + lane];
    // detrend 1: subtract the mean; 2: subtract the least-squares line, written around the segment centre so that
    // slope and mean decouple (scipy.signal.detrend type='linear'): x - mean - slope (i - (n - 1) / 2)
    float mean = 0.f, slope = 0.f;
    constexpr float JC = 0.5f * (float)(NSEG - 1);
    if (DETREND) {
      float s = 0.f, sj = 0.f;
#pragma unroll
      for (int a = 0; a < 16; ++a) {
        s += (raw[a].x + raw[a].y) + (raw[a].z + raw[a].w);
        if (DETREND == 2) {
          const float i0 = (float)(4 * (64 * a + lane)) - JC;
          sj += (i0 * raw[a].x + (i0 + 1.f) * raw[a].y) + ((i0 + 2.f) * raw[a].z + (i0 + 3.f) * raw[a].w);
        }
      }
      mean = wave_sum(s) * (1.f / (float)NSEG);
      if (DETREND == 2) {
        const double nn = (double)NSEG;
        slope = (float)((double)wave_sum(sj) / (nn * (nn * nn - 1.0) / 12.0));
      }
    }
    float2 ve[16], vo[16];
    {
      const float4* w4 = reinterpret_cast<const float4*>(L.win);
#pragma unroll
      for (int a = 0; a < 16; ++a) {
        const float4 ww = w4[64 * a + lane];
        float4 r = raw[a];
        if (DETREND == 1) { r.x -= mean; r.y -= mean; r.z -= mean; r.w -= mean; }
        if (DETREND == 2) {
          const float i0 = (float)(4 * (64 * a + lane)) - JC;
          r.x -= mean + slope * i0; r.y -= mean + slope * (i0 + 1.f);
          r.z -= mean + slope * (i0 + 2.f); r.w -= mean + slope * (i0 + 3.f);
        }
        ve[a] = make_float2(r.x * ww.x, r.y * ww.y);
        vo[a] = make_float2(r.z * ww.z, r.w * ww.w);
      }
    }
    float2 ek[2][4], em[2][4], ok[2][4], om[2][4], e512, o512;
    wfft::cfft1024(ve, lc, L.sc[w], L.tw1l, L.tw2l, lane, ek, em, e512);
    wfft::cfft1024(vo, lc, L.sc[w], L.tw1l, L.tw2l, lane, ok, om, o512);
    // radix-2 combine (Z[k] = E[k] + W_2048^k O[k], Z[k + 1024] = E[k] - W_2048^k O[k]) and the real-input split
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        const int q = 4 * j + d;
        const float2 w2 = L.t2048[q][lane], w4 = L.t4096[q][lane];
        const float2 wok = cmul(w2, ok[j][d]);
        const float2 A = cadd(ek[j][d], wok), Bv = csub(ek[j][d], wok);            // Z[k], Z[1024 + k]
        const float2 wom = cmulc(om[j][d], w2);                                      // conj(W^k) O[1024 - k]
        const float2 C = csub(em[j][d], wom), D = cadd(em[j][d], wom);             // Z[1024 - k], Z[2048 - k]
        split_pow(A, D, w4, acc[q][0], acc[q][1]);                                   // bins k, 2048 - k
        split_pow(C, Bv, make_float2(-w4.y, -w4.x), acc[q][2], acc[q][3]);           // bins 1024 - k, 1024 + k
      }
    {   // bin 512 of the two transforms (lane 0): Z[512] = E - i O, Z[1536] = E + i O, split twiddle W_4096^512
      constexpr float R = 0.70710678118654752440f;
      const float2 A = make_float2(e512.x + o512.y, e512.y - o512.x), D = make_float2(e512.x - o512.y, e512.y + o512.x);
      split_pow(A, D, make_float2(R, -R), a512, a1536);
    }
  }
  float* po = partial + ((int64_t)b * nblk + g) * (int64_t)(MBIN + 1);
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      const int q = 4 * j + d;
      const int k = wfft::bin_of(lane, j, d);
      po[k] = acc[q][0];
      po[MBIN - k] = acc[q][1];
      po[1024 - k] = acc[q][2];
      if (k != 0) po[1024 + k] = acc[q][3];      // (k = 0: bin 1024 once)
    }
  if (lane == 0) { po[512] = a512; po[1536] = a1536; }
}

}  // namespace

// Partial sums of nblk waves per stream into work [B, nblk, 2049]; the values carry a factor 4 (see split_pow).
// Preconditions (checked by the caller): nperseg = nfft = 4096, 16-byte aligned segments, nblk % 4 == 0, nblk <= nseg
// rounded up to a multiple of 4 (idle waves write zeros).
int welch_wave_launch(const float* x, int64_t B, int64_t ldx, int step, int64_t nseg, const float* window,
                      const float* twiddle, int detrend, int nblk, float* work, hipStream_t st) {
  const dim3 grid((unsigned)(nblk / WW), (unsigned)B), block(WW * 64);
  if (detrend == 0)
    hipLaunchKernelGGL(welch_wave_kernel<0>, grid, block, 0, st, x, ldx, step, nseg, window, (const float2*)twiddle, nblk, work);
  else if (detrend == 1)
    hipLaunchKernelGGL(welch_wave_kernel<1>, grid, block, 0, st, x, ldx, step, nseg, window, (const float2*)twiddle, nblk, work);
  else
    hipLaunchKernelGGL(welch_wave_kernel<2>, grid, block, 0, st, x, ldx, step, nseg, window, (const float2*)twiddle, nblk, work);
  SYG_CHECK_LAUNCH("welch_wave");
  return SYG_OK;
}

}  // namespace syg
