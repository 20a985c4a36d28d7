// Stand-alone per-frame spectral statistics and spectral-contrast tail means on frame-major
// magnitude spectra mag [N, F] resident in HBM (the generic path: any F, any bin-frequency
// table).  One wave per frame, coalesced row reads, wave-level reductions and scans.
//
// Formulas follow sygnals/core/features/frequency_domain.py:
//   spectral_centroid :24-74, spectral_bandwidth :76-145, spectral_flatness :214-271,
//   spectral_rolloff :274-351 (on POWER, first bin with cumsum >= roll*total),
//   dominant_frequency :354-386, and librosa.feature.spectral_contrast (:200-207).
#include "common.h"

namespace syg {
namespace {

constexpr float EPS64 = 2.220446049250313e-16f;

__global__ __launch_bounds__(256) void spectral_stats_kernel(const float* __restrict__ mag, int64_t N, int F,
                                                             const float* __restrict__ freqs, float roll_percent,
                                                             float bw_p, float* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= N) return;
  const float* m = mag + row * F;
  const int ch = (F + 63) / 64;          // lane owns the contiguous bins [lane*ch, lane*ch + ch)
  const int b0 = lane * ch, b1 = min(F, b0 + ch);
  float msum = 0.f, fsum = 0.f, psum = 0.f, lsum = 0.f, mmax = -1.f;
  int amax = 0;
  for (int k = b0; k < b1; ++k) {
    const float v = fabsf(m[k]);
    msum += v;
    fsum = fmaf(v, freqs[k], fsum);
    psum = fmaf(v, v, psum);
    lsum += logf(v + EPS64);
    if (v > mmax) { mmax = v; amax = k; }
  }
  const float tot_m = wave_sum(msum), tot_f = wave_sum(fsum), tot_p = wave_sum(psum), tot_l = wave_sum(lsum);
  const float gm = wave_max(mmax);
  const int cand = wave_min_i((mmax == gm && b0 < b1) ? amax : 0x7fffffff);
  const bool live = tot_m >= EPS64;
  const float cen = live ? tot_f / tot_m : 0.f;
  float dsum = 0.f;
  for (int k = b0; k < b1; ++k) {
    const float d = fabsf(freqs[k] - cen);
    dsum = fmaf(fabsf(m[k]), (bw_p == 2.f) ? d * d : powf(d, bw_p), dsum);
  }
  const float tot_d = wave_sum(dsum);
  const float excl = wave_excl_scan(psum, lane);
  const float thr = roll_percent * tot_p;
  int rb = 0x7fffffff;
  float margin = 3.4e38f;
  {
    float c = excl;
    for (int k = b0; k < b1; ++k) {
      const float v = m[k];
      const float cprev = c;
      c = fmaf(v, v, c);
      if (c >= thr && rb == 0x7fffffff) {
        rb = k;
        margin = fminf(c - thr, (k > 0) ? thr - cprev : 3.4e38f);
      }
    }
  }
  int rbmin = wave_min_i(rb);
  const float mg = wave_min((rb == rbmin && rb != 0x7fffffff) ? margin : 3.4e38f);
  if (rbmin == 0x7fffffff || tot_p < EPS64) rbmin = F - 1;
  if (lane == 0) {
    const float am = tot_m / (float)F;
    float flat = 0.f;
    if (am >= EPS64) flat = fminf(fmaxf(expf(tot_l / (float)F) / am, 0.f), 1.f);
    float bw = 0.f;
    if (live) bw = (bw_p == 2.f) ? sqrtf(fmaxf(tot_d / tot_m, 0.f)) : powf(fmaxf(tot_d / tot_m, 0.f), 1.f / bw_p);
    out[SYG_STAT_CENTROID * N + row] = cen;
    out[SYG_STAT_BANDWIDTH * N + row] = bw;
    out[SYG_STAT_FLATNESS * N + row] = flat;
    out[SYG_STAT_ROLLOFF_BIN * N + row] = (float)rbmin;
    out[SYG_STAT_DOMINANT_BIN * N + row] = (float)cand;
    out[SYG_STAT_MAG_SUM * N + row] = tot_m;
    out[SYG_STAT_POWER_SUM * N + row] = tot_p;
    out[SYG_STAT_ROLLOFF_MARGIN * N + row] = (tot_p > 0.f) ? mg / tot_p : 0.f;
  }
}

struct CPlan {
  int n_rows;
  int lo[SYG_MAX_BANDS];
  int hi[SYG_MAX_BANDS];
  int k[SYG_MAX_BANDS];
};

// k-th order statistic of non-negative floats by a 32-step radix select on their bit patterns
__device__ uint32_t kth_bits(const float* __restrict__ v, int n, int kk, bool largest, int lane) {
  uint32_t prefix = 0;
  int remaining = kk;
  for (int bit = 31; bit >= 0; --bit) {
    const uint32_t mask = ~((1u << bit) - 1u);
    const uint32_t want = largest ? (prefix | (1u << bit)) : prefix;
    int cnt = 0;
    for (int i = lane; i < n; i += 64) cnt += ((__float_as_uint(fabsf(v[i])) & mask) == want) ? 1 : 0;
    cnt = wave_sum_i(cnt);
    if (largest) {
      if (cnt >= remaining) prefix |= (1u << bit); else remaining -= cnt;
    } else {
      if (cnt < remaining) { remaining -= cnt; prefix |= (1u << bit); }
    }
  }
  return prefix;
}

__global__ __launch_bounds__(256) void contrast_pv_kernel(const float* __restrict__ mag, int64_t N, int F, CPlan cp,
                                                          float* __restrict__ out) {
  __shared__ int pl[3 * SYG_MAX_BANDS];
#pragma unroll
  for (int r = 0; r < SYG_MAX_BANDS; ++r)
    if (threadIdx.x == r) { pl[r] = cp.lo[r]; pl[SYG_MAX_BANDS + r] = cp.hi[r]; pl[2 * SYG_MAX_BANDS + r] = cp.k[r]; }
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= N) return;
  const float* m = mag + row * F;
  for (int r = 0; r < cp.n_rows; ++r) {
    const int lo = pl[r], n = pl[SYG_MAX_BANDS + r] - lo, k = pl[2 * SYG_MAX_BANDS + r];
    const float* v = m + lo;
    const uint32_t tlo = kth_bits(v, n, k, false, lane), thi = kth_bits(v, n, k, true, lane);
    float slo = 0.f, shi = 0.f;
    int clo = 0, chi = 0;
    for (int i = lane; i < n; i += 64) {
      const float x = fabsf(v[i]);
      const uint32_t u = __float_as_uint(x);
      if (u < tlo) { slo += x; ++clo; }
      if (u > thi) { shi += x; ++chi; }
    }
    slo = wave_sum(slo); shi = wave_sum(shi);
    clo = wave_sum_i(clo); chi = wave_sum_i(chi);
    if (lane == 0) {
      out[((int64_t)0 * cp.n_rows + r) * N + row] = (shi + (float)(k - chi) * __uint_as_float(thi)) / (float)k;
      out[((int64_t)1 * cp.n_rows + r) * N + row] = (slo + (float)(k - clo) * __uint_as_float(tlo)) / (float)k;
    }
  }
}


// contrast[b, r, t] = power_to_db(peak)[r, t] - power_to_db(valley)[r, t], each power_to_db with
// ref = 1, amin, and the top_db clamp relative to the maximum of its own [R, T] matrix
// (librosa.feature.spectral_contrast, linear=False).  One workgroup per clip.
__global__ __launch_bounds__(256) void contrast_db_kernel(const float* __restrict__ pv, int R, int64_t T, float amin,
                                                          float top_db, float* __restrict__ out) {
  __shared__ float red[2][4];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int64_t b = blockIdx.x, n = (int64_t)R * T;
  const float* pk = pv + (b * 2 + 0) * n;
  const float* vl = pv + (b * 2 + 1) * n;
  if (amin <= 0.f) {                      // linear=True: plain difference of the means
    for (int64_t i = tid; i < n; i += 256) out[b * n + i] = pk[i] - vl[i];
    return;
  }
  float m0 = 0.f, m1 = 0.f;
  for (int64_t i = tid; i < n; i += 256) { m0 = fmaxf(m0, pk[i]); m1 = fmaxf(m1, vl[i]); }
  m0 = wave_max(m0); m1 = wave_max(m1);
  if (lane == 0) { red[0][w] = m0; red[1][w] = m1; }
  __syncthreads();
  m0 = fmaxf(fmaxf(red[0][0], red[0][1]), fmaxf(red[0][2], red[0][3]));
  m1 = fmaxf(fmaxf(red[1][0], red[1][1]), fmaxf(red[1][2], red[1][3]));
  const float f0 = (top_db >= 0.f) ? 10.f * log10f(fmaxf(amin, m0)) - top_db : -3.4e38f;
  const float f1 = (top_db >= 0.f) ? 10.f * log10f(fmaxf(amin, m1)) - top_db : -3.4e38f;
  for (int64_t i = tid; i < n; i += 256) {
    const float a = fmaxf(10.f * log10f(fmaxf(amin, pk[i])), f0);
    const float c = fmaxf(10.f * log10f(fmaxf(amin, vl[i])), f1);
    out[b * n + i] = a - c;
  }
}

}  // namespace
}  // namespace syg

using namespace syg;

extern "C" int syg_spectral_stats_f32(const float* mag, int64_t N, int F, const float* freqs, float roll_percent,
                                      float bw_p, float* stats_out, void* stream) {
  SYG_REQUIRE(mag && freqs && stats_out, "spectral_stats: null pointer argument");
  SYG_REQUIRE(N >= 1 && F >= 1, "spectral_stats: need N >= 1 and F >= 1");
  SYG_REQUIRE(roll_percent >= 0.f && roll_percent <= 1.f, "roll_percent must be between 0.0 and 1.0.");
  SYG_REQUIRE(bw_p > 0.f, "Order 'p' for spectral bandwidth must be positive.");
  SYG_REQUIRE((N + 3) / 4 < (int64_t)0x7fffffff, "spectral_stats: grid too large");
  hipLaunchKernelGGL(spectral_stats_kernel, dim3((unsigned)((N + 3) / 4)), dim3(256), 0, (hipStream_t)stream, mag, N,
                     F, freqs, roll_percent, bw_p, stats_out);
  SYG_CHECK_LAUNCH("spectral_stats");
  return SYG_OK;
}

extern "C" int syg_contrast_pv_f32(const float* mag, int64_t N, int F, const int32_t* cplan_host, float* out,
                                   void* stream) {
  SYG_REQUIRE(mag && cplan_host && out, "contrast_pv: null pointer argument");
  SYG_REQUIRE(N >= 1 && F >= 1, "contrast_pv: need N >= 1 and F >= 1");
  CPlan cp;
  cp.n_rows = cplan_host[0];
  SYG_REQUIRE(cp.n_rows >= 1 && cp.n_rows <= SYG_MAX_BANDS, "contrast_pv: rows must be in [1, %d]", SYG_MAX_BANDS);
  for (int r = 0; r < SYG_MAX_BANDS; ++r) {
    cp.lo[r] = cplan_host[1 + r];
    cp.hi[r] = cplan_host[1 + SYG_MAX_BANDS + r];
    cp.k[r] = cplan_host[1 + 2 * SYG_MAX_BANDS + r];
    if (r < cp.n_rows)
      SYG_REQUIRE(cp.lo[r] >= 0 && cp.hi[r] <= F && cp.lo[r] < cp.hi[r] && cp.k[r] >= 1 &&
                      cp.k[r] <= cp.hi[r] - cp.lo[r],
                  "contrast_pv: band %d invalid (lo=%d hi=%d k=%d, F=%d)", r, cp.lo[r], cp.hi[r], cp.k[r], F);
  }
  hipLaunchKernelGGL(contrast_pv_kernel, dim3((unsigned)((N + 3) / 4)), dim3(256), 0, (hipStream_t)stream, mag, N, F,
                     cp, out);
  SYG_CHECK_LAUNCH("contrast_pv");
  return SYG_OK;
}

extern "C" int syg_contrast_db_f32(const float* pv, int64_t B, int R, int64_t T, float amin, float top_db,
                                   float* out, void* stream) {
  SYG_REQUIRE(pv && out, "contrast_db: null pointer argument");
  SYG_REQUIRE(B >= 1 && B < (int64_t)0x7fffffff && R >= 1 && T >= 1, "contrast_db: bad shape");
  SYG_REQUIRE(amin >= 0.f, "contrast_db: amin must be positive (or 0 for the linear difference)");
  hipLaunchKernelGGL(contrast_db_kernel, dim3((unsigned)B), dim3(256), 0, (hipStream_t)stream, pv, R, T, amin, top_db,
                     out);
  SYG_CHECK_LAUNCH("contrast_db");
  return SYG_OK;
}
