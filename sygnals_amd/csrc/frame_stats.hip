// Time-domain frame features: one wave per frame.
//
//   rows 0..6  sygnals/core/features/time_domain.py:23-227, driven per frame by manager.py:264-286 on the
//              zero-padded (center=True: frame_length//2 each side) signal:
//              mean |x|, population std, scipy.stats.skew(bias=False), scipy.stats.kurtosis(fisher, bias=False),
//              max |x|, crest factor max|x| / sqrt(mean x^2), Shannon entropy of np.histogram(frame, num_bins)
//   row 7      RMS energy   (core/audio/features.py:73-131 -> librosa.feature.rms, zero padding)
//   row 8      zero-crossing rate (core/audio/features.py:26-71 -> librosa.feature.zero_crossing_rate:
//              EDGE padding, |x| <= 1e-10 counts as +0, the first sample of a frame never counts)
//
// The samples are float32; every sum, the histogram binning (NumPy's index-then-correct rule on float64 linspace
// edges) and the moment formulas run in float64, so the result differs from the float64 reference only by the
// final float32 rounding.  Two passes over the frame (mean first, then central moments + histogram): a frame is
// at most a few KiB and stays in L1/L2.
#include "common.h"

namespace syg {
namespace {

constexpr int FS_WAVES = 4;       // frames per workgroup
constexpr int FS_MAXBINS = 256;

__device__ __forceinline__ double wsum(double v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double wmax(double v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v = fmax(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ double wmin(double v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v = fmin(v, __shfl_xor(v, o, 64));
  return v;
}

// np.linspace(first, last, nb + 1)[j]: arange(j) * step + start with the end point set exactly; mul and add are
// separate roundings in NumPy, so FMA contraction is switched off here
__device__ __forceinline__ double edge_at(double first, double last, double step, int j, int nb) {
#pragma clang fp contract(off)
  if (j == nb) return last;
  const double m = (double)j * step;
  return m + first;
}

// STAGED: the workgroup's FS_WAVES consecutive frames overlap (hop < frame_length), so their common sample run
// [t0 hop - pad, + (FS_WAVES-1) hop + frame_length) is copied once into LDS (zero padded) and both passes of every
// frame read it from there -- the passes are bound by L2 traffic otherwise (each sample is needed
// 2 x frame_length / hop times).
template <bool STAGED>
__global__ __launch_bounds__(FS_WAVES * 64) void frame_stats_kernel(
    const float* __restrict__ y, int64_t L, int64_t ldy, int flen, int hop, int pad, int64_t T, int num_bins,
    int mask, float* __restrict__ out) {
  __shared__ unsigned hist[FS_WAVES][FS_MAXBINS];
  extern __shared__ __attribute__((aligned(16))) float stage[];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int64_t t = (int64_t)blockIdx.x * FS_WAVES + w;
  const int64_t b = blockIdx.y;
  const bool live = t < T;
  const float* yb = y + b * ldy;
  const int64_t s0 = live ? t * (int64_t)hop - pad : 0;
  const double n = (double)flen;
  const double THR = 1e-10;
  const int64_t sbase = (int64_t)blockIdx.x * FS_WAVES * hop - pad;      // first staged sample
  const int span = (FS_WAVES - 1) * hop + flen;
  if (STAGED) {
    for (int i = threadIdx.x; i < span; i += FS_WAVES * 64) {
      const int64_t s = sbase + i;
      stage[i] = (s >= 0 && s < L) ? yb[s] : 0.f;
    }
    __syncthreads();
  }
  // sample s of the clip, zero padded / edge padded
  auto at0 = [&](int64_t s) -> float {
    if (STAGED) return stage[s - sbase];
    return (s >= 0 && s < L) ? yb[s] : 0.f;
  };
  auto ate = [&](int64_t s) -> float {
    const int64_t sc = s < 0 ? 0 : (s >= L ? L - 1 : s);
    if (STAGED && sc >= sbase && sc < sbase + span) return stage[sc - sbase];
    return yb[sc];
  };

  // ---- pass 1: raw sums, extrema, zero crossings
  double sx = 0.0, sa = 0.0, sq = 0.0, mx = -1.79e308, mn = 1.79e308, pk = 0.0;
  double zc = 0.0;
  if (live) {
    for (int i = lane; i < flen; i += 64) {
      const int64_t s = s0 + i;
      const double x = (double)at0(s);                                    // zero padding
      sx += x; sa += fabs(x); sq += x * x;
      mx = fmax(mx, x); mn = fmin(mn, x); pk = fmax(pk, fabs(x));
      if ((mask & 256) && i >= 1) {                                       // edge padding for the ZCR
        double a = (double)ate(s - 1), c = (double)ate(s);
        a = (fabs(a) <= THR) ? 0.0 : a;
        c = (fabs(c) <= THR) ? 0.0 : c;
        zc += ((a < 0.0) != (c < 0.0)) ? 1.0 : 0.0;
      }
    }
  }
  sx = wsum(sx); sa = wsum(sa); sq = wsum(sq); zc = wsum(zc);
  mx = wmax(mx); mn = wmin(mn); pk = wmax(pk);
  const double mean = sx / n;

  // ---- pass 2: central moments and the histogram
  const int nb = num_bins;
  for (int i = lane; i < nb; i += 64) hist[w][i] = 0u;
  __syncthreads();
  double m2 = 0.0, m3 = 0.0, m4 = 0.0;
  const bool constant = !(mx > mn);
  const bool want_hist = (mask & 64) && !constant && flen >= 2 && nb >= 1;
  const double first = mn, last = mx;
  const double step = (last - first) / (double)nb;          // linspace: delta / div
  const double norm = (double)nb / (last - first);
  if (live) {
    for (int i = lane; i < flen; i += 64) {
      const int64_t s = s0 + i;
      const double x = (double)at0(s);
      const double d = x - mean;
      const double d2 = d * d;
      m2 += d2; m3 += d2 * d; m4 += d2 * d2;
      if (want_hist) {
        int idx = (int)((x - first) * norm);                // astype(intp) truncates; the offset is >= 0
        if (idx == nb) idx -= 1;
        if (x < edge_at(first, last, step, idx, nb)) idx -= 1;
        if (idx != nb - 1 && x >= edge_at(first, last, step, idx + 1, nb)) idx += 1;
        atomicAdd(&hist[w][idx], 1u);
      }
    }
  }
  m2 = wsum(m2) / n; m3 = wsum(m3) / n; m4 = wsum(m4) / n;
  __syncthreads();
  double ent = 0.0;
  if (want_hist) {
    for (int i = lane; i < nb; i += 64) {
      const unsigned c = hist[w][i];
      if (c > 0u) {
        const double p = (double)c / n;                     // counts sum to n: scipy's renormalisation is exact
        ent -= p * log(p);
      }
    }
    ent = wsum(ent);
  }
  if (!live || lane != 0) return;

  const double EPS = 2.220446049250313e-16;
  const double var = m2;                                     // np.var
  double skew = 0.0, kurt = 0.0;
  if (flen >= 2 && var >= EPS) {
    const double g1 = m3 / (m2 * sqrt(m2));
    skew = (flen > 2) ? sqrt(n * (n - 1.0)) / (n - 2.0) * g1 : g1;
  }
  if (flen >= 4 && var >= EPS) kurt = (n - 1.0) / ((n - 2.0) * (n - 3.0)) * ((n + 1.0) * m4 / (m2 * m2) - 3.0 * (n - 1.0));
  const double rms = sqrt(sq / n);
  float* o = out + b * (int64_t)SYG_NFSTAT * T + t;
  if (mask & 1) o[0 * T] = (float)(sa / n);
  if (mask & 2) o[1 * T] = (float)sqrt(m2);
  if (mask & 4) o[2 * T] = (float)skew;
  if (mask & 8) o[3 * T] = (float)kurt;
  if (mask & 16) o[4 * T] = (float)pk;
  if (mask & 32) o[5 * T] = (float)((rms < EPS) ? 0.0 : pk / rms);
  if (mask & 64) o[6 * T] = (float)ent;
  if (mask & 128) o[7 * T] = (float)rms;
  if (mask & 256) o[8 * T] = (float)(zc / n);
}

// librosa.feature.rms(S=...): sqrt(2 sum_k w_k |S[k]|^2 / frame_length^2), w = 1/2 for DC (and for Nyquist when
// frame_length is even); S is [rows, F] magnitudes, one wave per row
__global__ __launch_bounds__(256) void rms_spec_kernel(const float* __restrict__ S, int64_t rows, int F, int flen,
                                                        float* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  const float* s = S + r * (int64_t)F;
  double acc = 0.0;
  for (int k = lane; k < F; k += 64) {
    const double v = (double)s[k];
    const double wgt = (k == 0 || (k == F - 1 && (flen % 2 == 0))) ? 0.5 : 1.0;
    acc += wgt * v * v;
  }
  acc = wsum(acc);
  if (lane == 0) out[r] = (float)sqrt(2.0 * acc / ((double)flen * (double)flen));
}

}  // namespace
}  // namespace syg

using namespace syg;

extern "C" int syg_frame_stats_f32(const float* y, int64_t B, int64_t L, int64_t ldy, int frame_length, int hop,
                                   int center, int64_t T, int num_bins, int mask, float* out, void* stream) {
  SYG_REQUIRE(y && out, "frame_stats: null pointer argument");
  SYG_REQUIRE(B >= 1 && L >= 1 && ldy >= L, "frame_stats: need B >= 1, L >= 1, ldy >= L");
  SYG_REQUIRE(frame_length >= 1 && hop >= 1, "frame_stats: frame_length and hop must be >= 1");
  const int64_t Texp = center ? 1 + L / hop : (L >= frame_length ? 1 + (L - frame_length) / hop : 0);
  SYG_REQUIRE(T >= 1 && T == Texp, "frame_stats: T=%lld does not match the framing rule (%lld)", (long long)T,
              (long long)Texp);
  SYG_REQUIRE(mask > 0 && mask < (1 << SYG_NFSTAT), "frame_stats: mask must select at least one of the %d rows",
              SYG_NFSTAT);
  SYG_REQUIRE(num_bins >= 1 && num_bins <= FS_MAXBINS, "frame_stats: num_bins must be in [1, %d] (got %d)", FS_MAXBINS,
              num_bins);
  SYG_REQUIRE(B <= 65535, "frame_stats: at most 65535 clips per call");
  const int64_t gx = (T + FS_WAVES - 1) / FS_WAVES;
  SYG_REQUIRE(gx < (int64_t)0x7fffffff, "frame_stats: too many frames");
  const size_t span_bytes = ((size_t)(FS_WAVES - 1) * hop + frame_length) * sizeof(float);
  const int pad = center ? frame_length / 2 : 0;
  if (hop < frame_length && span_bytes <= 48 * 1024) {
    hipLaunchKernelGGL(frame_stats_kernel<true>, dim3((unsigned)gx, (unsigned)B), dim3(FS_WAVES * 64), span_bytes,
                       (hipStream_t)stream, y, L, ldy, frame_length, hop, pad, T, num_bins, mask, out);
  } else {
    hipLaunchKernelGGL(frame_stats_kernel<false>, dim3((unsigned)gx, (unsigned)B), dim3(FS_WAVES * 64), 0,
                       (hipStream_t)stream, y, L, ldy, frame_length, hop, pad, T, num_bins, mask, out);
  }
  SYG_CHECK_LAUNCH("frame_stats");
  return SYG_OK;
}

extern "C" int syg_rms_from_spec_f32(const float* S, int64_t rows, int F, int frame_length, float* out, void* stream) {
  SYG_REQUIRE(S && out, "rms_from_spec: null pointer argument");
  SYG_REQUIRE(rows >= 1 && F >= 1 && frame_length >= 1, "rms_from_spec: need rows, F, frame_length >= 1");
  const int64_t gx = (rows + 3) / 4;
  SYG_REQUIRE(gx < (int64_t)0x7fffffff, "rms_from_spec: too many rows");
  hipLaunchKernelGGL(rms_spec_kernel, dim3((unsigned)gx), dim3(256), 0, (hipStream_t)stream, S, rows, F, frame_length, out);
  SYG_CHECK_LAUNCH("rms_from_spec");
  return SYG_OK;
}
