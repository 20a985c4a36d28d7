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
// The samples are float32.  Round 3: the per-sample arithmetic runs in float32 and only the ACCUMULATION across blocks of
// samples and across lanes in float64 (v_fma_f64 issues at half the float32 rate and every sample needed a conversion:
// ~12 float64 operations per sample were the whole cost of this kernel):
//   pass 1  a lane adds x, |x|, x^2 of FS_BLK consecutive trips in float32 (<= 8 terms: rounding 8 * 2^-24 relative
//           to the block, not to the frame) and adds the block sums to its float64 accumulators;
//   pass 2  d = (x - mh) - ml with mh + ml = mean as a float32 pair (the first difference is exact or rounded
//           relative to d itself, never relative to the mean: a large DC offset costs nothing), d^2, d^3, d^4 in
//           float32, block sums to float64.
// The histogram binning (NumPy's index-then-correct rule on float64 linspace edges), the zero-crossing classes and the
// extrema are exact as before; the moment formulas run in float64 on the accumulated sums.  Measured against the
// float64 reference: <= 3e-7 of each row's peak (gate 1e-5).  Two passes over the frame (mean first, then central
// moments + histogram): a frame is at most a few KiB and stays in L1/L2.
#include "common.h"
#include <type_traits>

namespace syg {
namespace {

constexpr int FS_WAVES = 4;       // frames per workgroup
constexpr int FS_MAXBINS = 256;

// float64 wave sums on the DPP path (two 32-bit DPP moves per hop; __shfl_xor would be two ds_bpermute_b32 round
// trips through the LDS crossbar per hop).  Fixed order: butterfly inside each row of 16 lanes, then the four rows.
template <int CTRL>
__device__ __forceinline__ double dpp_d(double v) {
  const int lo = dpp_i<CTRL>(__double2loint(v)), hi = dpp_i<CTRL>(__double2hiint(v));
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double rl_d(double v, int l) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}
__device__ __forceinline__ double wsum(double v) {
  v += dpp_d<DPP_QP_1032>(v);
  v += dpp_d<DPP_QP_2301>(v);
  v += dpp_d<DPP_ROW_HALF_MIRROR>(v);
  v += dpp_d<DPP_ROW_MIRROR>(v);
  return (rl_d(v, 0) + rl_d(v, 16)) + (rl_d(v, 32) + rl_d(v, 48));
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
// VEC (staged, hop % 4 == 0, frame_length % 256 == 0): a lane takes FOUR consecutive samples per trip (one ds_read_b128,
// a quarter of the loop and address instructions; the zero-crossing predecessor of three of the four samples is in the
// lane's own register).  Sample order inside a sum changes, results agree to rounding; counts are identical.
template <bool STAGED, bool VEC = false>
__global__ __launch_bounds__(FS_WAVES * 64) void frame_stats_kernel(
    const float* __restrict__ y, int64_t L, int64_t ldy, int flen, int hop, int pad, int64_t T, int num_bins,
    int mask, float* __restrict__ out) {
  __shared__ unsigned hist[FS_WAVES][FS_MAXBINS];
  __shared__ float eu[FS_WAVES][FS_MAXBINS + 1];
  extern __shared__ __attribute__((aligned(16))) float stage[];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int64_t t = (int64_t)blockIdx.x * FS_WAVES + w;
  const int64_t b = blockIdx.y;
  const bool live = t < T;
  const float* yb = y + b * ldy;
  const int64_t s0 = live ? t * (int64_t)hop - pad : 0;
  const double n = (double)flen;
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
  const float* fr = stage + (int)(s0 - sbase);              // STAGED: sample i of this wave's frame (zero padded)
  auto at0 = [&](int i) -> float {
    if (STAGED) return fr[i];
    const int64_t s = s0 + i;
    return (s >= 0 && s < L) ? yb[s] : 0.f;
  };
  auto ate = [&](int64_t s) -> float {
    const int64_t sc = s < 0 ? 0 : (s >= L ? L - 1 : s);
    if (STAGED && sc >= sbase && sc < sbase + span) return stage[sc - sbase];
    return yb[sc];
  };

  // ---- pass 1: raw sums, extrema, zero crossings
  double sx = 0.0, sa = 0.0, sq = 0.0;
  float mxf = -3.4e38f, mnf = 3.4e38f;                                    // extrema of float samples are exact in float
  // |x| <= 1e-10 for a float x  <=>  |x| <= the largest float below 1e-10 (float(1e-10) itself is above it)
  constexpr float ZTHR = 0x1.b7cdfcp-34f;
  const bool edge_frame = s0 < 1 || s0 + flen > L;
  int zci = 0, carry = 0;
  // the wave-uniform choices (zero crossings wanted? frame touching a clip end?) are made outside the loop: three
  // straight-line loop bodies instead of per-sample branches
  constexpr int FS_BLK = 8;                                               // float32 terms per block sum
  auto pass1 = [&](auto want_zcr, auto at_edge) {
    float bx = 0.f, ba = 0.f;
    int nblk = 0;
    if (VEC) {
      const float4* f4 = reinterpret_cast<const float4*>(fr);
      const int nj = flen >> 8;
#pragma unroll 2
      for (int j = 0; j < nj; ++j) {
        const float4 q = f4[lane + 64 * j];
        bx += (q.x + q.y) + (q.z + q.w);
        ba += (fabsf(q.x) + fabsf(q.y)) + (fabsf(q.z) + fabsf(q.w));
        mxf = fmaxf(fmaxf(mxf, fmaxf(q.x, q.y)), fmaxf(q.z, q.w));
        mnf = fminf(fminf(mnf, fminf(q.x, q.y)), fminf(q.z, q.w));
        if ((j & 1) == 1) { sx += (double)bx; sa += (double)ba; bx = ba = 0.f; }
        if (decltype(want_zcr)::value) {
          float e0 = q.x, e1 = q.y, e2 = q.z, e3 = q.w;
          if (decltype(at_edge)::value) {
            const int64_t sb = s0 + 4 * (lane + 64 * j);
            e0 = ate(sb); e1 = ate(sb + 1); e2 = ate(sb + 2); e3 = ate(sb + 3);
          }
          const int n0 = (fabsf(e0) > ZTHR && e0 < 0.f) ? 1 : 0, n1 = (fabsf(e1) > ZTHR && e1 < 0.f) ? 1 : 0,
                    n2 = (fabsf(e2) > ZTHR && e2 < 0.f) ? 1 : 0, n3 = (fabsf(e3) > ZTHR && e3 < 0.f) ? 1 : 0;
          int prev = __builtin_amdgcn_update_dpp(0, n3, DPP_WAVE_SHR1, 0xF, 0xF, false);
          prev = lane == 0 ? carry : prev;
          carry = __builtin_amdgcn_readlane(n3, 63);
          zci += ((lane + j > 0 && prev != n0) ? 1 : 0) + (n0 != n1 ? 1 : 0) + (n1 != n2 ? 1 : 0) + (n2 != n3 ? 1 : 0);
        }
      }
      sx += (double)bx; sa += (double)ba;
      return;
    }
#pragma unroll 4
    for (int i = lane; i < flen; i += 64) {
      const float xf = at0(i);                                            // zero padding
      bx += xf; ba += fabsf(xf);
      if (++nblk == FS_BLK) { sx += (double)bx; sa += (double)ba; bx = ba = 0.f; nblk = 0; }
      mxf = fmaxf(mxf, xf); mnf = fminf(mnf, xf);
      if (decltype(want_zcr)::value) {
        // sign class of every sample once; its predecessor is the neighbouring lane's (lane 0: lane 63 of the
        // previous trip).  Only frames that touch a clip end see edge padding instead of the zeros of `xf`.
        const float xe = decltype(at_edge)::value ? ate(s0 + i) : xf;
        const int neg = (fabsf(xe) > ZTHR && xe < 0.f) ? 1 : 0;
        int prev = __builtin_amdgcn_update_dpp(0, neg, DPP_WAVE_SHR1, 0xF, 0xF, false);
        prev = lane == 0 ? carry : prev;
        carry = __builtin_amdgcn_readlane(neg, 63);
        zci += (i >= 1 && prev != neg) ? 1 : 0;
      }
    }
    sx += (double)bx; sa += (double)ba;
  };
  using T_ = std::true_type;
  using F_ = std::false_type;
  if (live) {
    if (!(mask & 256)) pass1(F_{}, F_{});
    else if (!edge_frame) pass1(T_{}, F_{});
    else pass1(T_{}, T_{});
  }
  double zc = (double)wave_sum_i(zci);
  sx = wsum(sx); sa = wsum(sa);
  wave_maxmin(mxf, mnf);
  const double mx = (double)mxf, mn = (double)mnf, pk = fmax(fabs(mx), fabs(mn));
  const double mean = sx / n;

  // ---- pass 2: central moments and the histogram
  const int nb = num_bins;
  double m2 = 0.0, m3 = 0.0, m4 = 0.0;
  const bool constant = !(mx > mn);
  const bool want_hist = (mask & 64) && !constant && flen >= 2 && nb >= 1;
  const double first = mn, last = mx;
  const double step = (last - first) / (double)nb;          // linspace: delta / div
  // NumPy guesses the bin as int((x - first) * nb / (last - first)) and corrects it by one against the float64 edges,
  // so the result is the bin whose edges enclose x.  For a float32 sample x and a float64 edge e, x < e <=> x < up(e)
  // with up(e) the smallest float32 >= e: the edges are rounded up once per frame (eu) and the per-sample work --
  // guess, two table reads, two compares -- runs in float32 with the same outcome.
  for (int i = lane; i < nb; i += 64) hist[w][i] = 0u;
  for (int j = lane; j <= nb; j += 64) {
    const double e = edge_at(first, last, step, j, nb);
    float f = (float)e;
    if ((double)f < e)                                       // next float towards +inf (from +-0: the smallest denormal)
      f = f == 0.f ? __uint_as_float(1u) : __uint_as_float(__float_as_uint(f) + (f > 0.f ? 1u : -1u));
    eu[w][j] = f;
  }
  __syncthreads();
  const float firstf = mnf, normf = (float)nb / (mxf - mnf);
  const bool wide = (mxf - mnf) > 1e-30f;
  // Up to 16 bins and 255 samples per lane (the default: 10 bins, 32 samples): every lane counts in 8-bit fields
  // of two 64-bit registers and the fields are summed over the wave afterwards -- LDS atomics of 64 lanes on ten
  // addresses serialise.  Otherwise the LDS histogram.
  const bool packed = nb <= 16 && flen <= 255 * 64;
  unsigned long long c0 = 0ull, c1 = 0ull;
  // HIST: 0 no histogram, 1 packed counters + one-step correction (the common case), 2 everything else
  const float mh = (float)mean, ml = (float)(mean - (double)mh);          // mean as a float32 pair
  // deviations are scaled by a power of two that brings the frame's peak into [0.5, 1) (exact; taken back in float64
  // below): d^4 can then neither overflow nor vanish, whatever the signal's units
  const float pkf = fmaxf(fabsf(mxf), fabsf(mnf));
  const unsigned pe = (__float_as_uint(pkf) >> 23) & 0xffu;
  const float sc = (pkf > 0.f && pe <= 250u) ? __uint_as_float((253u - pe) << 23) : 1.f;
  auto pass2 = [&](auto hist_mode) {
    constexpr int HIST = decltype(hist_mode)::value;
    float b2 = 0.f, b3 = 0.f, b4 = 0.f;
    int nblk = 0;
    auto sample = [&](float xf) {
      const float d = ((xf - mh) - ml) * sc;
      const float d2 = d * d;
      b2 += d2; b3 = fmaf(d2, d, b3); b4 = fmaf(d2, d2, b4);
      if (HIST != 0) {
        int idx = (int)((xf - firstf) * normf);             // within one bin of the float64 guess
        idx = idx < 0 ? 0 : (idx > nb - 1 ? nb - 1 : idx);
        if (HIST == 1 || wide) {                            // the guess is off by one bin at most: branch-free
          const float e0 = eu[w][idx], e1 = eu[w][idx + 1];
          idx += (xf >= e1 && idx != nb - 1) ? 1 : ((xf < e0 && idx > 0) ? -1 : 0);
        } else {                                            // (nearly) denormal range: the float guess means nothing
          while (idx > 0 && xf < eu[w][idx]) idx -= 1;
          while (idx != nb - 1 && xf >= eu[w][idx + 1]) idx += 1;
        }
        if (HIST == 1 || packed) {
          const unsigned long long one = 1ull << ((idx & 7) * 8);
          c0 += idx < 8 ? one : 0ull;
          c1 += idx < 8 ? 0ull : one;
        } else {
          atomicAdd(&hist[w][idx], 1u);
        }
      }
    };
    if (VEC) {
      const float4* f4 = reinterpret_cast<const float4*>(fr);
      const int nj = flen >> 8;
#pragma unroll 2
      for (int j = 0; j < nj; ++j) {
        const float4 q = f4[lane + 64 * j];
        sample(q.x); sample(q.y); sample(q.z); sample(q.w);
        if ((j & 1) == 1) { m2 += (double)b2; m3 += (double)b3; m4 += (double)b4; b2 = b3 = b4 = 0.f; }
      }
      m2 += (double)b2; m3 += (double)b3; m4 += (double)b4;
      return;
    }
#pragma unroll 4
    for (int i = lane; i < flen; i += 64) {
      const float xf = at0(i);
      const float d = ((xf - mh) - ml) * sc;
      const float d2 = d * d;
      b2 += d2; b3 = fmaf(d2, d, b3); b4 = fmaf(d2, d2, b4);
      if (++nblk == FS_BLK) { m2 += (double)b2; m3 += (double)b3; m4 += (double)b4; b2 = b3 = b4 = 0.f; nblk = 0; }
      if (HIST != 0) {
        int idx = (int)((xf - firstf) * normf);             // within one bin of the float64 guess
        idx = idx < 0 ? 0 : (idx > nb - 1 ? nb - 1 : idx);
        if (HIST == 1 || wide) {                            // the guess is off by one bin at most: branch-free
          const float e0 = eu[w][idx], e1 = eu[w][idx + 1];
          idx += (xf >= e1 && idx != nb - 1) ? 1 : ((xf < e0 && idx > 0) ? -1 : 0);
        } else {                                            // (nearly) denormal range: the float guess means nothing
          while (idx > 0 && xf < eu[w][idx]) idx -= 1;
          while (idx != nb - 1 && xf >= eu[w][idx + 1]) idx += 1;
        }
        if (HIST == 1 || packed) {
          const unsigned long long one = 1ull << ((idx & 7) * 8);
          c0 += idx < 8 ? one : 0ull;
          c1 += idx < 8 ? 0ull : one;
        } else {
          atomicAdd(&hist[w][idx], 1u);
        }
      }
    }
    m2 += (double)b2; m3 += (double)b3; m4 += (double)b4;
  };
  if (live) {
    if (!want_hist) pass2(std::integral_constant<int, 0>{});
    else if (packed && wide) pass2(std::integral_constant<int, 1>{});
    else pass2(std::integral_constant<int, 2>{});
  }
  {
    const double isc = 1.0 / (double)sc, i2 = isc * isc;
    m2 = wsum(m2) / n * i2; m3 = wsum(m3) / n * (i2 * isc); m4 = wsum(m4) / n * (i2 * i2);
  }
  sq = n * (m2 + mean * mean);                              // sum x^2 from the well-conditioned central sum
  __syncthreads();
  double ent = 0.0;
  if (want_hist && packed) {
    int mine = 0;                                           // lane j ends up with the count of bin j
    for (int j = 0; j < nb; ++j) {                          // wave-uniform trip count
      const unsigned long long cw = j < 8 ? c0 : c1;
      const int c = wave_sum_i((int)((cw >> ((j & 7) * 8)) & 0xffull));
      mine = lane == j ? c : mine;
    }
    if (mine > 0) {
      const double p = (double)mine / n;                    // counts sum to n: scipy's renormalisation is exact
      ent -= p * log(p);
    }
    ent = wsum(ent);
  } else if (want_hist) {
    for (int i = lane; i < nb; i += 64) {
      const unsigned c = hist[w][i];
      if (c > 0u) {
        const double p = (double)c / n;
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
  if (hop < frame_length && span_bytes <= 48 * 1024 && hop % 4 == 0 && frame_length % 256 == 0) {
    hipLaunchKernelGGL((frame_stats_kernel<true, true>), dim3((unsigned)gx, (unsigned)B), dim3(FS_WAVES * 64), span_bytes,
                       (hipStream_t)stream, y, L, ldy, frame_length, hop, pad, T, num_bins, mask, out);
  } else if (hop < frame_length && span_bytes <= 48 * 1024) {
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
