// FFT-backed 1-D signal operations around the transform kernels:
//
//   convolution / correlation   sygnals/core/dsp.py:294-337 (scipy.signal.fftconvolve), :342-394
//                               (scipy.signal.correlate) and :396-433 (autocorrelation)
//   periodogram                 sygnals/core/dsp.py:438-498 (scipy.signal.periodogram)
//   analytic signal / envelope  sygnals/core/transforms.py:119-151, dsp.py:565-636 (scipy.signal.hilbert)
//
// Everything here is HBM-bound element-wise work between two passes of the power-of-two complex FFT
// (fft_generic.hip).  Real sequences use the packed real transform: a zero-padded real row of M floats IS the
// complex row z[n] = x[2n] + i x[2n+1] of M/2 elements, so "packing" is a copy, the product of two real spectra is
// formed directly on the packed transforms (one kernel untangles both spectra, multiplies and re-tangles) and the
// inverse transform of the result IS the real output row -- half the FFT work and no separate real/complex
// conversion passes.  The even/odd split keeps each transform's rounding relative to its own signal, which pairing
// two different signals in one complex transform would not.
#include "common.h"

namespace syg {
namespace {

constexpr int SUMCH = 64;         // partial sums per row for the mean (detrend='constant')

// part[r, c] = float64 sum of chunk c of row r;  part[rows * SUMCH + r * SUMCH + c] = the same of (i - (len-1)/2) x[i]
// (the moment the least-squares line needs; centred so that slope and mean decouple)
// Row r starts at x + (r / grp) * gstride + (r % grp) * ldx when grp > 0 (frames of several clips: grp frames per
// clip, clip stride gstride), else at x + r * ldx.
__device__ __forceinline__ const float* row_base(const float* x, int64_t r, int64_t ldx, int64_t grp, int64_t gstride) {
  if (grp <= 0) return x + r * ldx;
  const int64_t g = r / grp;
  return x + g * gstride + (r - g * grp) * ldx;
}

__global__ void row_sum_kernel(const float* __restrict__ x, int64_t len, int64_t ldx, int64_t grp, int64_t gstride,
                               int linear, double* __restrict__ part, int64_t rows) {
  __shared__ double red[8];
  const int64_t r = blockIdx.y;
  const float* xr = row_base(x, r, ldx, grp, gstride);
  const int64_t per = (len + SUMCH - 1) / SUMCH;
  const int64_t lo = blockIdx.x * per, hi = lo + per < len ? lo + per : len;
  const double jc = 0.5 * (double)(len - 1);
  double s = 0.0, sj = 0.0;
  for (int64_t i = lo + threadIdx.x; i < hi; i += blockDim.x) {
    const double v = (double)xr[i];
    s += v;
    if (linear) sj += ((double)i - jc) * v;
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) { s += __shfl_xor(s, o, 64); sj += __shfl_xor(sj, o, 64); }
  if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6] = s; red[4 + (threadIdx.x >> 6)] = sj; }
  __syncthreads();
  if (threadIdx.x == 0) {
    part[r * SUMCH + blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
    part[(rows + r) * SUMCH + blockIdx.x] = (red[4] + red[5]) + (red[6] + red[7]);
  }
}

// out[r, i] = ((x[r, j] - trend_r(j)) * (win ? win[i] : 1)) for i < len, 0 for len <= i < n;  j = reverse ? len-1-i : i;
// trend: nothing, the row mean, or the least-squares line mean + slope (j - (len-1)/2) (scipy.signal.detrend).
// CPLX: out rows are complex (value, 0); otherwise real rows of n floats.
template <bool CPLX>
__global__ void pack_rows_kernel(const float* __restrict__ x, int64_t len, int64_t ldx, int64_t grp, int64_t gstride,
                                 const float* __restrict__ win, const double* __restrict__ part, int linear,
                                 int64_t rows, int reverse, float* __restrict__ out, int64_t n) {
  __shared__ double mean_s, slope_s;
  const int64_t r = blockIdx.y;
  double mean = 0.0, slope = 0.0;
  const double jc = 0.5 * (double)(len - 1);
  if (part) {
    if (threadIdx.x < 64) {
      double s = part[r * SUMCH + threadIdx.x], sj = linear ? part[(rows + r) * SUMCH + threadIdx.x] : 0.0;
#pragma unroll
      for (int o = 32; o >= 1; o >>= 1) { s += __shfl_xor(s, o, 64); sj += __shfl_xor(sj, o, 64); }
      if (threadIdx.x == 0) {
        const double nn = (double)len;
        mean_s = s / nn;
        slope_s = (linear && len > 1) ? sj / (nn * (nn * nn - 1.0) / 12.0) : 0.0;
      }
    }
    __syncthreads();
    mean = mean_s; slope = slope_s;
  }
  const float* xr = row_base(x, r, ldx, grp, gstride);
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    float v = 0.f;
    if (i < len) {
      const int64_t j = reverse ? len - 1 - i : i;
      const float s = xr[j];
      v = part ? (float)((double)s - mean - slope * ((double)j - jc)) : s;
      if (win) v *= win[i];
    }
    if (CPLX) reinterpret_cast<float2*>(out)[r * n + i] = make_float2(v, 0.f);
    else out[r * n + i] = v;
  }
}

__device__ __forceinline__ float2 cconj(float2 a) { return make_float2(a.x, -a.y); }
__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ float2 cmulf(float2 a, float2 b) {
  return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}

// spectrum bins k and H-k of the real sequence whose packed transform is Z:  X[k] = Xe + W Xo, X[H-k] = conj(Xe - W Xo)
// with Xe = (Z[k] + conj Z[H-k]) / 2, Xo = (Z[k] - conj Z[H-k]) / (2i), W = exp(-2 pi i k / (2H))
__device__ __forceinline__ void untangle(float2 zk, float2 zm, float2 w, float2& xk, float2& xm) {
  const float2 b = cconj(zm);
  const float2 xe = make_float2(0.5f * (zk.x + b.x), 0.5f * (zk.y + b.y));
  const float2 d = csub(zk, b);
  const float2 xo = make_float2(0.5f * d.y, -0.5f * d.x);     // d / (2i)
  const float2 t = cmulf(w, xo);
  xk = cadd(xe, t);
  xm = cconj(csub(xe, t));
}

// za [rows, H] and zb [rows_b (1 or rows), H] are the packed transforms of two real rows of length 2H; out [rows, H]
// receives the packed transform of their circular convolution (feed it to the inverse FFT of length H).
__global__ void rconv_spectrum_kernel(const float2* __restrict__ za, const float2* __restrict__ zb, int64_t rows_b,
                                      int64_t H, float2* __restrict__ out) {
  const int64_t r = blockIdx.y;
  const float2* a = za + r * H;
  const float2* b = zb + (rows_b == 1 ? 0 : r) * H;
  float2* o = out + r * H;
  for (int64_t k = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; k <= H / 2; k += (int64_t)gridDim.x * blockDim.x) {
    const int64_t m = k == 0 ? 0 : H - k;
    double sn, cs;
    sincospi(-(double)k / (double)H, &sn, &cs);
    const float2 w = make_float2((float)cs, (float)sn);
    float2 xk, xm, yk, ym;
    untangle(a[k], a[m], w, xk, xm);
    untangle(b[k], b[m], w, yk, ym);
    const float2 pk = cmulf(xk, yk), pm = cconj(cmulf(xm, ym));      // P[k], conj P[H-k]
    const float2 pe = make_float2(0.5f * (pk.x + pm.x), 0.5f * (pk.y + pm.y));
    const float2 d = make_float2(0.5f * (pk.x - pm.x), 0.5f * (pk.y - pm.y));
    const float2 po = cmulf(d, cconj(w));
    o[k] = make_float2(pe.x - po.y, pe.y + po.x);                    // Pe + i Po
    if (k != 0 && m != k) o[m] = make_float2(pe.x + po.y, -pe.y + po.x);   // conj(Pe) + i conj(Po)
  }
}

// scipy.signal.hilbert's one-sided mask, in place: h[0] = 1, h[n/2] = 1 (even n), h[1 .. ceil(n/2)-1] = 2, else 0
__global__ void analytic_mask_kernel(float2* __restrict__ X, int64_t n) {
  float2* xr = X + (int64_t)blockIdx.y * n;
  const int64_t half = (n + 1) / 2;                  // first bin that is not doubled (even n: n/2, odd: (n+1)/2)
  for (int64_t k = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; k < n; k += (int64_t)gridDim.x * blockDim.x) {
    float h = 0.f;
    if (k == 0 || ((n & 1) == 0 && k == n / 2)) h = 1.f;
    else if (k < half) h = 2.f;
    if (h != 1.f) { float2 v = xr[k]; xr[k] = make_float2(v.x * h, v.y * h); }
  }
}

// one-sided PSD of a full complex spectrum: out[k] = scale * |X[k]|^2 * (2 unless k is DC or, for even n, Nyquist)
__global__ void psd_onesided_kernel(const float2* __restrict__ X, int64_t n, float scale, float* __restrict__ out) {
  const int64_t F = n / 2 + 1;
  const float2* xr = X + (int64_t)blockIdx.y * n;
  float* orow = out + (int64_t)blockIdx.y * F;
  for (int64_t k = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; k < F; k += (int64_t)gridDim.x * blockDim.x) {
    const float2 v = xr[k];
    float p = (v.x * v.x + v.y * v.y) * scale;
    if (k != 0 && !((n & 1) == 0 && k == n / 2)) p *= 2.f;
    orow[k] = p;
  }
}

// acc[k] = (first ? 0 : acc[k]) + sum_r in[r, k] (float64, rows in order);  last: out[k] = acc[k] / divisor.
// Averages the per-segment periodograms of the generic Welch path in a fixed order, chunk after chunk.
__global__ void col_mean_kernel(const float* __restrict__ in, int64_t rows, int64_t F, double* __restrict__ acc,
                                int first, int last, double divisor, float* __restrict__ out) {
  const int64_t k = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (k >= F) return;
  double s = first ? 0.0 : acc[k];
  for (int64_t r = 0; r < rows; ++r) s += (double)in[r * F + k];
  acc[k] = s;
  if (last) out[k] = (float)(s / divisor);
}

unsigned grid_x(int64_t n, int64_t rows) {
  int64_t b = (n + 255) / 256;
  const int64_t cap = rows >= 64 ? 64 : 4096 / (rows < 1 ? 1 : rows);
  if (b > cap) b = cap;
  return (unsigned)(b < 1 ? 1 : b);
}

}  // namespace
}  // namespace syg

using namespace syg;

extern "C" int64_t syg_pack_rows_work_bytes(int64_t rows) { return 2 * rows * SUMCH * (int64_t)sizeof(double); }

extern "C" int syg_pack_frames_f32(const float* x, int64_t rows, int64_t len, int64_t ldx, int64_t rows_per_group,
                                   int64_t group_stride, const float* window, int detrend, int reverse, int cplx,
                                   float* out, int64_t n, void* work, void* stream) {
  SYG_REQUIRE(x && out, "pack_rows: null pointer argument");
  // ldx < len is allowed: overlapping rows (frames of one signal, row stride = hop)
  SYG_REQUIRE(rows >= 1 && rows <= 65535 && len >= 1 && n >= 1 && ldx >= 1, "pack_rows: bad sizes");
  SYG_REQUIRE(rows_per_group >= 0 && (rows_per_group == 0 || group_stride >= 1), "pack_rows: bad row grouping");
  SYG_REQUIRE(detrend >= 0 && detrend <= 2, "pack_rows: detrend must be 0 (none), 1 (constant) or 2 (linear)");
  SYG_REQUIRE(!detrend || work, "pack_rows: detrend needs the work buffer (syg_pack_rows_work_bytes)");
  if (len > n) len = n;
  double* part = nullptr;
  if (detrend) {
    part = (double*)work;
    hipLaunchKernelGGL(row_sum_kernel, dim3(SUMCH, (unsigned)rows), dim3(256), 0, (hipStream_t)stream, x, len, ldx,
                       rows_per_group, group_stride, detrend == 2, part, rows);
    SYG_CHECK_LAUNCH("pack_rows(sum)");
  }
  const dim3 grid(grid_x(n, rows), (unsigned)rows);
  if (cplx)
    hipLaunchKernelGGL(pack_rows_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, x, len, ldx, rows_per_group,
                       group_stride, window, part, detrend == 2, rows, reverse, out, n);
  else
    hipLaunchKernelGGL(pack_rows_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, x, len, ldx, rows_per_group,
                       group_stride, window, part, detrend == 2, rows, reverse, out, n);
  SYG_CHECK_LAUNCH("pack_rows");
  return SYG_OK;
}

extern "C" int syg_pack_rows_f32(const float* x, int64_t rows, int64_t len, int64_t ldx, const float* window,
                                 int detrend, int reverse, int cplx, float* out, int64_t n, void* work,
                                 void* stream) {
  return syg_pack_frames_f32(x, rows, len, ldx, 0, 0, window, detrend, reverse, cplx, out, n, work, stream);
}

extern "C" int syg_rconv_spectrum_c64(const float* za, const float* zb, int64_t rows, int64_t rows_b, int64_t H,
                                      float* out, void* stream) {
  SYG_REQUIRE(za && zb && out, "rconv_spectrum: null pointer argument");
  SYG_REQUIRE(rows >= 1 && rows <= 65535 && (rows_b == 1 || rows_b == rows), "rconv_spectrum: bad row counts");
  SYG_REQUIRE(H >= 2, "rconv_spectrum: H must be >= 2");
  hipLaunchKernelGGL(rconv_spectrum_kernel, dim3(grid_x(H / 2 + 1, rows), (unsigned)rows), dim3(256), 0,
                     (hipStream_t)stream, (const float2*)za, (const float2*)zb, rows_b, H, (float2*)out);
  SYG_CHECK_LAUNCH("rconv_spectrum");
  return SYG_OK;
}

extern "C" int syg_analytic_mask_c64(float* X, int64_t rows, int64_t n, void* stream) {
  SYG_REQUIRE(X, "analytic_mask: null pointer argument");
  SYG_REQUIRE(rows >= 1 && rows <= 65535 && n >= 1, "analytic_mask: bad sizes");
  hipLaunchKernelGGL(analytic_mask_kernel, dim3(grid_x(n, rows), (unsigned)rows), dim3(256), 0, (hipStream_t)stream,
                     (float2*)X, n);
  SYG_CHECK_LAUNCH("analytic_mask");
  return SYG_OK;
}

extern "C" int syg_psd_onesided_f32(const float* X, int64_t rows, int64_t n, double scale, float* out, void* stream) {
  SYG_REQUIRE(X && out, "psd_onesided: null pointer argument");
  SYG_REQUIRE(rows >= 1 && rows <= 65535 && n >= 1, "psd_onesided: bad sizes");
  hipLaunchKernelGGL(psd_onesided_kernel, dim3(grid_x(n / 2 + 1, rows), (unsigned)rows), dim3(256), 0,
                     (hipStream_t)stream, (const float2*)X, n, (float)scale, out);
  SYG_CHECK_LAUNCH("psd_onesided");
  return SYG_OK;
}

extern "C" int syg_col_mean_f32(const float* in, int64_t rows, int64_t F, double* acc, int first, int last,
                                double divisor, float* out, void* stream) {
  SYG_REQUIRE(in && acc, "col_mean: null pointer argument");
  SYG_REQUIRE(rows >= 1 && F >= 1, "col_mean: bad sizes");
  SYG_REQUIRE(!last || (out && divisor != 0.0), "col_mean: the last call needs out and a non-zero divisor");
  hipLaunchKernelGGL(col_mean_kernel, dim3((unsigned)((F + 63) / 64)), dim3(64), 0, (hipStream_t)stream, in, rows, F,
                     acc, first, last, divisor, out);
  SYG_CHECK_LAUNCH("col_mean");
  return SYG_OK;
}
