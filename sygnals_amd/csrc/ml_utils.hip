// Feature formatting for ML on the device: column statistics and affine scaling behind
// sygnals/core/ml_utils/scaling.py:49-175 (scikit-learn's StandardScaler / MinMaxScaler / RobustScaler fit and
// transform), image resizing + [0, 1] normalisation behind core/ml_utils/formatters.py:256-334
// (scipy.ndimage.zoom, order 0 / 1, mode='nearest').  Small matrices ([frames, features]); everything is a
// column-wise reduction or an element-wise map, NaN-aware like scikit-learn's fit (NaNs are ignored in the statistics
// and pass through the transform).
#include "common.h"

namespace syg {
namespace {

constexpr int CS_ROWS = 4;          // row lanes per workgroup (64 columns x 4 rows)

// out[0..4][c] = count of non-NaN, mean, population variance, min, max of column c of x [n, F] (float64 results)
__global__ __launch_bounds__(64 * CS_ROWS) void col_stats_kernel(const float* __restrict__ x, int64_t n, int64_t F,
                                                                 double* __restrict__ out) {
  __shared__ double red[4][CS_ROWS][64];
  const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6;
  const int64_t c = (int64_t)blockIdx.x * 64 + cx;
  double cnt = 0.0, sum = 0.0, mn = 1.79e308, mx = -1.79e308;
  if (c < F)
    for (int64_t r = ry; r < n; r += CS_ROWS) {
      const double v = (double)x[r * F + c];
      if (v == v) { cnt += 1.0; sum += v; mn = fmin(mn, v); mx = fmax(mx, v); }
    }
  red[0][ry][cx] = cnt; red[1][ry][cx] = sum; red[2][ry][cx] = mn; red[3][ry][cx] = mx;
  __syncthreads();
  cnt = 0.0; sum = 0.0; mn = 1.79e308; mx = -1.79e308;
#pragma unroll
  for (int i = 0; i < CS_ROWS; ++i) {
    cnt += red[0][i][cx]; sum += red[1][i][cx]; mn = fmin(mn, red[2][i][cx]); mx = fmax(mx, red[3][i][cx]);
  }
  const double mean = cnt > 0.0 ? sum / cnt : 0.0;
  __syncthreads();
  double ss = 0.0;
  if (c < F)
    for (int64_t r = ry; r < n; r += CS_ROWS) {
      const double v = (double)x[r * F + c];
      if (v == v) { const double d = v - mean; ss += d * d; }
    }
  red[0][ry][cx] = ss;
  __syncthreads();
  if (ry != 0 || c >= F) return;
  ss = 0.0;
#pragma unroll
  for (int i = 0; i < CS_ROWS; ++i) ss += red[0][i][cx];
  const double nanv = __longlong_as_double(0x7ff8000000000000LL);
  out[0 * F + c] = cnt;
  out[1 * F + c] = cnt > 0.0 ? mean : nanv;
  out[2 * F + c] = cnt > 0.0 ? ss / cnt : nanv;
  out[3 * F + c] = cnt > 0.0 ? mn : nanv;
  out[4 * F + c] = cnt > 0.0 ? mx : nanv;
}

// out[r, c] = (x[r, c] - sub[c]) * mul[c] + add[c]   (float64 arithmetic, rounded once)
__global__ void affine_cols_kernel(const float* __restrict__ x, int64_t total, int64_t F, const double* __restrict__ sub,
                                   const double* __restrict__ mul, const double* __restrict__ add,
                                   float* __restrict__ out) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t c = i % F;
    out[i] = (float)(((double)x[i] - sub[c]) * mul[c] + add[c]);
  }
}

// np.nanpercentile(x, q, axis=0) (linear interpolation) for the nq fractions q in [0, 1]: one workgroup sorts one
// column in LDS (bitonic network on the next power of two, NaNs replaced by +inf and left out of the count)
__global__ __launch_bounds__(256) void col_quantiles_kernel(const float* __restrict__ x, int64_t n, int64_t F, int np2,
                                                            const double* __restrict__ q, int nq,
                                                            double* __restrict__ out) {
  extern __shared__ float v[];
  __shared__ int cnt_s;
  const int64_t c = blockIdx.x;
  const int tid = threadIdx.x;
  if (tid == 0) cnt_s = 0;
  __syncthreads();
  int mine = 0;
  for (int i = tid; i < np2; i += 256) {
    float f = __int_as_float(0x7f800000);
    if (i < n) {
      const float t = x[(int64_t)i * F + c];
      if (t == t) { f = t; ++mine; }
    }
    v[i] = f;
  }
  atomicAdd(&cnt_s, mine);
  __syncthreads();
  for (int k = 2; k <= np2; k <<= 1)
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = tid; i < np2; i += 256) {
        const int l = i ^ j;
        if (l > i) {
          const float a = v[i], b = v[l];
          const bool up = (i & k) == 0;
          if ((a > b) == up) { v[i] = b; v[l] = a; }
        }
      }
      __syncthreads();
    }
  const int m = cnt_s;
  if (tid < nq) {
    double r = __longlong_as_double(0x7ff8000000000000LL);
    if (m > 0) {
      const double pos = q[tid] * (double)(m - 1);
      int lo = (int)floor(pos);
      if (lo > m - 1) lo = m - 1;
      const int hi = lo + 1 < m ? lo + 1 : m - 1;
      const double t = pos - (double)lo, a = (double)v[lo], b = (double)v[hi];
      r = t < 0.5 ? a + (b - a) * t : b - (b - a) * (1.0 - t);          // numpy's _lerp
    }
    out[(int64_t)tid * F + c] = r;
  }
}

// scipy.ndimage.zoom(img, (H2 / H, W2 / W), order = 0 | 1, mode = 'nearest'): output pixel (i, j) samples the input at
// (i (H - 1) / (H2 - 1), j (W - 1) / (W2 - 1)); then optionally (v - lo) * inv (the [0, 1] normalisation)
__global__ void zoom_kernel(const float* __restrict__ img, int H, int W, int H2, int W2, int order,
                            float* __restrict__ out) {
  const double sy = H2 > 1 ? (double)(H - 1) / (double)(H2 - 1) : 0.0, sx = W2 > 1 ? (double)(W - 1) / (double)(W2 - 1) : 0.0;
  const int64_t total = (int64_t)H2 * W2;
  for (int64_t p = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; p < total; p += (int64_t)gridDim.x * blockDim.x) {
    const int i = (int)(p / W2), j = (int)(p - (int64_t)i * W2);
    const double y = i * sy, xx = j * sx;
    double r;
    if (order == 0) {
      int yi = (int)floor(y + 0.5), xi = (int)floor(xx + 0.5);
      yi = yi > H - 1 ? H - 1 : yi; xi = xi > W - 1 ? W - 1 : xi;
      r = (double)img[(int64_t)yi * W + xi];
    } else {
      int y0 = (int)floor(y), x0 = (int)floor(xx);
      y0 = y0 > H - 1 ? H - 1 : y0; x0 = x0 > W - 1 ? W - 1 : x0;
      const int y1 = y0 + 1 < H ? y0 + 1 : H - 1, x1 = x0 + 1 < W ? x0 + 1 : W - 1;
      const double fy = y - (double)y0, fx = xx - (double)x0;
      const double a = (double)img[(int64_t)y0 * W + x0], b = (double)img[(int64_t)y0 * W + x1];
      const double cc = (double)img[(int64_t)y1 * W + x0], d = (double)img[(int64_t)y1 * W + x1];
      r = (a * (1.0 - fx) + b * fx) * (1.0 - fy) + (cc * (1.0 - fx) + d * fx) * fy;
    }
    out[p] = (float)r;
  }
}

}  // namespace
}  // namespace syg

using namespace syg;

extern "C" int syg_col_stats_f32(const float* x, int64_t n, int64_t F, double* out, void* stream) {
  SYG_REQUIRE(x && out, "col_stats: null pointer argument");
  SYG_REQUIRE(n >= 1 && F >= 1 && (F + 63) / 64 < (int64_t)0x7fffffff, "col_stats: bad shape");
  hipLaunchKernelGGL(col_stats_kernel, dim3((unsigned)((F + 63) / 64)), dim3(64 * CS_ROWS), 0, (hipStream_t)stream, x, n,
                     F, out);
  SYG_CHECK_LAUNCH("col_stats");
  return SYG_OK;
}

extern "C" int syg_affine_cols_f32(const float* x, int64_t n, int64_t F, const double* sub, const double* mul,
                                   const double* add, float* out, void* stream) {
  SYG_REQUIRE(x && sub && mul && add && out, "affine_cols: null pointer argument");
  SYG_REQUIRE(n >= 1 && F >= 1, "affine_cols: bad shape");
  int64_t blocks = (n * F + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(affine_cols_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, n * F, F, sub,
                     mul, add, out);
  SYG_CHECK_LAUNCH("affine_cols");
  return SYG_OK;
}

extern "C" int syg_col_quantiles_f32(const float* x, int64_t n, int64_t F, const double* q, int nq, double* out,
                                     void* stream) {
  SYG_REQUIRE(x && q && out, "col_quantiles: null pointer argument");
  SYG_REQUIRE(n >= 1 && F >= 1 && F < (int64_t)0x7fffffff && nq >= 1 && nq <= 256, "col_quantiles: bad shape");
  SYG_REQUIRE(n <= 32768, "col_quantiles: at most 32768 rows (one column is sorted in LDS), got %lld", (long long)n);
  int np2 = 2;
  while (np2 < n) np2 <<= 1;
  const size_t lds = (size_t)np2 * sizeof(float);
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute((const void*)col_quantiles_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)lds);
    if (e != hipSuccess) { set_error("col_quantiles: cannot reserve %zu B of LDS", lds); return SYG_E_LAUNCH; }
  }
  hipLaunchKernelGGL(col_quantiles_kernel, dim3((unsigned)F), dim3(256), lds, (hipStream_t)stream, x, n, F, np2, q, nq,
                     out);
  SYG_CHECK_LAUNCH("col_quantiles");
  return SYG_OK;
}

extern "C" int syg_zoom_f32(const float* img, int H, int W, int H2, int W2, int order, float* out, void* stream) {
  SYG_REQUIRE(img && out, "zoom: null pointer argument");
  SYG_REQUIRE(H >= 1 && W >= 1 && H2 >= 1 && W2 >= 1, "zoom: bad shape");
  SYG_REQUIRE(order == 0 || order == 1, "zoom: order must be 0 (nearest) or 1 (linear), got %d", order);
  int64_t blocks = ((int64_t)H2 * W2 + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(zoom_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, img, H, W, H2, W2, order,
                     out);
  SYG_CHECK_LAUNCH("zoom");
  return SYG_OK;
}
