// Mixed-radix FFT in LDS for lengths N = 2^a 3^b 5^c 7^d <= 8192 -- scipy.fft.fft(x, n) as called by compute_fft /
// compute_ifft (sygnals/core/dsp.py:104, 151) takes ANY n and the default is n = len(data): one second of audio is
// 48000 = 2^7 3 5^3, 44100 = 2^2 3^2 5^2 7^2, 16000 = 2^7 5^3 samples, none a power of two.  Without this kernel such
// lengths go through Bluestein (three power-of-two transforms of >= 2 n points each); with it a 7-smooth length costs
// one transform of its own size, and longer 7-smooth lengths are composed four-step from two of them by the caller.
//
// Stockham autosort with one pass per radix (8 / 4 / 2 for the power-of-two part, then 3, 5, 7): pass with radix R at
// stride s (product of the radices done) reads a[j] = x[q + s (p + j m)], m = N / (s R), p = i / s, q = i mod s,
// transforms the R points in registers and writes y[q + s (R p + j)] = a[j] W_N^(j p s); input and output in natural
// order.  Same strided addressing and four-step twiddle as syg_fft_pow2_strided_c2c_f32.
#include "common.h"

namespace syg {
namespace {

constexpr int MAXPASS = 16;
constexpr int MIX_MAXN = 8192;

struct Radices {
  int n;
  int r[MAXPASS];
};

__device__ __forceinline__ void dft_small(float2 (&a)[2]) {
  const float2 t = a[1];
  a[1] = csub(a[0], t);
  a[0] = cadd(a[0], t);
}
__device__ __forceinline__ void dft_small(float2 (&a)[4]) { bfly4(a[0], a[1], a[2], a[3], a[0], a[1], a[2], a[3]); }
__device__ __forceinline__ void dft_small(float2 (&a)[8]) { dft8(a); }
__device__ __forceinline__ void dft_small(float2 (&a)[3]) {
  constexpr float S = 0.86602540378443864676f;               // sin(2 pi / 3)
  const float2 t = cadd(a[1], a[2]), d = csub(a[1], a[2]);
  const float2 m1 = make_float2(a[0].x - 0.5f * t.x, a[0].y - 0.5f * t.y);
  const float2 m2 = make_float2(S * d.y, -S * d.x);          // -i S d
  a[0] = cadd(a[0], t);
  a[1] = cadd(m1, m2);
  a[2] = csub(m1, m2);
}
__device__ __forceinline__ void dft_small(float2 (&a)[5]) {
  constexpr float C1 = 0.30901699437494742410f, C2 = -0.80901699437494742410f;   // cos(2 pi / 5), cos(4 pi / 5)
  constexpr float S1 = 0.95105651629515357212f, S2 = 0.58778525229247312917f;    // sin(2 pi / 5), sin(4 pi / 5)
  const float2 t1 = cadd(a[1], a[4]), t2 = cadd(a[2], a[3]), u1 = csub(a[1], a[4]), u2 = csub(a[2], a[3]);
  const float2 b1 = make_float2(a[0].x + C1 * t1.x + C2 * t2.x, a[0].y + C1 * t1.y + C2 * t2.y);
  const float2 b2 = make_float2(a[0].x + C2 * t1.x + C1 * t2.x, a[0].y + C2 * t1.y + C1 * t2.y);
  const float2 d1 = make_float2(S1 * u1.x + S2 * u2.x, S1 * u1.y + S2 * u2.y);
  const float2 d2 = make_float2(S2 * u1.x - S1 * u2.x, S2 * u1.y - S1 * u2.y);
  a[0] = cadd(a[0], cadd(t1, t2));
  a[1] = make_float2(b1.x + d1.y, b1.y - d1.x);              // b1 - i d1
  a[4] = make_float2(b1.x - d1.y, b1.y + d1.x);
  a[2] = make_float2(b2.x + d2.y, b2.y - d2.x);
  a[3] = make_float2(b2.x - d2.y, b2.y + d2.x);
}
__device__ __forceinline__ void dft_small(float2 (&a)[7]) {
  constexpr float C1 = 0.62348980185873353053f, C2 = -0.22252093395631440429f, C3 = -0.90096886790241912624f;
  constexpr float S1 = 0.78183148246802980871f, S2 = 0.97492791218182360702f, S3 = 0.43388373911755812048f;
  const float2 t1 = cadd(a[1], a[6]), t2 = cadd(a[2], a[5]), t3 = cadd(a[3], a[4]);
  const float2 u1 = csub(a[1], a[6]), u2 = csub(a[2], a[5]), u3 = csub(a[3], a[4]);
  // b_k = a0 + sum_j cos(2 pi j k / 7) t_j ;  d_k = sum_j sin(2 pi j k / 7) u_j   (k = 1, 2, 3)
  const float2 b1 = make_float2(a[0].x + C1 * t1.x + C2 * t2.x + C3 * t3.x, a[0].y + C1 * t1.y + C2 * t2.y + C3 * t3.y);
  const float2 b2 = make_float2(a[0].x + C2 * t1.x + C3 * t2.x + C1 * t3.x, a[0].y + C2 * t1.y + C3 * t2.y + C1 * t3.y);
  const float2 b3 = make_float2(a[0].x + C3 * t1.x + C1 * t2.x + C2 * t3.x, a[0].y + C3 * t1.y + C1 * t2.y + C2 * t3.y);
  const float2 d1 = make_float2(S1 * u1.x + S2 * u2.x + S3 * u3.x, S1 * u1.y + S2 * u2.y + S3 * u3.y);
  const float2 d2 = make_float2(S2 * u1.x - S3 * u2.x - S1 * u3.x, S2 * u1.y - S3 * u2.y - S1 * u3.y);
  const float2 d3 = make_float2(S3 * u1.x - S1 * u2.x + S2 * u3.x, S3 * u1.y - S1 * u2.y + S2 * u3.y);
  a[0] = cadd(cadd(a[0], t1), cadd(t2, t3));
  a[1] = make_float2(b1.x + d1.y, b1.y - d1.x); a[6] = make_float2(b1.x - d1.y, b1.y + d1.x);
  a[2] = make_float2(b2.x + d2.y, b2.y - d2.x); a[5] = make_float2(b2.x - d2.y, b2.y + d2.x);
  a[3] = make_float2(b3.x + d3.y, b3.y - d3.x); a[4] = make_float2(b3.x - d3.y, b3.y + d3.x);
}

template <int R>
__device__ __forceinline__ void mixed_pass(const float2* x, float2* y, int N, int s, const float2* __restrict__ tw,
                                           int tid, int nt) {
  const int m = N / (s * R);
  for (int i = tid; i < N / R; i += nt) {
    const int p = i / s, q = i - p * s;
    float2 a[R];
#pragma unroll
    for (int j = 0; j < R; ++j) a[j] = x[q + s * (p + j * m)];
    dft_small(a);
    const int e = p * s, ob = q + s * R * p;
    y[ob] = a[0];
#pragma unroll
    for (int j = 1; j < R; ++j) y[ob + j * s] = cmul(a[j], tw[j * e]);      // j e < N: no wrap
  }
}

__device__ __forceinline__ float2* block_fft_mixed(float2* x, float2* y, int N, const Radices& rd,
                                                   const float2* __restrict__ tw, int tid, int nt) {
  int s = 1;
  for (int ps = 0; ps < rd.n; ++ps) {
    const int R = rd.r[ps];
    switch (R) {
      case 8: mixed_pass<8>(x, y, N, s, tw, tid, nt); break;
      case 4: mixed_pass<4>(x, y, N, s, tw, tid, nt); break;
      case 2: mixed_pass<2>(x, y, N, s, tw, tid, nt); break;
      case 3: mixed_pass<3>(x, y, N, s, tw, tid, nt); break;
      case 5: mixed_pass<5>(x, y, N, s, tw, tid, nt); break;
      default: mixed_pass<7>(x, y, N, s, tw, tid, nt); break;
    }
    __syncthreads();
    float2* t = x; x = y; y = t;
    s *= R;
  }
  return x;
}

// element e of transform (o, b) at in[o*in_os + b*in_bs + e*in_es]; output k of transform b times W_bign^(b k) when
// bign > 0 (the four-step twiddle), times scale; inverse via conj(FFT(conj(x)))
__global__ void fft_mixed_strided_kernel(const float2* __restrict__ in, float2* __restrict__ out, int n, Radices rd,
                                         int inverse, const float2* __restrict__ tw, int64_t in_os, int64_t in_bs,
                                         int64_t in_es, int64_t out_os, int64_t out_bs, int64_t out_es, int64_t bign,
                                         float scale, int flags, int64_t mask_n, int64_t in_valid) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float2* x = reinterpret_cast<float2*>(lds);
  float2* y = x + n;
  const int tid = threadIdx.x, nt = blockDim.x;
  const int64_t b = blockIdx.x, o = blockIdx.y;
  const int64_t ibase = o * in_os, irow = b * in_bs, obase = o * out_os + b * out_bs;
  for (int i = tid; i < n; i += nt) {
    float2 v = fft_load(in, ibase, irow + (int64_t)i * in_es, flags, mask_n, in_valid);
    if (inverse) v.y = -v.y;
    x[i] = v;
  }
  __syncthreads();
  float2* r = block_fft_mixed(x, y, n, rd, tw, tid, nt);
  for (int k = tid; k < n; k += nt) {
    float2 v = r[k];
    if (bign > 0) {
      const int64_t e = (b * (int64_t)k) % bign;
      double sn, cs;
      sincospi(-2.0 * (double)e / (double)bign, &sn, &cs);
      v = cmul(v, make_float2((float)cs, (float)sn));
    }
    v.x *= scale; v.y *= scale;
    if (inverse) v.y = -v.y;
    fft_store(out, obase + (int64_t)k * out_es, v, flags);
  }
}

// Column-tiled form for the two passes of a four-step transform (see fft_cols_kernel in fft_generic.hip): the
// transforms are columns of a row-major matrix (in_bs == 1); a workgroup takes CB adjacent columns, so global
// accesses are runs of CB complex values, and runs the CB transforms side by side (256 / CB threads each).
constexpr int MCOLS_NT = 256, MCOLS_PAD = 2;

template <bool KFAST>
__global__ __launch_bounds__(MCOLS_NT) void fft_mixed_cols_kernel(const float2* __restrict__ in, float2* __restrict__ out,
                                                                  int n, Radices rd, int cb_log, int inverse,
                                                                  const float2* __restrict__ tw, int64_t in_os,
                                                                  int64_t in_es, int64_t out_os, int64_t out_bs,
                                                                  int64_t out_es, int64_t bign, float scale, int flags,
                                                                  int64_t mask_n, int64_t in_valid) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int CB = 1 << cb_log, LP = n + MCOLS_PAD;
  float2* x = reinterpret_cast<float2*>(lds);
  float2* y = x + CB * LP;
  const int tid = threadIdx.x;
  const int64_t c0 = (int64_t)blockIdx.x << cb_log, o = blockIdx.y;
  const int64_t ibase = o * in_os, obase = o * out_os;
  const int total = n << cb_log;
  for (int idx = tid; idx < total; idx += MCOLS_NT) {
    const int c = idx & (CB - 1), e = idx >> cb_log;
    const int64_t pos = (int64_t)e * in_es + c0 + c;           // position inside the row (= the bin, for the analytic weights)
    float2 v = fft_load(in, ibase, pos, flags, mask_n, in_valid);
    if (inverse) v.y = -v.y;
    x[c * LP + e] = v;
  }
  __syncthreads();
  const int tpc_log = 8 - cb_log;
  const int g = tid >> tpc_log, lt = tid & ((1 << tpc_log) - 1);
  const float2* r = block_fft_mixed(x + g * LP, y + g * LP, n, rd, tw, lt, 1 << tpc_log) - g * LP;
  for (int idx = tid; idx < total; idx += MCOLS_NT) {
    int c, k;
    if (KFAST) { c = idx / n; k = idx - c * n; }
    else { c = idx & (CB - 1); k = idx >> cb_log; }
    float2 v = r[c * LP + k];
    if (bign > 0) {
      int64_t e = (c0 + c) * (int64_t)k;
      if (e >= bign) e %= bign;                                // (column * k < bign in a four-step split: never taken there)
      double sn, cs;
      sincospi(-2.0 * (double)e / (double)bign, &sn, &cs);
      v = cmul(v, make_float2((float)cs, (float)sn));
    }
    v.x *= scale; v.y *= scale;
    if (inverse) v.y = -v.y;
    fft_store(out, obase + (c0 + c) * out_bs + (int64_t)k * out_es, v, flags);
  }
}

}  // namespace
}  // namespace syg

using namespace syg;

// 1 when n factors into 2, 3, 5, 7 only; fills the pass radices (8s first, then 4 / 2, then 3, 5, 7)
extern "C" int syg_fft_mixed_plan(int64_t n, int32_t* radices_host, int max_passes) {
  if (n < 2 || !radices_host || max_passes < 1) return 0;
  int cnt = 0;
  int64_t m = n;
  auto push = [&](int r) { if (cnt < max_passes) radices_host[cnt] = r; ++cnt; };
  while (m % 8 == 0) { push(8); m /= 8; }
  if (m % 4 == 0) { push(4); m /= 4; }
  if (m % 2 == 0) { push(2); m /= 2; }
  for (int r : {3, 5, 7})
    while (m % r == 0) { push(r); m /= r; }
  if (m != 1 || cnt > max_passes) return 0;
  return cnt;
}

extern "C" int syg_fft_mixed_strided_ex_f32(const float* in, float* out, int64_t outer, int64_t batch, int n,
                                             int inverse, const float* twiddle, int64_t in_os, int64_t in_bs,
                                             int64_t in_es, int64_t out_os, int64_t out_bs, int64_t out_es,
                                             int64_t bign, float scale, int flags, int64_t mask_n, int64_t in_valid, void* stream) {
  SYG_REQUIRE(in && out && twiddle, "fft_mixed: null pointer argument");
  SYG_REQUIRE(n >= 2 && n <= MIX_MAXN, "fft_mixed: n must be in [2, %d] (got %d)", MIX_MAXN, n);
  SYG_REQUIRE(batch >= 1 && batch < (int64_t)0x7fffffff && outer >= 1 && outer <= 65535, "fft_mixed: bad batch/outer");
  SYG_REQUIRE(in != out, "fft_mixed: in-place operation is not supported");
  SYG_REQUIRE(flags >= 0 && flags <= 7 && (flags & 5) != 5 && mask_n >= 0 && in_valid >= 0, "fft_mixed: bad flags / mask length");
  Radices rd;
  int32_t rr[MAXPASS];
  rd.n = syg_fft_mixed_plan(n, rr, MAXPASS);
  SYG_REQUIRE(rd.n >= 1, "fft_mixed: n = %d has a prime factor other than 2, 3, 5, 7", n);
  for (int i = 0; i < MAXPASS; ++i) rd.r[i] = i < rd.n ? rr[i] : 1;
  if (in_bs == 1 && (out_es == 1 || out_bs == 1) && n <= 1024 && n >= 8) {
    int cb_log = 4;                                            // 16 columns = 128-byte runs
    while (cb_log > 2 && (((int64_t)n << cb_log) > 4096 || batch % (1 << cb_log) != 0)) --cb_log;
    // two workgroups per CU hide too little: above 40 KB of LDS take 8 columns (64-byte runs) -- 758 -> 646 us for the two
    // passes of 1024 x 48000 (200 x 240), 1145 -> 760 us for 1024 x 65536 (256 x 256); 4 columns are slower again
    if (cb_log == 4 && (size_t)2 * ((size_t)(n + MCOLS_PAD) << 4) * sizeof(float2) > 40 * 1024) cb_log = 3;
    if (((int64_t)n << cb_log) <= 4096 && batch % (1 << cb_log) == 0) {
      const bool kfast = out_es == 1;
      const void* fn = kfast ? (const void*)fft_mixed_cols_kernel<true> : (const void*)fft_mixed_cols_kernel<false>;
      const size_t clds = (size_t)2 * ((size_t)(n + MCOLS_PAD) << cb_log) * sizeof(float2);
      if (clds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)clds);
        if (e != hipSuccess) { set_error("fft_mixed(cols): cannot reserve %zu B of LDS", clds); return SYG_E_LAUNCH; }
      }
      const dim3 grid((unsigned)(batch >> cb_log), (unsigned)outer);
      if (kfast)
        hipLaunchKernelGGL(fft_mixed_cols_kernel<true>, grid, dim3(MCOLS_NT), clds, (hipStream_t)stream,
                           (const float2*)in, (float2*)out, n, rd, cb_log, inverse, (const float2*)twiddle, in_os,
                           in_es, out_os, out_bs, out_es, bign, scale, flags, mask_n, in_valid);
      else
        hipLaunchKernelGGL(fft_mixed_cols_kernel<false>, grid, dim3(MCOLS_NT), clds, (hipStream_t)stream,
                           (const float2*)in, (float2*)out, n, rd, cb_log, inverse, (const float2*)twiddle, in_os,
                           in_es, out_os, out_bs, out_es, bign, scale, flags, mask_n, in_valid);
      SYG_CHECK_LAUNCH("fft_mixed(cols)");
      return SYG_OK;
    }
  }
  const size_t lds = (size_t)n * 2 * sizeof(float2);
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute((const void*)fft_mixed_strided_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)lds);
    if (e != hipSuccess) { set_error("fft_mixed: cannot reserve %zu B of LDS", lds); return SYG_E_LAUNCH; }
  }
  int nt = n / 8;
  nt = nt < 64 ? 64 : (nt > 1024 ? 1024 : ((nt + 63) / 64) * 64);
  hipLaunchKernelGGL(fft_mixed_strided_kernel, dim3((unsigned)batch, (unsigned)outer), dim3(nt), lds,
                     (hipStream_t)stream, (const float2*)in, (float2*)out, n, rd, inverse, (const float2*)twiddle,
                     in_os, in_bs, in_es, out_os, out_bs, out_es, bign, scale, flags, mask_n, in_valid);
  SYG_CHECK_LAUNCH("fft_mixed");
  return SYG_OK;
}

extern "C" int syg_fft_mixed_strided_c2c_f32(const float* in, float* out, int64_t outer, int64_t batch, int n,
                                             int inverse, const float* twiddle, int64_t in_os, int64_t in_bs,
                                             int64_t in_es, int64_t out_os, int64_t out_bs, int64_t out_es,
                                             int64_t bign, float scale, void* stream) {
  return syg_fft_mixed_strided_ex_f32(in, out, outer, batch, n, inverse, twiddle, in_os, in_bs, in_es, out_os, out_bs, out_es,
                                      bign, scale, 0, 0, 0, stream);
}
