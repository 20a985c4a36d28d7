// Constant-Q transform building blocks (librosa.cqt as called by compute_cqt, sygnals/core/dsp.py:276-284):
//   decimate2_kernel   y[n] = scale * sum_j h[j] x[2n + half - j]   (the FIR decimation between octaves; taps are
//                      supplied by the caller -- see DESIGN.md for the resampler deviation)
//   cqt_octave_kernel  one octave: rectangular-window STFT frame (n_fft = 2^k) -> rfft in LDS -> n_filt complex dot
//                      products with the frequency-domain constant-Q basis -> out[b, row0 + f, t]
// One 64-lane wave per frame, FRAMES_PER_WG frames per workgroup (their outputs are adjacent in t).
#include "common.h"

namespace syg {
namespace {

constexpr int FPW = 8;            // frames per workgroup
constexpr int MAXFILT = 24;       // filters per octave (bins_per_octave <= 24 handled in registers)

__global__ void decimate2_kernel(const float* __restrict__ x, int64_t L, int64_t ldx, const float* __restrict__ taps,
                                 int ntaps, float scale, float* __restrict__ y, int64_t Lout, int64_t ldy) {
  const int64_t b = blockIdx.y;
  const float* xb = x + b * ldx;
  const int half = (ntaps - 1) / 2;
  for (int64_t n = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; n < Lout; n += (int64_t)gridDim.x * blockDim.x) {
    float acc = 0.f;
    const int64_t c = 2 * n + half;
    for (int j = 0; j < ntaps; ++j) {
      const int64_t i = c - j;
      if (i >= 0 && i < L) acc = fmaf(taps[j], xb[i], acc);
    }
    y[b * ldy + n] = acc * scale;
  }
}

__global__ __launch_bounds__(FPW * 64) void cqt_octave_kernel(
    const float* __restrict__ ysig, int64_t L, int64_t ldy, int n_fft, int hop, int64_t T,
    const float2* __restrict__ tw, const float2* __restrict__ basis, int n_filt, float2* __restrict__ out,
    int64_t out_bstride, int row0) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int M = n_fft >> 1, F = M + 1;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  float2* xa = reinterpret_cast<float2*>(lds) + (size_t)w * (2 * M + F);
  float2* xb = xa + M;
  float2* X = xb + M;                                  // [F] spectrum of this wave's frame
  const int64_t b = blockIdx.y;
  const int64_t t = (int64_t)blockIdx.x * FPW + w;
  const bool live = t < T;
  const float* yb = ysig + b * ldy;
  const int64_t s0 = t * (int64_t)hop - M;             // center=True, zero padding, rectangular window
  for (int m = lane; m < M; m += 64) {
    const int64_t s = s0 + 2 * m;
    const float a = (live && s >= 0 && s < L) ? yb[s] : 0.f;
    const float c = (live && s + 1 >= 0 && s + 1 < L) ? yb[s + 1] : 0.f;
    xa[m] = make_float2(a, c);
  }
  __syncthreads();
  float2* Z = block_fft(xa, xb, M, tw + n_fft, lane, 64);   // one wave per frame; all waves run the same trip counts
  for (int k = lane; k <= M; k += 64) {
    const float2 zk = Z[k & (M - 1)], zm = Z[(M - k) & (M - 1)];
    const float2 E = make_float2(zk.x + zm.x, zk.y - zm.y);
    const float2 O = make_float2(zk.y + zm.y, zm.x - zk.x);
    const float2 wO = cmul(tw[k], O);
    X[k] = make_float2(0.5f * (E.x + wO.x), 0.5f * (E.y + wO.y));
  }
  __syncthreads();
  // n_filt complex dot products, lanes stride over bins, then a wave reduction
  float2 acc[MAXFILT];
#pragma unroll
  for (int f = 0; f < MAXFILT; ++f) acc[f] = make_float2(0.f, 0.f);
  for (int k = lane; k < F; k += 64) {
    const float2 xk = X[k];
#pragma unroll
    for (int f = 0; f < MAXFILT; ++f)
      if (f < n_filt) {
        const float2 p = cmul(basis[f * F + k], xk);
        acc[f] = cadd(acc[f], p);
      }
  }
#pragma unroll
  for (int f = 0; f < MAXFILT; ++f)
    if (f < n_filt) {
      const float re = wave_sum(acc[f].x), im = wave_sum(acc[f].y);
      if (lane == 0 && live) out[b * out_bstride + (int64_t)(row0 + f) * T + t] = make_float2(re, im);
    }
}

bool is_pow2(int n) { return n >= 2 && (n & (n - 1)) == 0; }

}  // namespace
}  // namespace syg

using namespace syg;

extern "C" int syg_decimate2_f32(const float* x, int64_t B, int64_t L, int64_t ldx, const float* taps, int ntaps,
                                 float scale, float* y, int64_t ldy, void* stream) {
  SYG_REQUIRE(x && taps && y, "decimate2: null pointer argument");
  SYG_REQUIRE(B >= 1 && B <= 65535 && L >= 1 && ldx >= L, "decimate2: bad B/L/ldx");
  SYG_REQUIRE(ntaps >= 1 && (ntaps & 1) == 1 && ntaps <= 1025, "decimate2: ntaps must be odd and <= 1025");
  const int64_t Lout = (L + 1) / 2;
  SYG_REQUIRE(ldy >= Lout, "decimate2: ldy too small");
  int64_t blocks = (Lout + 255) / 256;
  if (blocks > 16384) blocks = 16384;
  hipLaunchKernelGGL(decimate2_kernel, dim3((unsigned)blocks, (unsigned)B), dim3(256), 0, (hipStream_t)stream, x, L, ldx,
                     taps, ntaps, scale, y, Lout, ldy);
  SYG_CHECK_LAUNCH("decimate2");
  return SYG_OK;
}

extern "C" int syg_cqt_octave_f32(const float* y, int64_t B, int64_t L, int64_t ldy, int n_fft, int hop, int64_t T,
                                  const float* twiddle, const float* basis, int n_filt, float* out,
                                  int64_t out_bstride, int row0, void* stream) {
  SYG_REQUIRE(y && twiddle && basis && out, "cqt_octave: null pointer argument");
  SYG_REQUIRE(is_pow2(n_fft) && n_fft >= 8 && n_fft <= 4096, "cqt_octave: n_fft must be a power of two in [8, 4096]");
  SYG_REQUIRE(B >= 1 && B <= 65535 && L >= 1 && ldy >= L && hop >= 1, "cqt_octave: bad B/L/ldy/hop");
  SYG_REQUIRE(T >= 1 && T <= 1 + L / hop, "cqt_octave: T=%lld exceeds the centred frame count %lld", (long long)T,
              (long long)(1 + L / hop));
  SYG_REQUIRE(n_filt >= 1 && n_filt <= MAXFILT, "cqt_octave: n_filt must be in [1, %d]", MAXFILT);
  SYG_REQUIRE(row0 >= 0 && out_bstride >= (int64_t)(row0 + n_filt) * T, "cqt_octave: output rows out of range");
  const int M = n_fft / 2;
  const size_t lds = (size_t)FPW * (2 * M + M + 1) * sizeof(float2);
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute((const void*)cqt_octave_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)lds);
    if (e != hipSuccess) { set_error("cqt_octave: cannot reserve LDS: %s", hipGetErrorString(e)); return SYG_E_LAUNCH; }
  }
  const int64_t gx = (T + FPW - 1) / FPW;
  SYG_REQUIRE(gx < (int64_t)0x7fffffff, "cqt_octave: grid too large");
  hipLaunchKernelGGL(cqt_octave_kernel, dim3((unsigned)gx, (unsigned)B), dim3(FPW * 64), lds, (hipStream_t)stream, y, L,
                     ldy, n_fft, hop, T, (const float2*)twiddle, (const float2*)basis, n_filt, (float2*)out, out_bstride,
                     row0);
  SYG_CHECK_LAUNCH("cqt_octave");
  return SYG_OK;
}
