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

// FIR decimation by 2:  y[n] = scale * sum_j h[j] x[2n + half - j],  half = (ntaps - 1) / 2.
// A workgroup produces DEC_NBO consecutive outputs: the input run it needs is staged once in LDS, split into its
// even and odd samples (E[m] = x[2m], O[m] = x[2m+1]: the stride-2 reads of the filter become contiguous,
// conflict-free LDS reads; samples outside [0, L) are staged as zeros), the taps sit in LDS too, and every thread
// accumulates DEC_OUTS outputs per tap read.  Taps are applied in index order (same sum order as a direct loop).
constexpr int DEC_NT = 256, DEC_OUTS = 4, DEC_NBO = DEC_NT * DEC_OUTS;

__global__ __launch_bounds__(DEC_NT) void decimate2_kernel(const float* __restrict__ x, int64_t L, int64_t ldx,
                                                           const float* __restrict__ taps, int ntaps, float scale,
                                                           float* __restrict__ y, int64_t Lout, int64_t ldy) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int half = (ntaps - 1) / 2;
  const int nstage = DEC_NBO + half + 2;             // staged pairs (even, odd)
  float* hs = lds;                                   // [ntaps]
  float* E = hs + ((ntaps + 3) & ~3);
  float* O = E + nstage;
  const int64_t b = blockIdx.y;
  const float* xb = x + b * ldx;
  const int tid = threadIdx.x;
  for (int j = tid; j < ntaps; j += DEC_NT) hs[j] = taps[j];
  for (int64_t n0 = (int64_t)blockIdx.x * DEC_NBO; n0 < Lout; n0 += (int64_t)gridDim.x * DEC_NBO) {
    // lowest input index used: 2 n0 + half - (ntaps - 1) = 2 n0 - half; mbase = floor(that / 2)
    const int64_t ilo = 2 * n0 - half;
    const int64_t mbase = (ilo >= 0) ? ilo / 2 : -((-ilo + 1) / 2);
    __syncthreads();                                  // the previous round's reads are done
    for (int u = tid; u < nstage; u += DEC_NT) {
      const int64_t i0 = 2 * (mbase + u);
      E[u] = (i0 >= 0 && i0 < L) ? xb[i0] : 0.f;
      O[u] = (i0 + 1 >= 0 && i0 + 1 < L) ? xb[i0 + 1] : 0.f;
    }
    __syncthreads();
    float acc[DEC_OUTS];
#pragma unroll
    for (int o = 0; o < DEC_OUTS; ++o) acc[o] = 0.f;
    // input index of tap j for output n: i = 2n + half - j = 2 (n + q) + r with (q, r) from c = half - j
    for (int j = 0; j < ntaps; ++j) {
      const int c = half - j;
      const int q = (c >= 0) ? c / 2 : -((-c + 1) / 2);     // floor(c / 2)
      const int r = c - 2 * q;                               // 0 or 1
      const float* src = (r ? O : E) + (int)(n0 - mbase) + q + tid;
      const float h = hs[j];
      if (h == 0.f) continue;        // half-band filters: every other tap beside the centre is zero (wave-uniform skip)
#pragma unroll
      for (int o = 0; o < DEC_OUTS; ++o) acc[o] = fmaf(h, src[o * DEC_NT], acc[o]);
    }
#pragma unroll
    for (int o = 0; o < DEC_OUTS; ++o) {
      const int64_t n = n0 + tid + o * DEC_NT;
      if (n < Lout) y[b * ldy + n] = acc[o] * scale;
    }
  }
}

// Non-zero column range of every filter's frequency-domain row: librosa sparsifies the basis (entries below
// the 1 % magnitude quantile are set to zero), so each constant-Q filter keeps a run of a few dozen bins.
struct FiltHull {
  int k0[MAXFILT];
  int len[MAXFILT];
};

__global__ __launch_bounds__(FPW * 64) void cqt_octave_kernel(
    const float* __restrict__ ysig, int64_t L, int64_t ldy, int n_fft, int hop, int64_t T,
    const float2* __restrict__ tw, const float2* __restrict__ basis, int n_filt, FiltHull hull,
    float2* __restrict__ out, int64_t out_bstride, int row0) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  __shared__ int hl[2 * MAXFILT];
  if (threadIdx.x < MAXFILT) {
    hl[threadIdx.x] = hull.k0[threadIdx.x];
    hl[MAXFILT + threadIdx.x] = hull.len[threadIdx.x];
  }
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
  float2* Z = block_fft<true>(xa, xb, M, tw + n_fft, lane, 64);   // one wave per frame, ordered by the wave itself
  for (int k = lane; k <= M; k += 64) {
    const float2 zk = Z[k & (M - 1)], zm = Z[(M - k) & (M - 1)];
    const float2 E = make_float2(zk.x + zm.x, zk.y - zm.y);
    const float2 O = make_float2(zk.y + zm.y, zm.x - zk.x);
    const float2 wO = cmul(tw[k], O);
    X[k] = make_float2(0.5f * (E.x + wO.x), 0.5f * (E.y + wO.y));
  }
  wave_lds_sync();
  // n_filt sparse complex dot products: each 16-lane row of the wave takes one filter at a time (filter f -> row
  // f & 3 of pass f >> 2) and walks only the filter's non-zero bin run; the row sum is four DPP steps.
  const int row = lane >> 4, l16 = lane & 15;
  for (int f0 = 0; f0 < n_filt; f0 += 4) {
    const int f = f0 + row;
    const int k0 = (f < n_filt) ? hl[f] : 0;
    const int len = (f < n_filt) ? hl[MAXFILT + f] : 0;
    int maxlen = 0;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int lr = __builtin_amdgcn_readlane(len, 16 * r);
      maxlen = lr > maxlen ? lr : maxlen;
    }
    float2 acc = make_float2(0.f, 0.f);
    const float2* bf = basis + (int64_t)(f < n_filt ? f : 0) * F + k0;
    for (int j = l16; j < maxlen; j += 16)
      if (j < len) acc = cadd(acc, cmul(bf[j], X[k0 + j]));
    float re = acc.x, im = acc.y;
    re += dpp_f<DPP_QP_1032>(re); im += dpp_f<DPP_QP_1032>(im);
    re += dpp_f<DPP_QP_2301>(re); im += dpp_f<DPP_QP_2301>(im);
    re += dpp_f<DPP_ROW_HALF_MIRROR>(re); im += dpp_f<DPP_ROW_HALF_MIRROR>(im);
    re += dpp_f<DPP_ROW_MIRROR>(re); im += dpp_f<DPP_ROW_MIRROR>(im);
    if (l16 == 0 && f < n_filt && live) out[b * out_bstride + (int64_t)(row0 + f) * T + t] = make_float2(re, im);
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
  int64_t blocks = (Lout + DEC_NBO - 1) / DEC_NBO;
  if (blocks > 16384) blocks = 16384;
  const int half = (ntaps - 1) / 2;
  const size_t lds = (size_t)(((ntaps + 3) & ~3) + 2 * (DEC_NBO + half + 2)) * sizeof(float);
  hipLaunchKernelGGL(decimate2_kernel, dim3((unsigned)blocks, (unsigned)B), dim3(DEC_NT), lds, (hipStream_t)stream, x, L,
                     ldx, taps, ntaps, scale, y, Lout, ldy);
  SYG_CHECK_LAUNCH("decimate2");
  return SYG_OK;
}

extern "C" int syg_cqt_octave_f32(const float* y, int64_t B, int64_t L, int64_t ldy, int n_fft, int hop, int64_t T,
                                  const float* twiddle, const float* basis, int n_filt, const int32_t* hull_host,
                                  float* out, int64_t out_bstride, int row0, void* stream) {
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
  // hull_host: [2 * n_filt] = first non-zero bin and run length of every basis row (NULL: rows are dense)
  FiltHull hull;
  for (int f = 0; f < MAXFILT; ++f) {
    hull.k0[f] = 0;
    hull.len[f] = (f < n_filt) ? M + 1 : 0;
    if (hull_host && f < n_filt) {
      hull.k0[f] = hull_host[f];
      hull.len[f] = hull_host[n_filt + f];
      SYG_REQUIRE(hull.k0[f] >= 0 && hull.len[f] >= 0 && hull.k0[f] + hull.len[f] <= M + 1,
                  "cqt_octave: non-zero run of filter %d out of range (k0=%d len=%d)", f, hull.k0[f], hull.len[f]);
    }
  }
  hipLaunchKernelGGL(cqt_octave_kernel, dim3((unsigned)gx, (unsigned)B), dim3(FPW * 64), lds, (hipStream_t)stream, y, L,
                     ldy, n_fft, hop, T, (const float2*)twiddle, (const float2*)basis, n_filt, hull, (float2*)out,
                     out_bstride, row0);
  SYG_CHECK_LAUNCH("cqt_octave");
  return SYG_OK;
}
