// Fused STFT(4096) -> |X|^2 -> mel for frame_length 4096: librosa.stft + np.abs(.)**2 + melspectrogram as the feature
// manager calls them (sygnals/core/features/manager.py:184-187, 198, 219-222) at `frame_length=4096`, one WAVE per frame.
//
// The transform is welch_wave.hip's: a 4096-sample real frame is the 2048-point complex sequence z[m] = x[2m] + i x[2m+1],
// built from two 1024-point wave FFTs (wave_fft.h) of the even and odd elements of z, the radix-2 combine and the
// real-input split on mirror pairs, all in registers.  The 2049 powers go to the wave's own LDS row (the exchange scratch
// of the transforms aliases it) and the same wave projects the row onto the mel bands by segment sums
// (mel_segments.h, four passes of 64 lanes; table: sygnals_amd._tables.pack_mel_segments(..., n_pass=4)): no weight
// matrix, no spectrogram in HBM, no workgroup barrier behind the table set-up -- the waves of a workgroup only share
// constant tables.  The band values leave from the lanes that hold them (4-byte stores; the 16 waves of a CU work on
// consecutive frames, so the stores of a band meet in L2).
// ROWS (syg_stft_rows_w4096_f32): the same launch also hands the finished row to the per-frame row functions of
// row_features.h (spectral statistics, contrast tail means: 2049 bins, two 16-bin blocks per lane), behind the projection
// (which is skipped without a piece table); the results leave from the lanes that hold them.
#include <string.h>
#include "wave_fft.h"

namespace syg {
namespace {
#include "mel_segments.h"
#include "row_features.h"

constexpr int W4_WAVES = 4;                         // waves per workgroup (186-245 registers: two waves per SIMD)
constexpr int W4_N = 4096, W4_BINS = 2049;
constexpr int W4_ROW = 2200;                        // row_pos(2048) = 2176, + the 17-word window of the last piece, 8-aligned
constexpr int W4_SEG_WORDS = 4 * 2 * 64 * 4;        // piece table, four passes

struct W4Lds {
  static constexpr int O_ROW = 0;
  static constexpr int O_TW2 = O_ROW + W4_WAVES * W4_ROW;
  static constexpr int O_TW1 = O_TW2 + wfft::TW2_COMPLEX * 2;
  static constexpr int O_T2048 = O_TW1 + wfft::TW1_COMPLEX * 2;
  static constexpr int O_T4096 = O_T2048 + 8 * 64 * 2;
  static constexpr int O_SEG = O_T4096 + 8 * 64 * 2;
  static constexpr int O_CPL = O_SEG + W4_SEG_WORDS;       // ROWS: the contrast plan (lo / hi / k per band)
  static constexpr int TOTAL = O_CPL + 3 * SYG_MAX_BANDS;
  static_assert(W4_ROW >= 2 * wfft::SC_COMPLEX && W4_ROW % 4 == 0 && O_SEG % 4 == 0, "scratch inside the row, aligned tables");
};

__device__ __forceinline__ int w4_pos(int k) { return k + (k >> 4); }      // == _tables.row_pos

// E = zk + conj(zm), O = -i (zk - conj(zm));  X[k] = (E + w O) / 2,  X[N/2 - k] = conj(E - w O) / 2; returns 4 |X|^2
__device__ __forceinline__ void w4_split_pow(float2 zk, float2 zm, float2 w, float& pk, float& pm) {
  const float2 E = make_float2(zk.x + zm.x, zk.y - zm.y);
  const float2 O = make_float2(zk.y + zm.y, zm.x - zk.x);
  const float2 wO = cmul(w, O);
  const float ax = E.x + wO.x, ay = E.y + wO.y, bx = E.x - wO.x, by = E.y - wO.y;
  pk = fmaf(ax, ax, ay * ay);
  pm = fmaf(bx, bx, by * by);
}

struct W4Rows {                                       // arguments of the row functions (ROWS kernel)
  float binhz, roll_percent, bw_p;
  int smask;
  float* stats_out;                                   // [B, SYG_NSTAT, T] or null
  float* contrast_out;                                // [B, 2, n_rows, T] or null
  int n_rows, ascending;
  int lo[SYG_MAX_BANDS], hi[SYG_MAX_BANDS], k[SYG_MAX_BANDS];
};

template <bool ROWS>
__global__ __launch_bounds__(W4_WAVES * 64) void stft_mel_w4096_kernel(
    const float* __restrict__ y, int64_t L, int64_t ldy, int hop, int pad, int64_t T, int64_t n_frames,
    const float* __restrict__ win, const float2* __restrict__ tw4096, const float* __restrict__ segtab, int n_mels,
    float* __restrict__ mel_out, int aligned, W4Rows rw) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  float* prow = lds + W4Lds::O_ROW + w * W4_ROW;
  float2* sc = reinterpret_cast<float2*>(prow);
  float2* tw2l = reinterpret_cast<float2*>(lds + W4Lds::O_TW2);
  float2* tw1l = reinterpret_cast<float2*>(lds + W4Lds::O_TW1);
  float2* t2048 = reinterpret_cast<float2*>(lds + W4Lds::O_T2048);      // [8][64]: W_2048^k of (lane, unit j, pair d)
  float2* t4096 = reinterpret_cast<float2*>(lds + W4Lds::O_T4096);      // W_4096^k
  int* segl = reinterpret_cast<int*>(lds + W4Lds::O_SEG);
  wfft::Lane lc;
  wfft::init_lane(lc, lane);
  wfft::init_tables(tw2l, tw1l, tw4096, 4096, tid, W4_WAVES * 64);
  for (int i = tid; i < 8 * 64; i += W4_WAVES * 64) {
    const int q = i >> 6, l = i & 63;
    const int k = wfft::bin_of(l, q >> 2, q & 3);
    t2048[i] = tw4096[2 * k];
    t4096[i] = tw4096[k];
  }
  int* cpl = reinterpret_cast<int*>(lds + W4Lds::O_CPL);
  const bool project = !ROWS || segtab != nullptr;
  if (project)
    for (int i = tid; i < W4_SEG_WORDS; i += W4_WAVES * 64) segl[i] = reinterpret_cast<const int*>(segtab)[i];
  if (ROWS && tid < SYG_MAX_BANDS) {
    int lo = 0, hi = 0, k = 0;
#pragma unroll
    for (int r = 0; r < SYG_MAX_BANDS; ++r)
      if (tid == r) { lo = rw.lo[r]; hi = rw.hi[r]; k = rw.k[r]; }
    cpl[tid] = lo; cpl[SYG_MAX_BANDS + tid] = hi; cpl[2 * SYG_MAX_BANDS + tid] = k;
  }
  __syncthreads();
  unsigned lk = 0;
  if (project) {
#pragma unroll
    for (int p = 0; p < 4; ++p) lk |= (unsigned)(segl[4 * (128 * p + lane) + 2] | segl[4 * (128 * p + lane) + 3]);
  }
  const bool scan8 = __builtin_amdgcn_ballot_w64((lk >> 24) != 0) != 0;

  // consecutive frames go to consecutive waves (of this and of the neighbouring workgroups): overlapping samples meet in L2
  for (int64_t f = (int64_t)blockIdx.x * W4_WAVES + w; f < n_frames; f += (int64_t)gridDim.x * W4_WAVES) {
    const int64_t b = f / T, t = f - b * T;
    const float* yb = y + b * ldy;
    const int64_t s0 = t * (int64_t)hop - pad;
    int lf = lane;
    asm volatile("" : "+v"(lf));                    // (per-lane addresses are recomputed per frame, not hoisted)
    float2 ve[16], vo[16];
    {
      const float4* w4 = reinterpret_cast<const float4*>(win);
      if (aligned && s0 >= 0 && s0 + W4_N <= L) {
        const float4* sp = reinterpret_cast<const float4*>(yb + s0);
#pragma unroll
        for (int a = 0; a < 16; ++a) {
          const float4 r = sp[64 * a + lf], ww = w4[64 * a + lf];
          ve[a] = make_float2(r.x * ww.x, r.y * ww.y);
          vo[a] = make_float2(r.z * ww.z, r.w * ww.w);
        }
      } else {
        // frames over the clip's ends (the zero padding of center=True) and unaligned shapes: element loads
#pragma unroll
        for (int a = 0; a < 16; ++a) {
          const int64_t s = s0 + 4 * (64 * a + lf);
          const float4 ww = w4[64 * a + lf];
          float r[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) r[i] = (s + i >= 0 && s + i < L) ? yb[s + i] : 0.f;
          ve[a] = make_float2(r[0] * ww.x, r[1] * ww.y);
          vo[a] = make_float2(r[2] * ww.z, r[3] * ww.w);
        }
      }
    }
    float2 ek[2][4], em[2][4], ok[2][4], om[2][4], e512, o512;
    wfft::cfft1024(ve, lc, sc, tw1l, tw2l, lane, ek, em, e512);
    wfft::cfft1024(vo, lc, sc, tw1l, tw2l, lane, ok, om, o512);
    wave_lds_sync();                                // the scratch is dead: the row may be written
    // radix-2 combine (Z[k] = E[k] + W_2048^k O[k], Z[k + 1024] = E[k] - W_2048^k O[k]), the real-input split, powers
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        const int q = 4 * j + d;
        const float2 w2 = t2048[q * 64 + lf], w4 = t4096[q * 64 + lf];
        const float2 wok = cmul(w2, ok[j][d]);
        const float2 A = cadd(ek[j][d], wok), Bv = csub(ek[j][d], wok);            // Z[k], Z[1024 + k]
        const float2 wom = cmulc(om[j][d], w2);                                      // conj(W^k) O[1024 - k]
        const float2 C = csub(em[j][d], wom), D = cadd(em[j][d], wom);             // Z[1024 - k], Z[2048 - k]
        float p0, p1, p2, p3;
        w4_split_pow(A, D, w4, p0, p1);                                              // bins k, 2048 - k
        w4_split_pow(C, Bv, make_float2(-w4.y, -w4.x), p2, p3);                      // bins 1024 - k, 1024 + k
        const int k = wfft::bin_of(lf, j, d);
        prow[w4_pos(k)] = p0;
        prow[w4_pos(2048 - k)] = p1;
        prow[w4_pos(1024 - k)] = p2;
        if (k != 0) prow[w4_pos(1024 + k)] = p3;                                     // (k = 0: bin 1024 once)
      }
    if (lf == 0) {   // bin 512 of the two transforms: Z[512] = E - i O, Z[1536] = E + i O, split twiddle W_4096^512
      constexpr float R = 0.70710678118654752440f;
      const float2 A = make_float2(e512.x + o512.y, e512.y - o512.x), D = make_float2(e512.x - o512.y, e512.y + o512.x);
      float p512, p1536;
      w4_split_pow(A, D, make_float2(R, -R), p512, p1536);
      prow[w4_pos(512)] = p512;
      prow[w4_pos(1536)] = p1536;
    }
    wave_lds_sync();
    // ---- the row holds 4 |X|^2: project it (this wave alone), take the factor back at the store (exact)
    if (project) {
      float* mo = mel_out + (b * n_mels) * T + t;
      tri_project<4>(prow, reinterpret_cast<const float4*>(segl), lf, scan8,
                     [&](int band, float v) { mo[(int64_t)band * T] = 0.25f * v; });
    }
    if (ROWS) {
      // statistics / contrast of the row (4 |X|^2: PS = 1), one out-of-line call; the values come back in lanes 0 .. 15
      wave_lds_sync();                              // (the projection has read the row: a wide contrast band may park in it)
      lds_row pr = (lds_row)prow;
      float3 fr = make_float3(0.f, 0.f, 0.f);
      if (rw.stats_out != nullptr && rw.contrast_out != nullptr) {
        fr = row_features<W4_BINS, 1>(pr, lf, rw.binhz, rw.roll_percent, rw.bw_p, rw.smask, (lds_iptr)cpl, rw.n_rows, rw.ascending);
      } else if (rw.stats_out != nullptr) {
        fr.x = row_stats<W4_BINS, 1>(pr, lf, rw.binhz, rw.roll_percent, rw.bw_p, rw.smask);
      } else {
        const float2 pv = row_contrast_all<1, true>(pr, lf, (lds_iptr)cpl, rw.n_rows, rw.ascending);
        fr.y = pv.x; fr.z = pv.y;
      }
      int lq = lane;
      asm volatile("" : "+v"(lq));
      if (rw.contrast_out != nullptr && lq < rw.n_rows) {
        rw.contrast_out[((b * 2 + 0) * rw.n_rows + lq) * T + t] = fr.y;
        rw.contrast_out[((b * 2 + 1) * rw.n_rows + lq) * T + t] = fr.z;
      }
      if (rw.stats_out != nullptr && lq < SYG_NSTAT && ((stats_row_mask(rw.smask) >> lq) & 1))
        rw.stats_out[(b * SYG_NSTAT + lq) * T + t] = fr.x;
    }
    wave_lds_sync();                                // the row is read: the next frame's transforms may use the scratch
  }
}

}  // namespace
}  // namespace syg

using namespace syg;

static int w4096_launch(const char* who, const float* y, int64_t B, int64_t L, int64_t ldy, int hop, int center, int64_t T,
                        const float* window, const float* twiddle, const float* segtab, int n_segtab, int n_mels,
                        float* mel_out, const W4Rows* rows, hipStream_t st) {
  SYG_REQUIRE(y && window && twiddle, "%s: null pointer argument", who);
  SYG_REQUIRE(B >= 1 && L >= 1 && ldy >= L, "%s: need B >= 1, L >= 1, ldy >= L", who);
  SYG_REQUIRE(hop >= 1, "%s: hop must be >= 1", who);
  const int64_t Texp = center ? 1 + L / hop : (L >= W4_N ? 1 + (L - W4_N) / hop : 0);
  SYG_REQUIRE(T >= 1 && T == Texp, "%s: T=%lld does not match the framing rule (%lld)", who, (long long)T, (long long)Texp);
  if (segtab) {
    SYG_REQUIRE(mel_out, "%s: a piece table without mel_out", who);
    SYG_REQUIRE(n_segtab == W4_SEG_WORDS, "%s: the piece table has %d words, this library reads %d "
                "(sygnals_amd._tables.pack_mel_segments(..., n_pass=4))", who, n_segtab, W4_SEG_WORDS);
    SYG_REQUIRE(((uintptr_t)segtab) % 16 == 0, "%s: tables must be 16-byte aligned", who);
    SYG_REQUIRE(n_mels >= 1 && n_mels <= 255, "%s: n_mels must be in [1, 255]", who);
  }
  SYG_REQUIRE(((uintptr_t)window) % 16 == 0, "%s: tables must be 16-byte aligned", who);
  SYG_REQUIRE(B * T < ((int64_t)1 << 40), "%s: too many frames", who);
  const int pad = center ? W4_N / 2 : 0;
  const int aligned = (hop % 4 == 0) && (ldy % 4 == 0) && (((uintptr_t)y) % 16 == 0);
  const int64_t n_frames = B * T;
  int dev = 0, cus = 256;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cus = 256;
  const size_t lds = (size_t)W4Lds::TOTAL * sizeof(float);
  int64_t wgs = (n_frames + W4_WAVES - 1) / W4_WAVES;
  const int64_t cap = (int64_t)cus * 2 * 4;          // two workgroups per CU resident; a few rounds each
  if (wgs > cap) wgs = cap;
  const void* fn = rows ? (const void*)stft_mel_w4096_kernel<true> : (const void*)stft_mel_w4096_kernel<false>;
  hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) {
    set_error("%s: cannot reserve %zu B LDS: %s", who, lds, hipGetErrorString(e));
    return SYG_E_LAUNCH;
  }
  W4Rows rw;
  if (rows) rw = *rows; else memset(&rw, 0, sizeof(rw));
  if (rows)
    hipLaunchKernelGGL(stft_mel_w4096_kernel<true>, dim3((unsigned)wgs), dim3(W4_WAVES * 64), lds, st, y, L, ldy, hop, pad, T,
                       n_frames, window, (const float2*)twiddle, segtab, n_mels, mel_out, aligned, rw);
  else
    hipLaunchKernelGGL(stft_mel_w4096_kernel<false>, dim3((unsigned)wgs), dim3(W4_WAVES * 64), lds, st, y, L, ldy, hop, pad, T,
                       n_frames, window, (const float2*)twiddle, segtab, n_mels, mel_out, aligned, rw);
  SYG_CHECK_LAUNCH(who);
  return SYG_OK;
}

// y [B, L] (row stride ldy) -> mel_out [B, n_mels, T], power 2, for triangular filterbanks with a four-pass piece table
// (segtab: 2048 words on the device, 16-byte aligned).  window [4096]; twiddle: W_4096^k, k = 0 .. 4095.
extern "C" int syg_stft_mel_w4096_f32(const float* y, int64_t B, int64_t L, int64_t ldy, int hop, int center, int64_t T,
                                      const float* window, const float* twiddle, const float* segtab, int n_segtab,
                                      int n_mels, float* mel_out, void* stream) {
  SYG_REQUIRE(segtab && mel_out, "stft_mel_w4096: null pointer argument");
  return w4096_launch("stft_mel_w4096", y, B, L, ldy, hop, center, T, window, twiddle, segtab, n_segtab, n_mels, mel_out, nullptr,
                      (hipStream_t)stream);
}

// The per-frame statistics / contrast rows of syg_stft2048_mel_f32 for frame length 4096 (bins 0 .. 2048, bin frequency
// k sr / 4096) from the same launch: stats_out [B, SYG_NSTAT, T] (rows selected by stats_mask) and / or cplan_host +
// contrast_out [B, 2, n_rows, T] -- at least one; with segtab (and mel_out) also the mel power block, else nothing is projected.
extern "C" int syg_stft_rows_w4096_f32(const float* y, int64_t B, int64_t L, int64_t ldy, int hop, int center, int64_t T,
                                       const float* window, const float* twiddle, const float* segtab, int n_segtab, int n_mels,
                                       float* mel_out, float sr, float roll_percent, float bw_p, int stats_mask,
                                       float* stats_out, const int32_t* cplan_host, float* contrast_out, void* stream) {
  SYG_REQUIRE(stats_out || contrast_out, "stft_rows_w4096: no statistics requested");
  SYG_REQUIRE(T < ((int64_t)1 << 27), "stft_rows_w4096: too many frames per clip");
  W4Rows rw;
  memset(&rw, 0, sizeof(rw));
  if (stats_out) SYG_REQUIRE(sr > 0.f && roll_percent >= 0.f && roll_percent <= 1.f && bw_p > 0.f && (stats_mask & 31) != 0 &&
                                 stats_mask > 0 && stats_mask < 64, "stft_rows_w4096: invalid statistics parameters");
  if (contrast_out) {
    SYG_REQUIRE(cplan_host, "stft_rows_w4096: contrast_out given without cplan_host");
    rw.n_rows = cplan_host[0];
    SYG_REQUIRE(rw.n_rows >= 1 && rw.n_rows <= SYG_MAX_BANDS, "stft_rows_w4096: contrast rows must be in [1, %d]", SYG_MAX_BANDS);
    for (int r = 0; r < rw.n_rows; ++r) {
      rw.lo[r] = cplan_host[1 + r];
      rw.hi[r] = cplan_host[1 + SYG_MAX_BANDS + r];
      rw.k[r] = cplan_host[1 + 2 * SYG_MAX_BANDS + r];
      SYG_REQUIRE(rw.lo[r] >= 0 && rw.hi[r] <= W4_BINS && rw.lo[r] < rw.hi[r] && rw.k[r] >= 1 && rw.k[r] <= rw.hi[r] - rw.lo[r],
                  "stft_rows_w4096: contrast band %d invalid (lo=%d hi=%d k=%d)", r, rw.lo[r], rw.hi[r], rw.k[r]);
    }
    rw.ascending = 1;
    for (int r = 1; r < rw.n_rows; ++r)
      if (rw.lo[r] < rw.hi[r - 1] - 1 || rw.hi[r] < rw.hi[r - 1]) rw.ascending = 0;
  }
  rw.binhz = sr / (float)W4_N; rw.roll_percent = roll_percent; rw.bw_p = bw_p; rw.smask = stats_mask;
  rw.stats_out = stats_out; rw.contrast_out = contrast_out;
  return w4096_launch("stft_rows_w4096", y, B, L, ldy, hop, center, T, window, twiddle, segtab, n_segtab, n_mels, mel_out, &rw,
                      (hipStream_t)stream);
}
