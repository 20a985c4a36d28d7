// Fused STFT(1024) -> |X|^2 -> mel for frame_length 1024 (the frame length of the reference's own tests and CLI:
// tests/test_features_manager.py:183-220, cli/features_cmd.py:35; librosa.stft + np.abs(.)**2 + melspectrogram as
// manager.py:184-187, 198, 219-222 call them), free-running waves: one wave owns TWO frames per 1024-point complex
// transform (z = a + i b: A[k] = (Z[k] + conj Z[1024 - k]) / 2, B[k] = -i (Z[k] - conj Z[1024 - k]) / 2, see
// stft_mel_pow2.hip) and projects its own two power rows onto the mel bands by segment sums (mel_segments.h: the
// four-pass table holds two passes per row).  No weight matrix, no spectrogram in HBM, and behind the table set-up no
// workgroup barrier: the dense-matrix form (stft_mel_w1024_kernel) spends two barriers and six of eight waves' matrix
// pipe per 16 frames on the projection.  The samples of the next pair of frames are requested before the current pair is
// transformed.  power = 2 only; other powers and filterbanks without a piece table stay on the matrix form.
// ROWS: the per-frame row functions of the 2048 kernel (row_features.h: spectral centroid / bandwidth / flatness / rolloff /
// dominant frequency, contrast tail means -- frequency_domain.py:24-386 as driven by manager.py:289-343) on the wave's own
// two 513-bin rows, between the rows' completion and the projection: extract_features(frame_length=1024, [spectral
// features]) -- the call of the reference's own manager tests (tests/test_features_manager.py:58-62, 167-174) -- from one
// launch, no spectrogram in HBM.  The filterbank is then optional (segtab == nullptr: statistics only).
#include "wave_fft.h"
#include <string.h>

namespace syg {
namespace {
#include "mel_segments.h"
#include "row_features.h"

constexpr int S1_WAVES = 8;
constexpr int S1_BASE = 4;                           // words in front of bin 0: room for the lead of a first piece of 1-3 bins
constexpr int S1_ROW = 568;                          // words per skewed row: base + row_pos(512) = 548, + the 17-word window, 8-aligned
constexpr int S1_SCW = 2 * S1_ROW + 16;              // per-wave scratch = its two rows (>= 528 complex for the transform)
constexpr int S1_SEG_WORDS = 4 * 2 * 64 * 4;
static_assert(S1_SCW >= 2 * wfft::SC_COMPLEX, "the exchange scratch must fit inside the two rows");

__device__ __forceinline__ int s1_pos(int k) { return S1_BASE + k + (k >> 4); }      // == _tables.row_pos + row_base

#ifndef SYG_S1_WAVES_PER_SIMD
#define SYG_S1_WAVES_PER_SIMD 4
#endif
struct S1Rows {                                       // arguments of the row functions (ROWS kernels)
  float binhz, roll_percent, bw_p;
  int smask;
  float* stats_out;                                   // [B, SYG_NSTAT, T] or null
  float* contrast_out;                                // [B, 2, n_rows, T] or null
  int n_rows, ascending;
  int lo[SYG_MAX_BANDS], hi[SYG_MAX_BANDS], k[SYG_MAX_BANDS];
};

template <bool ROWS>
__global__ __launch_bounds__(S1_WAVES * 64, SYG_S1_WAVES_PER_SIMD) void stft_mel_w1024_seg_kernel(
    const float* __restrict__ y, int64_t L, int64_t ldy, int hop, int pad, int64_t T, int64_t pairs_per_clip,
    int64_t n_pairs, const float* __restrict__ win, const float2* __restrict__ tw1024,
    const float4* __restrict__ segtab, int n_mels, float* __restrict__ mel_out, S1Rows rw) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  float* rowa = lds + w * S1_SCW;
  float* rowb = rowa + S1_ROW;
  float2* sc = reinterpret_cast<float2*>(rowa);
  float2* tw2l = reinterpret_cast<float2*>(lds + S1_WAVES * S1_SCW);
  float2* tw1l = tw2l + wfft::TW2_COMPLEX;
  float4* segl = reinterpret_cast<float4*>(tw1l + wfft::TW1_COMPLEX);
  wfft::Lane lc;
  wfft::init_lane(lc, lane);
  if (tid < 64) tw2l[(tid >> 4) * wfft::TW2_STRIDE + (tid & 15)] = tw1024[(16 * (tid >> 4) * (tid & 15)) & 1023];
  for (int i = tid; i < wfft::TW1_COMPLEX; i += S1_WAVES * 64) tw1l[i] = tw1024[(i & 63) * ((i >> 6) + 1)];
  const bool project = !ROWS || segtab != nullptr;
  if (project)
    for (int i = tid; i < S1_SEG_WORDS / 4; i += S1_WAVES * 64) segl[i] = segtab[i];
  // the window sits in LDS, not in 16 registers per lane: with it in registers the kernel spilled, and every reload of a
  // spilled register waits for ALL outstanding memory operations -- the next pair's samples included
  float* winl = reinterpret_cast<float*>(segl) + S1_SEG_WORDS;
  for (int i = tid; i < 1024; i += S1_WAVES * 64) winl[i] = win[i];
  int* cpl = reinterpret_cast<int*>(winl + 1024);     // ROWS: the contrast plan (lo / hi / k per band), lane = band
  // ROWS: the results of the pair's row functions wait here ([2 rows][3 registers][64 lanes] per wave) until the pair's
  // stores are issued behind the last call -- held in registers across the calls they were spilled
  float* resl = reinterpret_cast<float*>(cpl + 3 * SYG_MAX_BANDS) + w * (2 * 3 * 64);
  if (ROWS) {
#pragma unroll
    for (int r = 0; r < SYG_MAX_BANDS; ++r)
      if (tid == r) { cpl[r] = rw.lo[r]; cpl[SYG_MAX_BANDS + r] = rw.hi[r]; cpl[2 * SYG_MAX_BANDS + r] = rw.k[r]; }
    // (words of the rows that no bin is stored to -- the pad words, the base and the tail -- are read by the row
    // functions' masked lanes and by the projection's masked window words: cleared once, so that they hold numbers)
    for (int i = tid; i < S1_WAVES * S1_SCW; i += S1_WAVES * 64) lds[i] = 0.f;
  }
  __syncthreads();
  unsigned lk = 0;
  if (project) {
    const int* si = reinterpret_cast<const int*>(segl);
#pragma unroll
    for (int p = 0; p < 4; ++p) lk |= (unsigned)(si[4 * (128 * p + lane) + 2] | si[4 * (128 * p + lane) + 3]);
  }
  const bool scan8 = __builtin_amdgcn_ballot_w64((lk >> 24) != 0) != 0;

  // raw samples of the wave's two frames (x: frame ta, y: frame ta + 1; zero outside the clip: center=True's padding)
  float2 raw[16];
  auto fetch = [&](int64_t u) {
    const int64_t bq = u / pairs_per_clip;
    const float* yb = y + bq * ldy;
    const int64_t ta = (u - bq * pairs_per_clip) * 2, tb = ta + 1;
    const int64_t sa = ta * (int64_t)hop - pad, sb = tb * (int64_t)hop - pad;
    const bool hasb = tb < T;
    int lf = lane;
    asm volatile("" : "+v"(lf));                  // (per-lane addresses are recomputed per pair, not hoisted)
    if (sa >= 0 && sb + 1024 <= L && hasb) {
#pragma unroll
      for (int a = 0; a < 16; ++a) raw[a] = make_float2(yb[sa + 64 * a + lf], yb[sb + 64 * a + lf]);
    } else {
#pragma unroll
      for (int a = 0; a < 16; ++a) {
        const int64_t ia = sa + 64 * a + lf, ib = sb + 64 * a + lf;
        raw[a] = make_float2((ia >= 0 && ia < L) ? yb[ia] : 0.f, (hasb && ib >= 0 && ib < L) ? yb[ib] : 0.f);
      }
    }
  };
  const int64_t stride = (int64_t)gridDim.x * S1_WAVES;
  int64_t u = (int64_t)blockIdx.x * S1_WAVES + w;
  if (u < n_pairs) fetch(u);
  for (; u < n_pairs; u += stride) {
    const int64_t b = u / pairs_per_clip;
    const int64_t ta = (u - b * pairs_per_clip) * 2;
    float2 v[16];
#pragma unroll
    for (int a = 0; a < 16; ++a) {
      const float wva = winl[64 * a + lane];
      v[a] = make_float2(raw[a].x * wva, raw[a].y * wva);
    }
    if (u + stride < n_pairs) fetch(u + stride);       // the next pair's samples, behind this pair's transform
    float2 zk[2][4], zm[2][4], z512;
    if (ROWS) {                                        // (six lane constants live across the row functions were spilled: re-made per pair)
      int ll = lane;
      asm volatile("" : "+v"(ll));
      wfft::init_lane(lc, ll);
    }
    wfft::cfft1024(v, lc, sc, tw1l, tw2l, lane, zk, zm, z512);
    wave_lds_sync();                                   // the scratch is dead: the rows may be written
    int lq = lane;
    asm volatile("" : "+v"(lq));
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        const int k = wfft::bin_of(lq, j, d);
        const int kq = k <= 512 ? k : 1024 - k;                 // |A[k]| = |A[1024 - k]|: the pair's bin below 512
        const float ex = zk[j][d].x + zm[j][d].x, ey = zk[j][d].y - zm[j][d].y;
        const float ox = zk[j][d].y + zm[j][d].y, oy = zm[j][d].x - zk[j][d].x;
        rowa[s1_pos(kq)] = fmaf(ex, ex, ey * ey);               // 4 |A|^2 (taken back, exactly, at the store)
        rowb[s1_pos(kq)] = fmaf(ox, ox, oy * oy);
      }
    if (lq == 0) {
      rowa[s1_pos(512)] = 4.f * z512.x * z512.x;
      rowb[s1_pos(512)] = 4.f * z512.y * z512.y;
    }
    wave_lds_sync();
    const bool hasb = ta + 1 < T;
    // ROWS: statistics / contrast of the two rows first (out-of-line: a function's entry waits for every outstanding
    // memory operation, so nothing of this pair is stored before the last call; the next pair's samples, requested in
    // front of the transform, have had its whole duration to land).  The rows hold 4 |A|^2: PS = 1.
    if (ROWS) {
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        if (r == 1 && !hasb) break;
        lds_row pr = (lds_row)((r == 0 ? rowa : rowb) + S1_BASE);
        float3 f = make_float3(0.f, 0.f, 0.f);
        if (rw.stats_out != nullptr && rw.contrast_out != nullptr) {
          f = row_features<513, 1>(pr, lane, rw.binhz, rw.roll_percent, rw.bw_p, rw.smask, (lds_iptr)cpl, rw.n_rows, rw.ascending);
        } else if (rw.stats_out != nullptr) {
          f.x = row_stats<513, 1>(pr, lane, rw.binhz, rw.roll_percent, rw.bw_p, rw.smask);
        } else {
          const float2 pv = row_contrast_all<1>(pr, lane, (lds_iptr)cpl, rw.n_rows, rw.ascending);
          f.y = pv.x; f.z = pv.y;
        }
        int lr = lane;
        asm volatile("" : "+v"(lr));
        resl[(3 * r + 0) * 64 + lr] = f.x; resl[(3 * r + 1) * 64 + lr] = f.y; resl[(3 * r + 2) * 64 + lr] = f.z;
      }
    }
    if (project) {
      float* mo = mel_out + (b * n_mels) * T + ta;
      tri_project<4>(rowa, segl, lq, scan8, [&](int bw, float val) {
        const int band = bw & 255, r = bw >> 8;                   // (the host tags the band word with the row)
        if (r == 0 || hasb) mo[(int64_t)band * T + r] = 0.25f * val;
      });
    }
    if (ROWS) {
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        if (r == 1 && !hasb) break;
        const int64_t t = ta + r;
        if (rw.contrast_out != nullptr && lq < rw.n_rows) {
          rw.contrast_out[((b * 2 + 0) * rw.n_rows + lq) * T + t] = resl[(3 * r + 1) * 64 + lq];
          rw.contrast_out[((b * 2 + 1) * rw.n_rows + lq) * T + t] = resl[(3 * r + 2) * 64 + lq];
        }
        if (rw.stats_out != nullptr && lq < SYG_NSTAT && ((stats_row_mask(rw.smask) >> lq) & 1))
          rw.stats_out[(b * SYG_NSTAT + lq) * T + t] = resl[(3 * r + 0) * 64 + lq];
      }
    }
    wave_lds_sync();                                   // the rows are read: the next transform may use the scratch
  }
}

}  // namespace
}  // namespace syg

using namespace syg;

namespace syg {
namespace {
int w1024_launch(const char* who, const float* y, int64_t B, int64_t L, int64_t ldy, int hop, int center, int64_t T,
                 const float* window, const float* twiddle, const float* segtab, int n_segtab, int n_mels, float* mel_out,
                 const S1Rows* rows, void* stream) {
  SYG_REQUIRE(y && window && twiddle, "%s: null pointer argument", who);
  SYG_REQUIRE(B >= 1 && L >= 1 && ldy >= L, "%s: need B >= 1, L >= 1, ldy >= L", who);
  SYG_REQUIRE(hop >= 1, "%s: hop must be >= 1", who);
  const int64_t Texp = center ? 1 + L / hop : (L >= 1024 ? 1 + (L - 1024) / hop : 0);
  SYG_REQUIRE(T >= 1 && T == Texp, "%s: T=%lld does not match the framing rule (%lld)", who, (long long)T, (long long)Texp);
  if (segtab != nullptr) {
    SYG_REQUIRE(mel_out, "%s: a piece table without mel_out", who);
    SYG_REQUIRE(n_segtab == S1_SEG_WORDS, "%s: the piece table has %d words, this library reads %d "
                "(sygnals_amd._tables.pack_mel_segments_rows)", who, n_segtab, S1_SEG_WORDS);
    SYG_REQUIRE(((uintptr_t)segtab) % 16 == 0, "%s: the piece table must be 16-byte aligned", who);
    SYG_REQUIRE(n_mels >= 1 && n_mels <= 255, "%s: n_mels must be in [1, 255]", who);
  }
  const int64_t ppc = (T + 1) / 2, n_pairs = B * ppc;
  SYG_REQUIRE(n_pairs < ((int64_t)1 << 40), "%s: too many frames", who);
  const int pad = center ? 512 : 0;
  int dev = 0, cus = 256;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cus = 256;
  const size_t lds = ((size_t)S1_WAVES * S1_SCW + 2 * (wfft::TW2_COMPLEX + wfft::TW1_COMPLEX) + S1_SEG_WORDS + 1024 +
                      (rows ? 3 * SYG_MAX_BANDS + S1_WAVES * 2 * 3 * 64 : 0)) * sizeof(float);
  int64_t wgs = (n_pairs + S1_WAVES - 1) / S1_WAVES;
  const int64_t cap = (int64_t)cus * 2 * 2;            // two workgroups per CU resident (60 KiB of LDS each), two rounds
  if (wgs > cap) wgs = cap;
  S1Rows rw;
  memset(&rw, 0, sizeof(rw));
  if (rows) rw = *rows;
  auto kern = rows ? stft_mel_w1024_seg_kernel<true> : stft_mel_w1024_seg_kernel<false>;
  hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) {
    set_error("%s: cannot reserve %zu B LDS: %s", who, lds, hipGetErrorString(e));
    return SYG_E_LAUNCH;
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)wgs), dim3(S1_WAVES * 64), lds, (hipStream_t)stream, y, L, ldy, hop, pad, T, ppc,
                     n_pairs, window, (const float2*)twiddle, (const float4*)segtab, n_mels, mel_out, rw);
  SYG_CHECK_LAUNCH(who);
  return SYG_OK;
}
}  // namespace
}  // namespace syg

// y [B, L] (row stride ldy) -> mel_out [B, n_mels, T], power 2; segtab: the two-row table of
// sygnals_amd._tables.pack_mel_segments_rows(sr, 1024, n_mels, fmin, fmax, rows=2, row_words=568, row_base=4) (2048 words on the
// device, 16-byte aligned); window [1024]; twiddle: W_1024^k, k = 0 .. 1023.
extern "C" int syg_stft_mel_w1024_seg_f32(const float* y, int64_t B, int64_t L, int64_t ldy, int hop, int center, int64_t T,
                                          const float* window, const float* twiddle, const float* segtab, int n_segtab,
                                          int n_mels, float* mel_out, void* stream) {
  SYG_REQUIRE(segtab && mel_out, "stft_mel_w1024_seg: null pointer argument");
  return w1024_launch("stft_mel_w1024_seg", y, B, L, ldy, hop, center, T, window, twiddle, segtab, n_segtab, n_mels, mel_out,
                      nullptr, stream);
}

// The same launch with the per-frame statistics / contrast rows of syg_stft2048_mel_f32 (stats_out [B, SYG_NSTAT, T] rows
// selected by stats_mask; cplan_host / contrast_out [B, 2, n_rows, T]; bins 0 .. 512, bin frequency k sr / 1024) -- at least
// one of them -- and the mel block OPTIONAL (segtab == NULL and mel_out == NULL: statistics only, nothing projected).
extern "C" int syg_stft_rows_w1024_f32(const float* y, int64_t B, int64_t L, int64_t ldy, int hop, int center, int64_t T,
                                       const float* window, const float* twiddle, const float* segtab, int n_segtab,
                                       int n_mels, float* mel_out, float sr, float roll_percent, float bw_p, int stats_mask,
                                       float* stats_out, const int32_t* cplan_host, float* contrast_out, void* stream) {
  SYG_REQUIRE(stats_out || contrast_out, "stft_rows_w1024: no statistics requested (use syg_stft_mel_w1024_seg_f32)");
  SYG_REQUIRE((segtab == nullptr) == (mel_out == nullptr), "stft_rows_w1024: segtab and mel_out come together");
  SYG_REQUIRE(T < ((int64_t)1 << 27), "stft_rows_w1024: clip too long");
  S1Rows rw;
  memset(&rw, 0, sizeof(rw));
  if (stats_out) SYG_REQUIRE(sr > 0.f && roll_percent >= 0.f && roll_percent <= 1.f && bw_p > 0.f && (stats_mask & 31) != 0 &&
                                 stats_mask > 0 && stats_mask < 64, "stft_rows_w1024: invalid statistics parameters");
  if (contrast_out) {
    SYG_REQUIRE(cplan_host, "stft_rows_w1024: contrast_out given without cplan_host");
    rw.n_rows = cplan_host[0];
    SYG_REQUIRE(rw.n_rows >= 1 && rw.n_rows <= SYG_MAX_BANDS, "stft_rows_w1024: contrast rows must be in [1, %d]", SYG_MAX_BANDS);
    for (int r = 0; r < rw.n_rows; ++r) {
      rw.lo[r] = cplan_host[1 + r];
      rw.hi[r] = cplan_host[1 + SYG_MAX_BANDS + r];
      rw.k[r] = cplan_host[1 + 2 * SYG_MAX_BANDS + r];
      SYG_REQUIRE(rw.lo[r] >= 0 && rw.hi[r] <= 513 && rw.lo[r] < rw.hi[r] && rw.k[r] >= 1 && rw.k[r] <= rw.hi[r] - rw.lo[r],
                  "stft_rows_w1024: contrast band %d invalid (lo=%d hi=%d k=%d)", r, rw.lo[r], rw.hi[r], rw.k[r]);
    }
    rw.ascending = 1;
    for (int r = 1; r < rw.n_rows; ++r)
      if (rw.lo[r] < rw.hi[r - 1] - 1 || rw.hi[r] < rw.hi[r - 1]) rw.ascending = 0;
  }
  rw.binhz = sr / 1024.f; rw.roll_percent = roll_percent; rw.bw_p = bw_p; rw.smask = stats_mask;
  rw.stats_out = stats_out; rw.contrast_out = contrast_out;
  return w1024_launch("stft_rows_w1024", y, B, L, ldy, hop, center, T, window, twiddle, segtab, n_segtab, n_mels, mel_out, &rw,
                      stream);
}
