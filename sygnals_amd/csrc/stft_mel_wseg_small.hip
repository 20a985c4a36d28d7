// Fused STFT -> |X|^2 -> mel for frame_length 512 and 256 (256 is the frame length of the reference's short-signal tests,
// tests/test_features_manager.py:183-220; librosa.stft + np.abs(.)**2 + melspectrogram as manager.py:184-187, 198, 219-222
// call them), free-running waves: a wave owns FOUR (512) or EIGHT (256) frames per 1024-point complex transform -- the
// packings of stft_mel_pow2.hip (stft_mel_w512_kernel / stft_mel_w256_kernel: NF / 2 complex sequences of two real frames
// each, interleaved; the inter-sequence twiddles drop out of the powers) -- and projects its own power rows onto the mel
// bands by segment sums (mel_segments.h; one pass of 64 lanes per row, a four-row table, eight rows = two calls).  No
// weight matrix, no spectrogram in HBM, no workgroup barrier behind the table set-up; the samples of the next group of
// frames are requested before the current group is transformed.  power = 2 only.
// ROWS (syg_stft_rows_wsmall_f32): the same transform, the per-frame row functions of row_features.h (spectral statistics,
// contrast tail means: frequency_domain.py:24-386 as driven by manager.py:289-343) on the wave's 4 / 8 rows of 257 / 129
// bins INSTEAD of the projection -- extract_features(frame_length=512 / 256, [spectral features]) without a spectrogram in
// HBM (a mel block beside them is a second launch of the projection form: the [band][16 frames] tile's LDS holds the
// rows' results here).
#include "wave_fft.h"
#include <string.h>

namespace syg {
namespace {
#include "mel_segments.h"
#include "row_features.h"

constexpr int SS_WAVES = 16;                         // one workgroup per CU (the tables are then in LDS once)
constexpr int SS_GF = 16;                            // frames per group: a wave transforms 16 / NF groups of NF frames in a row and
                                                     // stores their band values as runs of 16 frames (64 bytes) per band
constexpr int SS_BASE = 4;                           // words in front of bin 0 (room for the lead of a short first piece)
constexpr int SS_SEG_WORDS = 4 * 2 * 64 * 4;         // four rows, one pass each
constexpr int SS_MAX_MELS = 48;                      // bands of the [band][16 frames] tile a wave keeps in LDS
constexpr int SS_GP = SS_GF + 1;                     // the tile's row stride (odd: the 40 lanes that hold a frame's bands write 40 banks)
template <int NF> struct SmallCfg;
// WINDOW: row words a lane reads per piece -- 17 (rows cut at 16-bin blocks) or 9 (256: cut at 8-bin blocks, the 129 bins
// still fit one pass of 64 lanes: half the masked sums per row)
template <> struct SmallCfg<4> { static constexpr int NFFT = 512, ROW = 296, LOGSEQ = 1, WINDOW = 17; static constexpr float SCALE = 0.0625f; };
template <> struct SmallCfg<8> { static constexpr int NFFT = 256, ROW = 160, LOGSEQ = 2, WINDOW = 9; static constexpr float SCALE = 0.015625f; };

__device__ __forceinline__ int ss_pos(int k) { return SS_BASE + k + (k >> 4); }      // == _tables.row_pos + row_base

struct SmallRows {                                    // arguments of the row functions (ROWS kernels)
  float binhz, roll_percent, bw_p;
  int smask;
  float* stats_out;                                   // [B, SYG_NSTAT, T] or null
  float* contrast_out;                                // [B, 2, n_rows, T] or null
  int n_rows, ascending;
  int lo[SYG_MAX_BANDS], hi[SYG_MAX_BANDS], k[SYG_MAX_BANDS];
};

template <int NF, bool ROWS>
__global__ __launch_bounds__(SS_WAVES * 64, 4) void stft_mel_wseg_small_kernel(
    const float* __restrict__ y, int64_t L, int64_t ldy, int hop, int pad, int64_t T, int64_t groups_per_clip,
    int64_t n_groups, const float* __restrict__ win, const float2* __restrict__ tw1024,
    const float4* __restrict__ segtab, int n_mels, float* __restrict__ mel_out, SmallRows rw) {
  typedef SmallCfg<NF> CF;
  constexpr int NBIN = CF::NFFT / 2 + 1, PS = (NF == 4) ? 2 : 3;      // the rows hold 16 |X|^2 / 64 |X|^2 = 4^PS |X|^2
  constexpr int ROW = CF::ROW, NSEQ = NF / 2, SPA = 64 / NSEQ;       // samples of a sequence per 64 elements of z
  constexpr int SCW = NF * ROW;
  static_assert(SCW >= 2 * wfft::SC_COMPLEX, "the exchange scratch must fit inside the rows");
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  float* row0 = lds + w * SCW;
  float2* sc = reinterpret_cast<float2*>(row0);
  float2* tw2l = reinterpret_cast<float2*>(lds + SS_WAVES * SCW);
  float2* tw1l = tw2l + wfft::TW2_COMPLEX;
  float4* segl = reinterpret_cast<float4*>(tw1l + wfft::TW1_COMPLEX);
  float* stg0 = reinterpret_cast<float*>(segl) + SS_SEG_WORDS;
  float* stg = stg0 + w * (SS_MAX_MELS * SS_GP);                                         // this wave's [band][16 frames] tile
  wfft::Lane lc;
  wfft::init_lane(lc, lane);
  if (tid < 64) tw2l[(tid >> 4) * wfft::TW2_STRIDE + (tid & 15)] = tw1024[(16 * (tid >> 4) * (tid & 15)) & 1023];
  for (int i = tid; i < wfft::TW1_COMPLEX; i += SS_WAVES * 64) tw1l[i] = tw1024[(i & 63) * ((i >> 6) + 1)];
  if (!ROWS)
    for (int i = tid; i < SS_SEG_WORDS / 4; i += SS_WAVES * 64) segl[i] = segtab[i];
  // element 64 a + lane of z: sequence r = lane % NSEQ = frames (first + 2 r, first + 2 r + 1), sample SPA a + lane / NSEQ
  const int rs = lane & (NSEQ - 1), nl = lane >> CF::LOGSEQ;
  // the window sits in LDS, not in 16 registers per lane: with it in registers the kernel spilled (11 registers at NF = 8),
  // and every reload of a spilled register waits for ALL outstanding memory operations -- the next group's samples included
  float* winl = stg0 + SS_WAVES * (SS_MAX_MELS * SS_GP);
  for (int i = tid; i < CF::NFFT; i += SS_WAVES * 64) winl[i] = win[i];
  int* cpl = reinterpret_cast<int*>(winl + CF::NFFT);     // ROWS: the contrast plan (lo / hi / k per band), lane = band
  if (ROWS) {
#pragma unroll
    for (int r = 0; r < SYG_MAX_BANDS; ++r)
      if (tid == r) { cpl[r] = rw.lo[r]; cpl[SYG_MAX_BANDS + r] = rw.hi[r]; cpl[2 * SYG_MAX_BANDS + r] = rw.k[r]; }
    // (row words that no bin is stored to -- pads, base, tail -- are read under masks: cleared once)
    for (int i = tid; i < SS_WAVES * SCW; i += SS_WAVES * 64) lds[i] = 0.f;
  }
  __syncthreads();
  unsigned lk = 0;
  if (!ROWS) {
    const int* si = reinterpret_cast<const int*>(segl);
#pragma unroll
    for (int p = 0; p < 4; ++p) lk |= (unsigned)(si[4 * (128 * p + lane) + 2] | si[4 * (128 * p + lane) + 3]);
  }
  const bool scan8 = __builtin_amdgcn_ballot_w64((lk >> 24) != 0) != 0;

  float2 raw[16];
  auto fetch = [&](int64_t bq, int64_t tfirst) {
    const float* yb = y + bq * ldy;
    const int64_t tf = tfirst + 2 * rs;
    const int64_t sa = tf * (int64_t)hop - pad, sb = sa + hop;
    const int64_t s_first = tfirst * (int64_t)hop - pad;
    int ln = nl;
    asm volatile("" : "+v"(ln));                  // (per-lane addresses are recomputed per group, not hoisted)
    if (s_first >= 0 && s_first + (NF - 1) * (int64_t)hop + CF::NFFT <= L && tfirst + NF - 1 < T) {
#pragma unroll
      for (int a = 0; a < 16; ++a) raw[a] = make_float2(yb[sa + SPA * a + ln], yb[sb + SPA * a + ln]);
    } else {
      const bool hasa = tf < T, hasb = tf + 1 < T;
#pragma unroll
      for (int a = 0; a < 16; ++a) {
        const int64_t ia = sa + SPA * a + ln, ib = sb + SPA * a + ln;
        raw[a] = make_float2((hasa && ia >= 0 && ia < L) ? yb[ia] : 0.f, (hasb && ib >= 0 && ib < L) ? yb[ib] : 0.f);
      }
    }
  };
  // this wave's units: the 16 / NF groups of NF frames of frame group g, then of g + stride, ...
  constexpr int UG = SS_GF / NF;
  const int64_t stride = (int64_t)gridDim.x * SS_WAVES;
  auto unit_of = [&](int64_t g, int uu, int64_t& bq, int64_t& tf) -> bool {
    if (g >= n_groups) return false;
    bq = g / groups_per_clip;
    tf = (g - bq * groups_per_clip) * SS_GF + (int64_t)uu * NF;
    return tf < T;
  };
  int64_t g = (int64_t)blockIdx.x * SS_WAVES + w, b = 0, tfirst = 0;
  int uu = 0;
  bool live = unit_of(g, 0, b, tfirst);
  if (live) fetch(b, tfirst);
  while (live) {
    float2 v[16];
#pragma unroll
    for (int a = 0; a < 16; ++a) {
      const float wva = winl[SPA * a + nl];
      v[a] = make_float2(raw[a].x * wva, raw[a].y * wva);
    }
    // the next unit's samples are requested before this unit is transformed
    int64_t g2 = g, b2 = 0, tf2 = 0;
    int uu2 = uu + 1;
    bool live2 = (uu2 < UG) && unit_of(g2, uu2, b2, tf2);
    if (!live2) { g2 = g + stride; uu2 = 0; live2 = unit_of(g2, 0, b2, tf2); }
    // (ROWS: the request waits until the row functions have returned -- 32 sample registers live across their calls were
    // spilled, and a spill's reload is a memory round trip)
    if (!ROWS && live2) fetch(b2, tf2);
    float2 zk[2][4], zm[2][4], z512;
    if (ROWS) {                                        // (lane constants live across the row functions were spilled: re-made per unit)
      int ll = lane;
      asm volatile("" : "+v"(ll));
      wfft::init_lane(lc, ll);
    }
    wfft::cfft1024(v, lc, sc, tw1l, tw2l, lane, zk, zm, z512);
    wave_lds_sync();                                   // the scratch is dead: the rows may be written
    int lq = lane;
    asm volatile("" : "+v"(lq));
    if (NF == 4) {
      // (za, zb, mb, ma) = (Z[k], Z[k + 512], Z[512 - k], Z[1024 - k]) -> 16 x the four frames' powers at one bin
      auto four = [&](float2 za, float2 zb, float2 mb, float2 ma, int bin) {
        const float sx = za.x + zb.x, sy = za.y + zb.y, dx = za.x - zb.x, dy = za.y - zb.y;
        const float mx = mb.x + ma.x, my = mb.y + ma.y, ex = mb.x - ma.x, ey = mb.y - ma.y;
        const float ax = sx + mx, ay = sy - my, bx = sx - mx, by = sy + my;
        const float cx = dx - ex, cy = dy + ey, gx = dx + ex, gy = dy - ey;
        const int q = ss_pos(bin);
        row0[q] = fmaf(ax, ax, ay * ay);
        row0[ROW + q] = fmaf(bx, bx, by * by);
        row0[2 * ROW + q] = fmaf(cx, cx, cy * cy);
        row0[3 * ROW + q] = fmaf(gx, gx, gy * gy);
      };
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int kb = wfft::bin_of(lq, j, 0);                       // 0 .. 127 (lane 0, unit 0: 0 -- overwritten below)
        four(zk[j][0], zk[j][2], zm[j][2], zm[j][0], kb);
        four(zk[j][1], zk[j][3], zm[j][3], zm[j][1], 256 - kb);
      }
      if (lq == 0) {
        // unit 0 of lane 0 holds Z at 0, 256, 128, 384 (zk) and 0, 768, 896, 640 (zm), Z[512] apart
        four(zk[0][0], z512, z512, zk[0][0], 0);
        four(zk[0][1], zm[0][1], zk[0][1], zm[0][1], 256);
        four(zk[0][2], zm[0][3], zk[0][3], zm[0][2], 128);
      }
    } else {
      // z[d] = Z[k + 256 d], m[d] = Z[256 - k + 256 d]  ->  64 x the eight frames' powers at bin k
      auto eight = [&](float2 z0, float2 z1, float2 z2, float2 z3, float2 m0, float2 m1, float2 m2, float2 m3, int bin) {
        float2 F[4], G[4];
        bfly4(z0, z1, z2, z3, F[0], F[1], F[2], F[3]);          // F[q] = sum_d z[d] (-i)^(d q): U_r = F[(4 - r) & 3]
        bfly4(m0, m1, m2, m3, G[0], G[1], G[2], G[3]);
        const int q = ss_pos(bin);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float2 U = F[(4 - r) & 3], V = G[(4 - r) & 3];
          // (-i)^r conj V
          const float2 Tt = r == 0 ? make_float2(V.x, -V.y) : r == 1 ? make_float2(-V.y, -V.x)
                          : r == 2 ? make_float2(-V.x, V.y) : make_float2(V.y, V.x);
          const float ax = U.x + Tt.x, ay = U.y + Tt.y, bx = U.x - Tt.x, by = U.y - Tt.y;
          row0[(2 * r) * ROW + q] = fmaf(ax, ax, ay * ay);
          row0[(2 * r + 1) * ROW + q] = fmaf(bx, bx, by * by);
        }
      };
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int kb = wfft::bin_of(lq, j, 0);
        eight(zk[j][0], zk[j][1], zk[j][2], zk[j][3], zm[j][3], zm[j][2], zm[j][1], zm[j][0], kb);
      }
      if (lq == 0) {
        eight(zk[0][0], zk[0][1], z512, zm[0][1], zk[0][1], z512, zm[0][1], zk[0][0], 0);
        eight(zk[0][2], zk[0][3], zm[0][3], zm[0][2], zk[0][2], zk[0][3], zm[0][3], zm[0][2], 128);
      }
    }
    wave_lds_sync();
    if (ROWS) {
      // statistics / contrast of the unit's NF rows, one out-of-line call per row (a function's entry waits for every
      // outstanding memory operation: the results of a row -- lanes 0 .. 15 of three registers -- wait in this wave's tile
      // area until the unit's last call has returned, then all of them are stored)
#pragma unroll 1
      for (int r = 0; r < NF; ++r) {
        if (tfirst + r >= T) break;
        lds_row pr = (lds_row)(row0 + r * ROW + SS_BASE);
        float3 f = make_float3(0.f, 0.f, 0.f);
        if (rw.stats_out != nullptr && rw.contrast_out != nullptr) {
          f = row_features<NBIN, PS>(pr, lane, rw.binhz, rw.roll_percent, rw.bw_p, rw.smask, (lds_iptr)cpl, rw.n_rows, rw.ascending);
        } else if (rw.stats_out != nullptr) {
          f.x = row_stats<NBIN, PS>(pr, lane, rw.binhz, rw.roll_percent, rw.bw_p, rw.smask);
        } else {
          const float2 pv = row_contrast_all<PS>(pr, lane, (lds_iptr)cpl, rw.n_rows, rw.ascending);
          f.y = pv.x; f.z = pv.y;
        }
        int lr = lane;
        asm volatile("" : "+v"(lr));
        if (lr < 16) { stg[(3 * r + 0) * 16 + lr] = f.x; stg[(3 * r + 1) * 16 + lr] = f.y; stg[(3 * r + 2) * 16 + lr] = f.z; }
      }
      wave_lds_sync();
#pragma unroll 1
      for (int r = 0; r < NF; ++r) {
        const int64_t t = tfirst + r;
        if (t >= T) break;
        if (rw.contrast_out != nullptr && lq < rw.n_rows) {
          rw.contrast_out[((b * 2 + 0) * rw.n_rows + lq) * T + t] = stg[(3 * r + 1) * 16 + lq];
          rw.contrast_out[((b * 2 + 1) * rw.n_rows + lq) * T + t] = stg[(3 * r + 2) * 16 + lq];
        }
        if (rw.stats_out != nullptr && lq < SYG_NSTAT && ((stats_row_mask(rw.smask) >> lq) & 1))
          rw.stats_out[(b * SYG_NSTAT + lq) * T + t] = stg[(3 * r + 0) * 16 + lq];
      }
      wave_lds_sync();                                 // the rows and the tile area are read: the next transform may run
      if (live2) fetch(b2, tf2);
      g = g2; uu = uu2; b = b2; tfirst = tf2; live = live2;
      continue;
    }
    // the band values go through a [band][16 frames] tile of this wave in LDS and leave as runs of 16 consecutive frames per
    // band (64 bytes): stored one by one from the lanes that hold them -- 40 scattered 4-byte stores per frame -- they cost
    // 30 % of the kernel at frame length 256 (300 against 208 us without any store; runs of 8 frames: 270)
#pragma unroll
    for (int h = 0; h < NF / 4; ++h)
      tri_project<4, CF::WINDOW>(row0 + 4 * h * ROW, segl, lq, scan8, [&](int bw, float val) {
        const int band = bw & 255, r = 4 * h + (bw >> 8);             // (the host tags the band word with the row)
        stg[band * SS_GP + uu * NF + r] = CF::SCALE * val;
      });
    wave_lds_sync();                                   // the rows are read (and the tile written): the next transform may use the scratch
    if (g2 != g || !live2) {                           // the frame group is complete: its tile leaves
      const int64_t t16 = tfirst - (int64_t)uu * NF;
      float* mo = mel_out + (b * n_mels) * T + t16;
      const int left = (int)((T - t16 < SS_GF) ? T - t16 : SS_GF);
      // a FIXED number of unconditional stores (lanes past the tile's bands / the clip's frames repeat the last valid
      // element: same address, same value): with a conditional store the compiler cannot count the stores behind the
      // next group's sample loads and waits for every one of them (s_waitcnt vmcnt(0)) before it touches the samples
      const int mlast = n_mels - 1, rlast = left - 1;
#pragma unroll
      for (int q = 0; q < SS_MAX_MELS * SS_GF / 64; ++q) {
        const int i = 64 * q + lq;
        int band = i >> 4, r = i & (SS_GF - 1);
        band = band < mlast ? band : mlast;
        r = r < rlast ? r : rlast;
        mo[(int64_t)band * T + r] = stg[band * SS_GP + r];
      }
      wave_lds_sync();
    }
    g = g2; uu = uu2; b = b2; tfirst = tf2; live = live2;
  }
}

template <int NF>
int launch_small(const float* y, int64_t B, int64_t L, int64_t ldy, int hop, int center, int64_t T, const float* window,
                 const float* twiddle, const float* segtab, int n_mels, float* mel_out, hipStream_t st,
                 const SmallRows* rows = nullptr) {
  typedef SmallCfg<NF> CF;
  const int64_t gpc = (T + SS_GF - 1) / SS_GF, n_groups = B * gpc;
  SYG_REQUIRE(n_groups < ((int64_t)1 << 40), "stft_mel_wseg_small: too many frames");
  const int pad = center ? CF::NFFT / 2 : 0;
  int dev = 0, cus = 256;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cus = 256;
  const size_t lds = ((size_t)SS_WAVES * NF * CF::ROW + 2 * (wfft::TW2_COMPLEX + wfft::TW1_COMPLEX) + SS_SEG_WORDS +
                      (size_t)SS_WAVES * SS_MAX_MELS * SS_GP + CF::NFFT + (rows ? 3 * SYG_MAX_BANDS : 0)) * sizeof(float);
  int64_t wgs = (n_groups + SS_WAVES - 1) / SS_WAVES;
  const int64_t cap = (int64_t)cus * 2;                // one workgroup per CU resident (141 / 148 KiB of LDS), two rounds
  if (wgs > cap) wgs = cap;
  SmallRows rw;
  memset(&rw, 0, sizeof(rw));
  if (rows) rw = *rows;
  auto kern = rows ? stft_mel_wseg_small_kernel<NF, true> : stft_mel_wseg_small_kernel<NF, false>;
  hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) {
    set_error("stft_mel_wseg_small: cannot reserve %zu B LDS: %s", lds, hipGetErrorString(e));
    return SYG_E_LAUNCH;
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)wgs), dim3(SS_WAVES * 64), lds, st, y, L, ldy, hop, pad, T, gpc, n_groups, window,
                     (const float2*)twiddle, (const float4*)segtab, n_mels, mel_out, rw);
  SYG_CHECK_LAUNCH("stft_mel_wseg_small");
  return SYG_OK;
}

}  // namespace
}  // namespace syg

using namespace syg;

// n_fft = 512 or 256, power 2: y [B, L] (row stride ldy) -> mel_out [B, n_mels, T].  segtab: the four-row table of
// sygnals_amd._tables.pack_mel_segments_rows(sr, n_fft, n_mels, fmin, fmax, rows=4, row_words=296 (512) / 160 (256),
// n_pass=1, block=16 (512) / 8 (256)) (2048 words on the device, 16-byte aligned); window [n_fft]; twiddle: W_1024^k, k = 0 .. 1023.
extern "C" int syg_stft_mel_wseg_small_f32(const float* y, int64_t B, int64_t L, int64_t ldy, int n_fft, int hop, int center,
                                           int64_t T, const float* window, const float* twiddle, const float* segtab,
                                           int n_segtab, int n_mels, float* mel_out, void* stream) {
  SYG_REQUIRE(y && window && twiddle && segtab && mel_out, "stft_mel_wseg_small: null pointer argument");
  SYG_REQUIRE(n_fft == 512 || n_fft == 256, "stft_mel_wseg_small: n_fft must be 512 or 256 (got %d)", n_fft);
  SYG_REQUIRE(B >= 1 && L >= 1 && ldy >= L, "stft_mel_wseg_small: need B >= 1, L >= 1, ldy >= L");
  SYG_REQUIRE(hop >= 1, "stft_mel_wseg_small: hop must be >= 1");
  const int64_t Texp = center ? 1 + L / hop : (L >= n_fft ? 1 + (L - n_fft) / hop : 0);
  SYG_REQUIRE(T >= 1 && T == Texp, "stft_mel_wseg_small: T=%lld does not match the framing rule (%lld)", (long long)T, (long long)Texp);
  SYG_REQUIRE(n_segtab == SS_SEG_WORDS, "stft_mel_wseg_small: the piece table has %d words, this library reads %d "
              "(sygnals_amd._tables.pack_mel_segments_rows)", n_segtab, SS_SEG_WORDS);
  SYG_REQUIRE(((uintptr_t)segtab) % 16 == 0, "stft_mel_wseg_small: the piece table must be 16-byte aligned");
  SYG_REQUIRE(n_mels >= 1 && n_mels <= SS_MAX_MELS, "stft_mel_wseg_small: n_mels must be in [1, %d]", SS_MAX_MELS);
  if (n_fft == 512) return launch_small<4>(y, B, L, ldy, hop, center, T, window, twiddle, segtab, n_mels, mel_out, (hipStream_t)stream);
  return launch_small<8>(y, B, L, ldy, hop, center, T, window, twiddle, segtab, n_mels, mel_out, (hipStream_t)stream);
}

// The per-frame statistics / contrast rows of syg_stft2048_mel_f32 for frame lengths 512 / 256 (bins 0 .. n_fft / 2, bin
// frequency k sr / n_fft) from the segment-sum kernel's transform, nothing projected: stats_out [B, SYG_NSTAT, T] (rows
// selected by stats_mask) and / or cplan_host + contrast_out [B, 2, n_rows, T] -- at least one.
extern "C" int syg_stft_rows_wsmall_f32(const float* y, int64_t B, int64_t L, int64_t ldy, int n_fft, int hop, int center,
                                        int64_t T, const float* window, const float* twiddle, float sr, float roll_percent,
                                        float bw_p, int stats_mask, float* stats_out, const int32_t* cplan_host,
                                        float* contrast_out, void* stream) {
  SYG_REQUIRE(y && window && twiddle, "stft_rows_wsmall: null pointer argument");
  SYG_REQUIRE(stats_out || contrast_out, "stft_rows_wsmall: no statistics requested");
  SYG_REQUIRE(n_fft == 512 || n_fft == 256, "stft_rows_wsmall: n_fft must be 512 or 256 (got %d)", n_fft);
  SYG_REQUIRE(B >= 1 && L >= 1 && ldy >= L, "stft_rows_wsmall: need B >= 1, L >= 1, ldy >= L");
  SYG_REQUIRE(hop >= 1, "stft_rows_wsmall: hop must be >= 1");
  const int64_t Texp = center ? 1 + L / hop : (L >= n_fft ? 1 + (L - n_fft) / hop : 0);
  SYG_REQUIRE(T >= 1 && T == Texp && T < ((int64_t)1 << 27), "stft_rows_wsmall: T=%lld does not match the framing rule (%lld)",
              (long long)T, (long long)Texp);
  SmallRows rw;
  memset(&rw, 0, sizeof(rw));
  if (stats_out) SYG_REQUIRE(sr > 0.f && roll_percent >= 0.f && roll_percent <= 1.f && bw_p > 0.f && (stats_mask & 31) != 0 &&
                                 stats_mask > 0 && stats_mask < 64, "stft_rows_wsmall: invalid statistics parameters");
  if (contrast_out) {
    SYG_REQUIRE(cplan_host, "stft_rows_wsmall: contrast_out given without cplan_host");
    rw.n_rows = cplan_host[0];
    SYG_REQUIRE(rw.n_rows >= 1 && rw.n_rows <= SYG_MAX_BANDS, "stft_rows_wsmall: contrast rows must be in [1, %d]", SYG_MAX_BANDS);
    for (int r = 0; r < rw.n_rows; ++r) {
      rw.lo[r] = cplan_host[1 + r];
      rw.hi[r] = cplan_host[1 + SYG_MAX_BANDS + r];
      rw.k[r] = cplan_host[1 + 2 * SYG_MAX_BANDS + r];
      SYG_REQUIRE(rw.lo[r] >= 0 && rw.hi[r] <= n_fft / 2 + 1 && rw.lo[r] < rw.hi[r] && rw.k[r] >= 1 &&
                      rw.k[r] <= rw.hi[r] - rw.lo[r],
                  "stft_rows_wsmall: contrast band %d invalid (lo=%d hi=%d k=%d)", r, rw.lo[r], rw.hi[r], rw.k[r]);
    }
    rw.ascending = 1;
    for (int r = 1; r < rw.n_rows; ++r)
      if (rw.lo[r] < rw.hi[r - 1] - 1 || rw.hi[r] < rw.hi[r - 1]) rw.ascending = 0;
  }
  rw.binhz = sr / (float)n_fft; rw.roll_percent = roll_percent; rw.bw_p = bw_p; rw.smask = stats_mask;
  rw.stats_out = stats_out; rw.contrast_out = contrast_out;
  if (n_fft == 512)
    return launch_small<4>(y, B, L, ldy, hop, center, T, window, twiddle, nullptr, 0, nullptr, (hipStream_t)stream, &rw);
  return launch_small<8>(y, B, L, ldy, hop, center, T, window, twiddle, nullptr, 0, nullptr, (hipStream_t)stream, &rw);
}
