// Fused STFT(n_fft = 2^k, 64 ... 4096) -> |X|^power -> mel filterbank (-> power_to_db -> DCT-II) for the frame
// lengths the wave-FFT kernel of stft_mel.hip (n_fft = 2048) does not take: the reference's own tests and CLI use
// frame_length 1024 and 256 (tests/test_features_manager.py:183-220, cli/features_cmd.py:35).  Round 2 ran these
// as complex STFT -> |X|^2 -> dense filterbank -> dB/DCT: four launches and a power-spectrogram round trip through HBM
// (2.4 GB at 1024 clips x 1 s, n_fft 1024 / hop 256).  Here a workgroup takes 16 consecutive frames of one clip:
//   transform  every wave runs whole frames by itself (block_fft<WAVE>: Stockham radix-8 in LDS buffers of its own, no
//              workgroup barrier), packs the real frame as n_fft/2 complex points, splits, and leaves |X|^power as one
//              row of the tile's LDS power matrix [16][F];
//   barrier
//   project    mel[m, t] = sum_f basis[m, f] P[t, f] on v_mfma_f32_16x16x4_f32 (exact fp32): B operand = the 16 LDS
//              rows (one ds_read_b128 per four k-steps), A operand = the zero-padded dense filterbank from L2;
//   barrier
// TILE mode writes the mel tile; CLIP mode (a workgroup owns a whole clip) keeps the clip's mel matrix in LDS and ends
// with power_to_db(ref = max of the clip, amin, top_db) + DCT-II rows (+ lifter): samples in, MFCCs out, one launch.
// Reference chain: manager.py:184-187, 198, 219-223 -> librosa.stft / melspectrogram / power_to_db; cepstral.py:106-115.
#include "wave_fft.h"
#include <string.h>

namespace syg {
namespace {

typedef float v4f __attribute__((ext_vector_type(4)));

struct Pow2Mfcc {
  const float* dct;      // [n_mfcc, n_mels]
  const float* lifter;   // [n_mfcc] or null
  float* out;            // [B, n_mfcc, T]
  int n_mfcc;
  int ref_is_max;
  float ref_value, amin, top_db;
  int tp;                // 16 * tiles per clip: row stride of the LDS mel matrix
};

// Real-input split: Z = FFT_M(z), z[m] = x[2m] + i x[2m+1]; X[k], k in [0, M];  tw2[k] = W_{2M}^k.
__device__ __forceinline__ float2 rbin(const float2* Z, int M, int k, const float2* __restrict__ tw2) {
  const float2 zk = Z[k & (M - 1)], zm = Z[(M - k) & (M - 1)];
  const float2 E = make_float2(zk.x + zm.x, zk.y - zm.y);
  const float2 O = make_float2(zk.y + zm.y, zm.x - zk.x);
  const float2 wO = cmul(tw2[k], O);
  return make_float2(0.5f * (E.x + wO.x), 0.5f * (E.y + wO.y));
}

// LDS (floats): fft [NW][2][M] complex | P [16][PS] | twl [n_fft + M] complex (the twiddle tables: the Stockham passes
// and the split read them per butterfly -- from global memory that was a dependent L2 round trip per pass at two waves per
// SIMD) | CLIP: clipmel [n_mels][tp], red [NW + 2]
template <int NW, bool CLIP>
__global__ __launch_bounds__(NW * 64) void stft_mel_pow2_kernel(
    const float* __restrict__ y, int64_t L, int64_t ldy, int n_fft, int hop, int pad, int64_t T,
    const float* __restrict__ win, const float2* __restrict__ tw, const float* __restrict__ basis_p, int Fp, int n_mels,
    int power, float* __restrict__ mel_out, int tiles_per_clip, Pow2Mfcc mf) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int M = n_fft >> 1, F = M + 1, PS = Fp + 4;
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  float2* fx = reinterpret_cast<float2*>(lds) + (size_t)w * 2 * M;
  float2* fz = fx + M;
  float* P = lds + (size_t)NW * 4 * M;
  float2* twl = reinterpret_cast<float2*>(P + 16 * PS);
  float* clipmel = P + 16 * PS + 2 * (n_fft + M);
  float* red = clipmel + (CLIP ? n_mels * mf.tp : 0);
  for (int i = tid; i < n_fft + M; i += NW * 64) twl[i] = tw[i];

  const int64_t b = CLIP ? blockIdx.x : blockIdx.x / tiles_per_clip;
  const int tile0 = CLIP ? 0 : (int)(blockIdx.x - b * tiles_per_clip);
  const int ntile = CLIP ? tiles_per_clip : 1;
  const float* yb = y + b * ldy;
  // the columns [F, PS) of the power rows meet zero weights: they must hold finite values
  for (int i = tid; i < 16 * PS; i += NW * 64) P[i] = 0.f;
  __syncthreads();

  const int n_mt = (n_mels + 15) >> 4;
  const int n = lane & 15, kk = lane >> 4;
  float cmax = 0.f;
  for (int ti = 0; ti < ntile; ++ti) {
    const int64_t t0 = (int64_t)(tile0 + ti) * 16;
    // ---- transforms: wave w takes the frames w, w + NW, ... of the tile
    for (int fi = w; fi < 16; fi += NW) {
      const int64_t t = t0 + fi;
      float* prow = P + fi * PS;
      if (t < T) {
        const int64_t s0 = t * (int64_t)hop - pad;
        if (s0 >= 0 && s0 + n_fft <= L) {
          for (int m = lane; m < M; m += 64)
            fx[m] = make_float2(yb[s0 + 2 * m] * win[2 * m], yb[s0 + 2 * m + 1] * win[2 * m + 1]);
        } else {
          for (int m = lane; m < M; m += 64) {
            const int64_t s = s0 + 2 * m;
            const float a = (s >= 0 && s < L) ? yb[s] * win[2 * m] : 0.f;
            const float c = (s + 1 >= 0 && s + 1 < L) ? yb[s + 1] * win[2 * m + 1] : 0.f;
            fx[m] = make_float2(a, c);
          }
        }
        wave_lds_sync();
        const float2* Z = block_fft<true>(fx, fz, M, twl + n_fft, lane, 64);
        for (int k = lane; k <= M; k += 64) {
          const float2 X = rbin(Z, M, k, twl);
          const float p2 = fmaf(X.x, X.x, X.y * X.y);
          prow[k] = (power == 2) ? p2 : sqrtf(p2);
        }
        wave_lds_sync();                 // the buffers are free for the wave's next frame
      } else {
        for (int k = lane; k < F; k += 64) prow[k] = 0.f;
      }
    }
    __syncthreads();
    // ---- projection: wave w takes the 16-row mel tiles w, w + NW, ...
    for (int mt = w; mt < n_mt; mt += NW) {
      const float* arow = basis_p + (size_t)(16 * mt + n) * Fp + 4 * kk;
      const float* brow = P + n * PS + 4 * kk;
      v4f acc = {0.f, 0.f, 0.f, 0.f}, acc2 = {0.f, 0.f, 0.f, 0.f};
      for (int s = 0; s < Fp; s += 16) {
        const float4 a4 = *reinterpret_cast<const float4*>(arow + s);
        const float4 b4 = *reinterpret_cast<const float4*>(brow + s);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.x, b4.x, acc, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.y, b4.y, acc2, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.z, b4.z, acc, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.w, b4.w, acc2, 0, 0, 0);
      }
      acc += acc2;
      // D[row 4 kk + i][col n]: mel row 16 mt + 4 kk + i of frame t0 + n
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int m = 16 * mt + 4 * kk + i;
        if (m < n_mels) {
          if (CLIP) {
            clipmel[m * mf.tp + (int)t0 + n] = acc[i];     // frames >= T hold 0 (their rows were cleared)
            cmax = fmaxf(cmax, acc[i]);
          }
          if (mel_out != nullptr && t0 + n < T) mel_out[(b * n_mels + m) * T + t0 + n] = acc[i];
        }
      }
    }
    __syncthreads();                     // every projection has read the rows: the next tile may overwrite them
  }
  if (!CLIP) return;

  // ---- clip epilogue: librosa.power_to_db(S, ref=np.max) (manager.py:223) -> scipy.fft.dct rows (cepstral.py:106-115)
  cmax = wave_max(cmax);
  if (lane == 0) red[w] = cmax;
  __syncthreads();
  if (tid == 0) {
    float m = red[0];
    for (int i = 1; i < NW; ++i) m = fmaxf(m, red[i]);
    const float ref = mf.ref_is_max ? m : fabsf(mf.ref_value);
    const float reflog = syg_log2(fmaxf(mf.amin, ref));
    red[NW] = reflog;
    red[NW + 1] = (mf.top_db >= 0.f) ? SYG_DB_PER_LOG2 * (syg_log2(fmaxf(mf.amin, m)) - reflog) - mf.top_db : -3.4e38f;
  }
  __syncthreads();
  const float reflog = red[NW], flo = red[NW + 1];
  const int nm = n_mels * mf.tp;
  for (int i = tid; i < nm; i += NW * 64)
    clipmel[i] = fmaxf(SYG_DB_PER_LOG2 * (syg_log2(fmaxf(mf.amin, clipmel[i])) - reflog), flo);   // exact 0 at x == ref
  __syncthreads();
  const int ktiles = (mf.n_mfcc + 15) >> 4, ttiles = mf.tp >> 4;
  for (int ot = w; ot < ktiles * ttiles; ot += NW) {
    const int kt = ot / ttiles, tq = ot - kt * ttiles;
    const int krow = kt * 16 + n, tcol = tq * 16 + n;
    v4f acc = {0.f, 0.f, 0.f, 0.f};
    for (int m0 = 0; m0 < n_mels; m0 += 4) {
      const int m = m0 + kk;
      const float a = (krow < mf.n_mfcc && m < n_mels) ? mf.dct[krow * n_mels + m] : 0.f;
      const float bv = (m < n_mels) ? clipmel[m * mf.tp + tcol] : 0.f;
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bv, acc, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int k = kt * 16 + 4 * kk + r;
      if (k < mf.n_mfcc && tcol < T) {
        float v = acc[r];
        if (mf.lifter) v *= mf.lifter[k];
        mf.out[(b * mf.n_mfcc + k) * T + tcol] = v;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// n_fft = 1024 on the wave FFT (wave_fft.h): TWO real frames per 1024-point complex transform.  With z[n] = a[n] + i b[n]
// (a, b: two windowed frames), A[k] = (Z[k] + conj Z[1024 - k]) / 2 and B[k] = -i (Z[k] - conj Z[1024 - k]) / 2 -- no
// twiddle in the split -- and the wave FFT hands every lane Z[k] and Z[1024 - k] of its bins together.  A wave owns the
// frames (2 w, 2 w + 1) of the 16-frame tile: one transform, 16 + 16 row stores, everything else as above.
// The projection is split over two K halves per 16-row mel tile (six of the eight waves work at 40 mels); the second
// halves leave their partial tiles in LDS and the first halves add them in a fixed order.
// The two power rows of a wave ALIAS its exchange scratch (written behind the transform, read by the projection, dead
// before the next transform): 46 KiB (tile form) / 79 KiB (clip form at 40 mels x 192 frames) of LDS per workgroup, two
// workgroups per CU.  The window is read from global memory (4 KiB, cache resident) beside the samples.
// LDS (floats): sc = P [8][2 * PS] | tw2l | tw1l | part [2 n_mt][256] | CLIP: clipmel, red
struct W1024Lds {
  static constexpr int FP = 528, PS = FP + 4;
  static constexpr int SCW = 2 * PS;                                  // per-wave scratch: 1064 floats >= 528 complex
  static constexpr int O_SC = 0;
  static constexpr int O_TW2 = O_SC + 8 * SCW;
  static constexpr int O_TW1 = O_TW2 + wfft::TW2_COMPLEX * 2;
  static constexpr int O_PART = O_TW1 + wfft::TW1_COMPLEX * 2;
  static_assert(SCW >= 2 * wfft::SC_COMPLEX, "the exchange scratch must fit the two aliased rows");
};

// tile form: 79 registers -> six waves per SIMD, three workgroups per CU (46 KiB of LDS each): 231 us per 1024 clips x 1 s
// against 269 us at the compiler's own choice (85 registers, two workgroups)
#ifndef SYG_P2_TILE_WAVES
#define SYG_P2_TILE_WAVES 6
#endif
template <bool CLIP>
__global__ __launch_bounds__(512, CLIP ? 4 : SYG_P2_TILE_WAVES) void stft_mel_w1024_kernel(
    const float* __restrict__ y, int64_t L, int64_t ldy, int hop, int pad, int64_t T, const float* __restrict__ win,
    const float2* __restrict__ tw1024, const float* __restrict__ basis_p, int n_mels, int power,
    float* __restrict__ mel_out, int tiles_per_clip, Pow2Mfcc mf) {
  typedef W1024Lds LM;
  constexpr int NW = 8, FP = LM::FP, PS = LM::PS;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  float2* sc = reinterpret_cast<float2*>(lds + LM::O_SC + w * LM::SCW);
  float2* tw2l = reinterpret_cast<float2*>(lds + LM::O_TW2);
  float2* tw1l = reinterpret_cast<float2*>(lds + LM::O_TW1);
  float* P = lds + LM::O_SC;                       // row fi at fi * PS: rows 2 w, 2 w + 1 inside wave w's scratch
  float* part = lds + LM::O_PART;
  const int n_mt = (n_mels + 15) >> 4;
  float* clipmel = part + 2 * n_mt * 256;
  float* red = clipmel + (CLIP ? n_mels * mf.tp : 0);

  wfft::Lane lc;
  wfft::init_lane(lc, lane);

  // CLIP: a workgroup owns clip blockIdx.x (all its tiles, the next tile's samples requested one tile ahead).  TILE: one
  // tile per workgroup (a persistent form with the same prefetch needed 128 registers and spilled: 323 vs 270 us).
  const int64_t gt0 = CLIP ? (int64_t)blockIdx.x * tiles_per_clip : (int64_t)blockIdx.x;
  const int ntile = CLIP ? tiles_per_clip : 1;
  const int n = lane & 15, kk = lane >> 4;
  // projection units: (mel tile, K half); unit u -> wave u (n_mt <= 4 here: up to 8 units); more tiles: round robin
  const int ngrp = FP / 16, gh = (ngrp + 1) / 2;                     // 33 groups of 16 bins: 17 + 16
  float cmax = 0.f;
  // raw samples of the wave's two frames (x: frame 2 w, y: frame 2 w + 1; zero outside the clip: center=True's padding)
  float2 raw[16];
  float wv[16];
  auto fetch = [&](int64_t gt) {                  // tile gt of the batch: clip gt / tiles_per_clip, first frame 16 (gt % ...)
    const int64_t bq = gt / tiles_per_clip;
    const float* yb = y + bq * ldy;
    const int64_t ta = (gt - bq * tiles_per_clip) * 16 + 2 * w, tb = ta + 1;
    if (ta >= T) return;
    const int64_t sa = ta * (int64_t)hop - pad, sb = tb * (int64_t)hop - pad;
    const bool hasb = tb < T;
    int lf = lane;
    asm volatile("" : "+v"(lf));                  // (per-lane addresses are recomputed per tile, not hoisted)
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
#pragma unroll
  for (int a = 0; a < 16; ++a) wv[a] = win[64 * a + lane];
  fetch(gt0);
  // tables of the wave FFT from tw1024[m] = W_1024^m:  W_64^(b' c') = W_1024^(16 b' c'),  W_1024^(b c), b c <= 945
  // (behind the sample requests: the table reads and the barrier then overlap the samples' way through memory)
  if (tid < 64) tw2l[(tid >> 4) * wfft::TW2_STRIDE + (tid & 15)] = tw1024[(16 * (tid >> 4) * (tid & 15)) & 1023];
  for (int i = tid; i < wfft::TW1_COMPLEX; i += NW * 64) tw1l[i] = tw1024[(i & 63) * ((i >> 6) + 1)];
  __syncthreads();
  for (int ti = 0; ti < ntile; ++ti) {
    const int64_t gt = gt0 + ti;
    const int64_t b = gt / tiles_per_clip;
    const int64_t t0 = (gt - b * tiles_per_clip) * 16;
    // ---- transform of the frames 2 w (real part) and 2 w + 1 (imaginary part); the samples were requested one tile
    // ahead (raw[], wv[]: the loads land behind the previous tile's projection)
    {
      const int64_t ta = t0 + 2 * w;
      float* rowa = P + (2 * w) * PS;
      float* rowb = rowa + PS;
      if (ta < T) {
        float2 v[16];
#pragma unroll
        for (int a = 0; a < 16; ++a) v[a] = make_float2(raw[a].x * wv[a], raw[a].y * wv[a]);
        float2 zk[2][4], zm[2][4], z512;
        wfft::cfft1024(v, lc, sc, tw1l, tw2l, lane, zk, zm, z512);
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int d = 0; d < 4; ++d) {
            const int k = wfft::bin_of(lane, j, d);
            const int kq = k <= 512 ? k : 1024 - k;                 // |A[k]| = |A[1024 - k]|: the pair's bin below 512
            const float ex = zk[j][d].x + zm[j][d].x, ey = zk[j][d].y - zm[j][d].y;
            const float ox = zk[j][d].y + zm[j][d].y, oy = zm[j][d].x - zk[j][d].x;
            const float pa = 0.25f * fmaf(ex, ex, ey * ey), pb = 0.25f * fmaf(ox, ox, oy * oy);
            rowa[kq] = (power == 2) ? pa : sqrtf(pa);
            rowb[kq] = (power == 2) ? pb : sqrtf(pb);
          }
        if (lane == 0) {
          rowa[512] = (power == 2) ? z512.x * z512.x : fabsf(z512.x);
          rowb[512] = (power == 2) ? z512.y * z512.y : fabsf(z512.y);
        }
        if (lane >= 1 && lane < PS - 512) { rowa[512 + lane] = 0.f; rowb[512 + lane] = 0.f; }   // [513, PS): zero weights
      } else {
        for (int k = lane; k < PS; k += 64) { rowa[k] = 0.f; rowb[k] = 0.f; }
      }
#ifndef SYG_P2_LATEFETCH
      if (ti + 1 < ntile) fetch(gt + 1);
#endif
    }
    __syncthreads();
    // ---- projection
    for (int u = w; u < 2 * n_mt; u += NW) {
      const int mt = u >> 1, half = u & 1;
      const int g0 = half ? gh : 0, g1 = half ? ngrp : gh;
      const float* arow = basis_p + (size_t)(16 * mt + n) * FP + 4 * kk;
      const float* brow = P + n * PS + 4 * kk;
      v4f acc = {0.f, 0.f, 0.f, 0.f}, acc2 = {0.f, 0.f, 0.f, 0.f};
      for (int g = g0; g < g1; ++g) {
        const float4 a4 = *reinterpret_cast<const float4*>(arow + 16 * g);
        const float4 b4 = *reinterpret_cast<const float4*>(brow + 16 * g);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.x, b4.x, acc, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.y, b4.y, acc2, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.z, b4.z, acc, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.w, b4.w, acc2, 0, 0, 0);
      }
      acc += acc2;
      // both halves leave their sums in `part`; the first half's wave adds them behind the barrier
      float* dst = part + u * 256 + lane * 4;
      dst[0] = acc[0]; dst[1] = acc[1]; dst[2] = acc[2]; dst[3] = acc[3];
    }
    __syncthreads();                     // rows read, partial tiles written
    for (int u = w; u < 2 * n_mt; u += NW) {
      if (u & 1) continue;
      const int mt = u >> 1;
      const float* kp = part + u * 256 + lane * 4;
      const float* pt = kp + 256;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float sum = kp[i] + pt[i];                              // first half + second half, fixed order
        const int m = 16 * mt + 4 * kk + i;
        if (m < n_mels) {
          if (CLIP) {
            clipmel[m * mf.tp + (int)t0 + n] = sum;
            cmax = fmaxf(cmax, sum);
          }
          if (mel_out != nullptr && t0 + n < T) mel_out[(b * n_mels + m) * T + t0 + n] = sum;
        }
      }
    }
#ifdef SYG_P2_LATEFETCH
    if (ti + 1 < ntile) fetch(gt + 1);          // (timing variant: the loads are requested right in front of their use)
#endif
    // (the next transform overwrites the rows -- every projection read them before the barrier above -- and `part` is
    // next written behind the next tile's first barrier)
  }
  if (!CLIP) return;
  const int64_t b = blockIdx.x;
  __syncthreads();
  cmax = wave_max(cmax);
  if (lane == 0) red[w] = cmax;
  __syncthreads();
  if (tid == 0) {
    float m = red[0];
    for (int i = 1; i < NW; ++i) m = fmaxf(m, red[i]);
    const float ref = mf.ref_is_max ? m : fabsf(mf.ref_value);
    const float reflog = syg_log2(fmaxf(mf.amin, ref));
    red[NW] = reflog;
    red[NW + 1] = (mf.top_db >= 0.f) ? SYG_DB_PER_LOG2 * (syg_log2(fmaxf(mf.amin, m)) - reflog) - mf.top_db : -3.4e38f;
  }
  __syncthreads();
  const float reflog = red[NW], flo = red[NW + 1];
  const int nm = n_mels * mf.tp;
  for (int i = tid; i < nm; i += NW * 64)
    clipmel[i] = fmaxf(SYG_DB_PER_LOG2 * (syg_log2(fmaxf(mf.amin, clipmel[i])) - reflog), flo);
  __syncthreads();
  const int ktiles = (mf.n_mfcc + 15) >> 4, ttiles = mf.tp >> 4;
  for (int ot = w; ot < ktiles * ttiles; ot += NW) {
    const int kt = ot / ttiles, tq = ot - kt * ttiles;
    const int krow = kt * 16 + n, tcol = tq * 16 + n;
    v4f acc = {0.f, 0.f, 0.f, 0.f};
    for (int m0 = 0; m0 < n_mels; m0 += 4) {
      const int m = m0 + kk;
      const float a = (krow < mf.n_mfcc && m < n_mels) ? mf.dct[krow * n_mels + m] : 0.f;
      const float bv = (m < n_mels) ? clipmel[m * mf.tp + tcol] : 0.f;
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bv, acc, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int k = kt * 16 + 4 * kk + r;
      if (k < mf.n_mfcc && tcol < T) {
        float v = acc[r];
        if (mf.lifter) v *= mf.lifter[k];
        mf.out[(b * mf.n_mfcc + k) * T + tcol] = v;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// n_fft = 512 on the wave FFT: FOUR real frames per 1024-point complex transform.  With p = a + i b and q = c + i d (two
// pairs of windowed 512-sample frames) interleaved as z[2n] = p[n], z[2n + 1] = q[n]:
//   Z[k] = P[k] + W_1024^k Q[k],  Z[k + 512] = P[k] - W_1024^k Q[k]      (P, Q: the 512-point transforms of p, q)
// and the real-input splits of P and Q need (k, 512 - k).  A lane of the wave FFT holds Z at k, k + 256, k + 512,
// k + 768 and at their mirrors 1024 - ..: with S = Z[k] + Z[k + 512], D = Z[k] - Z[k + 512], Sm = Z[512 - k] + Z[1024 - k],
// Dm = Z[512 - k] - Z[1024 - k] the four POWERS at bin k are
//   16 |A|^2 = |S + conj Sm|^2   16 |B|^2 = |S - conj Sm|^2   16 |C|^2 = |D - conj Dm|^2   16 |D|^2 = |D + conj Dm|^2
// -- the twiddle W^k only turns C and D and drops out of their magnitudes.  The pairs (k + 256, k + 768) give bin 256 - k
// the same way; lane 0's first unit (the self-mirrored groups) supplies bins 0, 128 and 256.
// A wave owns the frames 4 w .. 4 w + 3 of a 32-frame tile; its four power rows alias its exchange scratch.  Tile form only
// (the clip form of this frame length is the Stockham kernel above).
struct W512Lds {
  static constexpr int FP = 272, PS = FP + 4;
  static constexpr int SCW = 4 * PS;                                  // per-wave scratch: 1104 floats >= 528 complex
  static constexpr int O_SC = 0;
  static constexpr int O_TW2 = O_SC + 8 * SCW;
  static constexpr int O_TW1 = O_TW2 + wfft::TW2_COMPLEX * 2;
  static constexpr int TOTAL = O_TW1 + wfft::TW1_COMPLEX * 2;
  static_assert(SCW >= 2 * wfft::SC_COMPLEX, "the exchange scratch must fit inside the four aliased rows");
};

__global__ __launch_bounds__(512, SYG_P2_TILE_WAVES) void stft_mel_w512_kernel(
    const float* __restrict__ y, int64_t L, int64_t ldy, int hop, int pad, int64_t T, const float* __restrict__ win,
    const float2* __restrict__ tw512, const float* __restrict__ basis_p, int n_mels, int power,
    float* __restrict__ mel_out, int tiles_per_clip) {
  typedef W512Lds LM;
  constexpr int NW = 8, FP = LM::FP, PS = LM::PS;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  float2* sc = reinterpret_cast<float2*>(lds + LM::O_SC + w * LM::SCW);
  float2* tw2l = reinterpret_cast<float2*>(lds + LM::O_TW2);
  float2* tw1l = reinterpret_cast<float2*>(lds + LM::O_TW1);
  float* P = lds + LM::O_SC;                       // row fi at fi * PS: rows 4 w .. 4 w + 3 inside wave w's scratch
  wfft::Lane lc;
  wfft::init_lane(lc, lane);
  const int64_t b = blockIdx.x / tiles_per_clip;
  const int64_t t0 = (blockIdx.x - b * tiles_per_clip) * 32;
  const float* yb = y + b * ldy;
  // element 64 a + lane of z: even lanes carry p = frames (4 w, 4 w + 1), odd lanes q = frames (4 w + 2, 4 w + 3), sample
  // n = 32 a + lane / 2 of each
  const int64_t tf = t0 + 4 * w + 2 * (lane & 1);
  const int nl = lane >> 1;
  float2 v[16];
  if (t0 + 4 * w < T) {
    const int64_t sa = tf * (int64_t)hop - pad, sb = sa + hop;
    const bool hasa = tf < T, hasb = tf + 1 < T;
    const int64_t s_first = (t0 + 4 * w) * (int64_t)hop - pad;
    if (s_first >= 0 && s_first + 3 * (int64_t)hop + 512 <= L && t0 + 4 * w + 3 < T) {
#pragma unroll
      for (int a = 0; a < 16; ++a) {
        const float wv = win[32 * a + nl];
        v[a] = make_float2(yb[sa + 32 * a + nl] * wv, yb[sb + 32 * a + nl] * wv);
      }
    } else {
#pragma unroll
      for (int a = 0; a < 16; ++a) {
        const int64_t ia = sa + 32 * a + nl, ib = sb + 32 * a + nl;
        const float wv = win[32 * a + nl];
        v[a] = make_float2((hasa && ia >= 0 && ia < L) ? yb[ia] * wv : 0.f, (hasb && ib >= 0 && ib < L) ? yb[ib] * wv : 0.f);
      }
    }
  }
  // tables of the wave FFT from tw512[m] = W_1024^m (the third block of this frame length's twiddle argument)
  if (tid < 64) tw2l[(tid >> 4) * wfft::TW2_STRIDE + (tid & 15)] = tw512[(16 * (tid >> 4) * (tid & 15)) & 1023];
  for (int i = tid; i < wfft::TW1_COMPLEX; i += NW * 64) tw1l[i] = tw512[(i & 63) * ((i >> 6) + 1)];
  __syncthreads();
  {
    float* row0 = P + (4 * w) * PS;
    if (t0 + 4 * w < T) {
      float2 zk[2][4], zm[2][4], z512;
      wfft::cfft1024(v, lc, sc, tw1l, tw2l, lane, zk, zm, z512);
      // (za, zb, mb, ma) = (Z[k], Z[k + 512], Z[512 - k], Z[1024 - k]) -> the four frames' powers at one bin
      auto four = [&](float2 za, float2 zb, float2 mb, float2 ma, int bin) {
        const float sx = za.x + zb.x, sy = za.y + zb.y, dx = za.x - zb.x, dy = za.y - zb.y;
        const float mx = mb.x + ma.x, my = mb.y + ma.y, ex = mb.x - ma.x, ey = mb.y - ma.y;
        const float ax = sx + mx, ay = sy - my, bx = sx - mx, by = sy + my;
        const float cx = dx - ex, cy = dy + ey, gx = dx + ex, gy = dy - ey;
        const float pa = 0.0625f * fmaf(ax, ax, ay * ay), pb = 0.0625f * fmaf(bx, bx, by * by);
        const float pc = 0.0625f * fmaf(cx, cx, cy * cy), pd = 0.0625f * fmaf(gx, gx, gy * gy);
        row0[bin] = (power == 2) ? pa : sqrtf(pa);
        row0[PS + bin] = (power == 2) ? pb : sqrtf(pb);
        row0[2 * PS + bin] = (power == 2) ? pc : sqrtf(pc);
        row0[3 * PS + bin] = (power == 2) ? pd : sqrtf(pd);
      };
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int kb = wfft::bin_of(lane, j, 0);                     // 0 .. 127 (lane 0, unit 0: 0 -- overwritten below)
        four(zk[j][0], zk[j][2], zm[j][2], zm[j][0], kb);
        four(zk[j][1], zk[j][3], zm[j][3], zm[j][1], 256 - kb);
      }
      if (lane == 0) {
        // unit 0 of lane 0 holds Z at 0, 256, 128, 384 (zk) and 0, 768, 896, 640 (zm), Z[512] apart
        four(zk[0][0], z512, z512, zk[0][0], 0);
        four(zk[0][1], zm[0][1], zk[0][1], zm[0][1], 256);
        four(zk[0][2], zm[0][3], zk[0][3], zm[0][2], 128);
      }
      if (lane >= 1 && lane < PS - 256) {
#pragma unroll
        for (int f = 0; f < 4; ++f) row0[f * PS + 256 + lane] = 0.f;                       // [257, PS): zero weights
      }
    } else {
      for (int k = lane; k < 4 * PS; k += 64) row0[k] = 0.f;
    }
  }
  __syncthreads();
  // ---- projection: unit = (mel tile, half of the 32 frames)
  const int n_mt = (n_mels + 15) >> 4;
  const int n = lane & 15, kk = lane >> 4;
  for (int u = w; u < 2 * n_mt; u += NW) {
    const int mt = u >> 1, fh = u & 1;
    const float* arow = basis_p + (size_t)(16 * mt + n) * FP + 4 * kk;
    const float* brow = P + (16 * fh + n) * PS + 4 * kk;
    v4f acc = {0.f, 0.f, 0.f, 0.f}, acc2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 2
    for (int g = 0; g < FP / 16; ++g) {
      const float4 a4 = *reinterpret_cast<const float4*>(arow + 16 * g);
      const float4 b4 = *reinterpret_cast<const float4*>(brow + 16 * g);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.x, b4.x, acc, 0, 0, 0);
      acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.y, b4.y, acc2, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.z, b4.z, acc, 0, 0, 0);
      acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.w, b4.w, acc2, 0, 0, 0);
    }
    acc += acc2;
    const int64_t t = t0 + 16 * fh + n;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int m = 16 * mt + 4 * kk + i;
      if (m < n_mels && t < T) mel_out[(b * n_mels + m) * T + t] = acc[i];
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// n_fft = 256 on the wave FFT: EIGHT real frames per 1024-point complex transform.  Four complex sequences p_r = a_r + i b_r
// (r = 0 .. 3, two windowed 256-sample frames each) are interleaved as z[4 n + r] = p_r[n]:
//   Z[k + 256 d] = sum_r (-i)^(r d) W_1024^(r k) P_r[k]          (P_r: the 256-point transform of p_r)
// so that X_r = W^(r k) P_r[k] is the inverse 4-point transform of the lane's four values Z[k + 256 d] -- U_r / 4 with
// U_r = sum_d Z[k + 256 d] i^(r d) -- and the mirror bin 256 - k comes from the lane's four mirror values the same way
// (V_r = sum_d Z[256 - k + 256 d] i^(r d)): P_r[256 - k] = i^r W^(r k) V_r / 4.  The twiddle W^(r k) is common to
// P_r[k] and conj P_r[256 - k] and drops out of the POWERS:
//   64 |A_r[k]|^2 = |U_r + (-i)^r conj V_r|^2        64 |B_r[k]|^2 = |U_r - (-i)^r conj V_r|^2
// Lane 0's first unit (the self-mirrored groups) supplies bins 0 and 128.  A wave owns the frames 8 w .. 8 w + 7 of a
// 64-frame tile; its eight power rows alias its exchange scratch.  Tile form only.
struct W256Lds {
  static constexpr int FP = 144, PS = FP + 4;
  static constexpr int SCW = 8 * PS;                                  // per-wave scratch: 1184 floats >= 528 complex
  static constexpr int O_SC = 0;
  static constexpr int O_TW2 = O_SC + 8 * SCW;
  static constexpr int O_TW1 = O_TW2 + wfft::TW2_COMPLEX * 2;
  static constexpr int TOTAL = O_TW1 + wfft::TW1_COMPLEX * 2;
  static_assert(SCW >= 2 * wfft::SC_COMPLEX, "the exchange scratch must fit inside the eight aliased rows");
};

__global__ __launch_bounds__(512, SYG_P2_TILE_WAVES) void stft_mel_w256_kernel(
    const float* __restrict__ y, int64_t L, int64_t ldy, int hop, int pad, int64_t T, const float* __restrict__ win,
    const float2* __restrict__ tw1024, const float* __restrict__ basis_p, int n_mels, int power,
    float* __restrict__ mel_out, int tiles_per_clip) {
  typedef W256Lds LM;
  constexpr int NW = 8, FP = LM::FP, PS = LM::PS;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  float2* sc = reinterpret_cast<float2*>(lds + LM::O_SC + w * LM::SCW);
  float2* tw2l = reinterpret_cast<float2*>(lds + LM::O_TW2);
  float2* tw1l = reinterpret_cast<float2*>(lds + LM::O_TW1);
  float* P = lds + LM::O_SC;                       // row fi at fi * PS: rows 8 w .. 8 w + 7 inside wave w's scratch
  wfft::Lane lc;
  wfft::init_lane(lc, lane);
  const int64_t b = blockIdx.x / tiles_per_clip;
  const int64_t t0 = (blockIdx.x - b * tiles_per_clip) * 64;
  const float* yb = y + b * ldy;
  // element 64 a + lane of z: sequence r = lane & 3 = frames (8 w + 2 r, 8 w + 2 r + 1), sample n = 16 a + lane / 4
  const int64_t tf = t0 + 8 * w + 2 * (lane & 3);
  const int nl = lane >> 2;
  float2 v[16];
  if (t0 + 8 * w < T) {
    const int64_t sa = tf * (int64_t)hop - pad, sb = sa + hop;
    const bool hasa = tf < T, hasb = tf + 1 < T;
    const int64_t s_first = (t0 + 8 * w) * (int64_t)hop - pad;
    if (s_first >= 0 && s_first + 7 * (int64_t)hop + 256 <= L && t0 + 8 * w + 7 < T) {
#pragma unroll
      for (int a = 0; a < 16; ++a) {
        const float wv = win[16 * a + nl];
        v[a] = make_float2(yb[sa + 16 * a + nl] * wv, yb[sb + 16 * a + nl] * wv);
      }
    } else {
#pragma unroll
      for (int a = 0; a < 16; ++a) {
        const int64_t ia = sa + 16 * a + nl, ib = sb + 16 * a + nl;
        const float wv = win[16 * a + nl];
        v[a] = make_float2((hasa && ia >= 0 && ia < L) ? yb[ia] * wv : 0.f, (hasb && ib >= 0 && ib < L) ? yb[ib] * wv : 0.f);
      }
    }
  }
  if (tid < 64) tw2l[(tid >> 4) * wfft::TW2_STRIDE + (tid & 15)] = tw1024[(16 * (tid >> 4) * (tid & 15)) & 1023];
  for (int i = tid; i < wfft::TW1_COMPLEX; i += NW * 64) tw1l[i] = tw1024[(i & 63) * ((i >> 6) + 1)];
  __syncthreads();
  {
    float* row0 = P + (8 * w) * PS;
    if (t0 + 8 * w < T) {
      float2 zk[2][4], zm[2][4], z512;
      wfft::cfft1024(v, lc, sc, tw1l, tw2l, lane, zk, zm, z512);
      // z[d] = Z[k + 256 d], m[d] = Z[256 - k + 256 d]  ->  the eight frames' powers at bin k
      auto eight = [&](float2 z0, float2 z1, float2 z2, float2 z3, float2 m0, float2 m1, float2 m2, float2 m3, int bin) {
        float2 F[4], G[4];
        bfly4(z0, z1, z2, z3, F[0], F[1], F[2], F[3]);          // F[q] = sum_d z[d] (-i)^(d q): U_r = F[(4 - r) & 3]
        bfly4(m0, m1, m2, m3, G[0], G[1], G[2], G[3]);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float2 U = F[(4 - r) & 3], V = G[(4 - r) & 3];
          // (-i)^r conj V
          const float2 Tt = r == 0 ? make_float2(V.x, -V.y) : r == 1 ? make_float2(-V.y, -V.x)
                          : r == 2 ? make_float2(-V.x, V.y) : make_float2(V.y, V.x);
          const float ax = U.x + Tt.x, ay = U.y + Tt.y, bx = U.x - Tt.x, by = U.y - Tt.y;
          const float pa = 0.015625f * fmaf(ax, ax, ay * ay), pb = 0.015625f * fmaf(bx, bx, by * by);
          row0[(2 * r) * PS + bin] = (power == 2) ? pa : sqrtf(pa);
          row0[(2 * r + 1) * PS + bin] = (power == 2) ? pb : sqrtf(pb);
        }
      };
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int kb = wfft::bin_of(lane, j, 0);                     // 0 .. 127 (lane 0, unit 0: 0 -- overwritten below)
        eight(zk[j][0], zk[j][1], zk[j][2], zk[j][3], zm[j][3], zm[j][2], zm[j][1], zm[j][0], kb);
      }
      if (lane == 0) {
        // unit 0 of lane 0 holds Z at 0, 256, 128, 384 (zk) and 0, 768, 896, 640 (zm), Z[512] apart
        eight(zk[0][0], zk[0][1], z512, zm[0][1], zk[0][1], z512, zm[0][1], zk[0][0], 0);
        eight(zk[0][2], zk[0][3], zm[0][3], zm[0][2], zk[0][2], zk[0][3], zm[0][3], zm[0][2], 128);
      }
      if (lane >= 1 && lane < PS - 128) {
#pragma unroll
        for (int f = 0; f < 8; ++f) row0[f * PS + 128 + lane] = 0.f;                       // [129, PS): zero weights
      }
    } else {
      for (int k = lane; k < 8 * PS; k += 64) row0[k] = 0.f;
    }
  }
  __syncthreads();
  // ---- projection: unit = (mel tile, quarter of the 64 frames)
  const int n_mt = (n_mels + 15) >> 4;
  const int n = lane & 15, kk = lane >> 4;
  for (int u = w; u < 4 * n_mt; u += NW) {
    const int mt = u >> 2, fq = u & 3;
    const float* arow = basis_p + (size_t)(16 * mt + n) * FP + 4 * kk;
    const float* brow = P + (16 * fq + n) * PS + 4 * kk;
    v4f acc = {0.f, 0.f, 0.f, 0.f}, acc2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int g = 0; g < FP / 16; ++g) {
      const float4 a4 = *reinterpret_cast<const float4*>(arow + 16 * g);
      const float4 b4 = *reinterpret_cast<const float4*>(brow + 16 * g);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.x, b4.x, acc, 0, 0, 0);
      acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.y, b4.y, acc2, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.z, b4.z, acc, 0, 0, 0);
      acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.w, b4.w, acc2, 0, 0, 0);
    }
    acc += acc2;
    const int64_t t = t0 + 16 * fq + n;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int m = 16 * mt + 4 * kk + i;
      if (m < n_mels && t < T) mel_out[(b * n_mels + m) * T + t] = acc[i];
    }
  }
}

size_t w1024_lds_bytes(int n_mels, int tp) {
  const int n_mt = (n_mels + 15) / 16;
  return ((size_t)W1024Lds::O_PART + (size_t)2 * n_mt * 256 + (size_t)n_mels * tp + 8 + 2) * sizeof(float);
}

constexpr size_t LDS_LIMIT = 160 * 1024;
constexpr int NWAVES = 8;

size_t lds_bytes(int n_fft, int Fp, int n_mels, int tp) {
  return ((size_t)NWAVES * 2 * n_fft + 16 * (size_t)(Fp + 4) + 3 * (size_t)n_fft + (size_t)n_mels * tp + NWAVES + 2) * sizeof(float);
}

int check_args(const char* who, const float* y, int64_t B, int64_t L, int64_t ldy, int n_fft, int hop, int center, int64_t T,
               const float* window, const float* twiddle, const float* basis_p, int Fp, int n_mels, int power) {
  SYG_REQUIRE(y && window && twiddle && basis_p, "%s: null pointer argument", who);
  SYG_REQUIRE(B >= 1 && L >= 1 && ldy >= L, "%s: need B >= 1, L >= 1, ldy >= L", who);
  SYG_REQUIRE(n_fft >= 64 && n_fft <= 4096 && (n_fft & (n_fft - 1)) == 0, "%s: n_fft must be a power of two in [64, 4096] (got %d)",
              who, n_fft);
  SYG_REQUIRE(hop >= 1, "%s: hop must be >= 1", who);
  const int64_t Texp = center ? 1 + L / hop : (L >= n_fft ? 1 + (L - n_fft) / hop : 0);
  SYG_REQUIRE(T >= 1 && T == Texp, "%s: T=%lld does not match the framing rule (%lld)", who, (long long)T, (long long)Texp);
  SYG_REQUIRE(Fp % 16 == 0 && Fp >= n_fft / 2 + 1 && Fp < n_fft / 2 + 1 + 16, "%s: Fp must be 1 + n_fft/2 rounded up to a multiple of 16 (got %d)",
              who, Fp);
  SYG_REQUIRE(n_mels >= 1 && n_mels <= 256, "%s: n_mels must be in [1, 256]", who);
  SYG_REQUIRE(power == 1 || power == 2, "%s: power must be 1 or 2", who);
  SYG_REQUIRE(B * ((T + 15) / 16) < (int64_t)0x7fffffff, "%s: grid too large", who);
  SYG_REQUIRE(((uintptr_t)basis_p) % 16 == 0, "%s: the padded filterbank must be 16-byte aligned", who);
  return SYG_OK;
}

}  // namespace
}  // namespace syg

using namespace syg;

extern "C" int syg_stft_mel_pow2_f32(const float* y, int64_t B, int64_t L, int64_t ldy, int n_fft, int hop, int center,
                                     int64_t T, const float* window, const float* twiddle, const float* basis_p, int Fp,
                                     int n_mels, int power, float* mel_out, void* stream) {
  int rc = check_args("stft_mel_pow2", y, B, L, ldy, n_fft, hop, center, T, window, twiddle, basis_p, Fp, n_mels, power);
  if (rc) return rc;
  SYG_REQUIRE(mel_out, "stft_mel_pow2: null output");
  if (n_fft == 1024) {
    const size_t lds = w1024_lds_bytes(n_mels, 0);
    auto kern = stft_mel_w1024_kernel<false>;
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) { set_error("stft_mel_pow2: cannot reserve %zu B LDS: %s", lds, hipGetErrorString(e)); return SYG_E_LAUNCH; }
    const int tiles = (int)((T + 15) / 16);
    Pow2Mfcc mf;
    memset(&mf, 0, sizeof(mf));
    hipLaunchKernelGGL(kern, dim3((unsigned)(B * tiles)), dim3(512), lds, (hipStream_t)stream, y, L, ldy, hop,
                       center ? 512 : 0, T, window, (const float2*)twiddle, basis_p, n_mels, power, mel_out, tiles, mf);
    SYG_CHECK_LAUNCH("stft_mel_pow2");
    return SYG_OK;
  }
  if (n_fft == 256) {
    const size_t lds = (size_t)W256Lds::TOTAL * sizeof(float);
    hipError_t e = hipFuncSetAttribute((const void*)stft_mel_w256_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) { set_error("stft_mel_pow2: cannot reserve %zu B LDS: %s", lds, hipGetErrorString(e)); return SYG_E_LAUNCH; }
    const int tiles = (int)((T + 63) / 64);
    // (twiddle: [W_256^k (256) | W_128^k (128) | W_1024^k (1024)]: the wave FFT's tables come from the third block)
    hipLaunchKernelGGL(stft_mel_w256_kernel, dim3((unsigned)(B * tiles)), dim3(512), lds, (hipStream_t)stream, y, L, ldy, hop,
                       center ? 128 : 0, T, window, (const float2*)twiddle + 384, basis_p, n_mels, power, mel_out, tiles);
    SYG_CHECK_LAUNCH("stft_mel_pow2");
    return SYG_OK;
  }
  if (n_fft == 512) {
    const size_t lds = (size_t)W512Lds::TOTAL * sizeof(float);
    hipError_t e = hipFuncSetAttribute((const void*)stft_mel_w512_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) { set_error("stft_mel_pow2: cannot reserve %zu B LDS: %s", lds, hipGetErrorString(e)); return SYG_E_LAUNCH; }
    const int tiles = (int)((T + 31) / 32);
    // (twiddle: [W_512^k (512) | W_256^k (256) | W_1024^k (1024)]: the wave FFT's tables come from the third block)
    hipLaunchKernelGGL(stft_mel_w512_kernel, dim3((unsigned)(B * tiles)), dim3(512), lds, (hipStream_t)stream, y, L, ldy, hop,
                       center ? 256 : 0, T, window, (const float2*)twiddle + 768, basis_p, n_mels, power, mel_out, tiles);
    SYG_CHECK_LAUNCH("stft_mel_pow2");
    return SYG_OK;
  }
  const size_t lds = lds_bytes(n_fft, Fp, 0, 0);
  SYG_REQUIRE(lds <= LDS_LIMIT, "stft_mel_pow2: n_fft=%d needs %zu B of LDS", n_fft, lds);
  auto kern = stft_mel_pow2_kernel<NWAVES, false>;
  hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) { set_error("stft_mel_pow2: cannot reserve %zu B LDS: %s", lds, hipGetErrorString(e)); return SYG_E_LAUNCH; }
  const int tiles = (int)((T + 15) / 16);
  Pow2Mfcc mf;
  memset(&mf, 0, sizeof(mf));
  hipLaunchKernelGGL(kern, dim3((unsigned)(B * tiles)), dim3(NWAVES * 64), lds, (hipStream_t)stream, y, L, ldy, n_fft, hop,
                     center ? n_fft / 2 : 0, T, window, (const float2*)twiddle, basis_p, Fp, n_mels, power, mel_out, tiles, mf);
  SYG_CHECK_LAUNCH("stft_mel_pow2");
  return SYG_OK;
}

extern "C" int syg_stft_mfcc_pow2_fits(int n_fft, int n_mels, int64_t T, int n_mfcc) {
  if (n_fft < 64 || n_fft > 4096 || (n_fft & (n_fft - 1)) || n_mels < 1 || n_mels > 256 || T < 1 || T > (1 << 20) ||
      n_mfcc < 1 || n_mfcc > n_mels)
    return 0;
  const int Fp = ((n_fft / 2 + 1) + 15) / 16 * 16;
  const int tp = (int)((T + 15) / 16) * 16;
  if (n_fft == 1024) return w1024_lds_bytes(n_mels, tp) <= LDS_LIMIT ? 1 : 0;
  return lds_bytes(n_fft, Fp, n_mels, tp) <= LDS_LIMIT ? 1 : 0;
}

extern "C" int syg_stft_mfcc_pow2_f32(const float* y, int64_t B, int64_t L, int64_t ldy, int n_fft, int hop, int center,
                                      int64_t T, const float* window, const float* twiddle, const float* basis_p, int Fp,
                                      int n_mels, const float* dct, int n_mfcc, const float* lifter, float amin,
                                      float top_db, int ref_is_max, float ref_value, float* mel_out, float* mfcc_out,
                                      void* stream) {
  int rc = check_args("stft_mfcc_pow2", y, B, L, ldy, n_fft, hop, center, T, window, twiddle, basis_p, Fp, n_mels, 2);
  if (rc) return rc;
  SYG_REQUIRE(dct && mfcc_out, "stft_mfcc_pow2: null pointer argument");
  SYG_REQUIRE(n_mfcc >= 1 && n_mfcc <= n_mels, "stft_mfcc_pow2: need 1 <= n_mfcc <= n_mels");
  SYG_REQUIRE(amin >= 1.17549435e-38f, "stft_mfcc_pow2: amin must be strictly positive (a normal float)");
  SYG_REQUIRE(ref_is_max == 0 || ref_is_max == 1, "stft_mfcc_pow2: ref_is_max must be 0 or 1");
  SYG_REQUIRE(B <= 0x7fffffff, "stft_mfcc_pow2: too many clips");
  const int tiles = (int)((T + 15) / 16);
  Pow2Mfcc mf;
  mf.dct = dct; mf.lifter = lifter; mf.out = mfcc_out; mf.n_mfcc = n_mfcc; mf.ref_is_max = ref_is_max;
  mf.ref_value = ref_value; mf.amin = amin; mf.top_db = top_db; mf.tp = tiles * 16;
  if (n_fft == 1024) {
    const size_t lds = w1024_lds_bytes(n_mels, mf.tp);
    SYG_REQUIRE(lds <= LDS_LIMIT, "stft_mfcc_pow2: the clip's mel matrix (%d x %d) does not fit the LDS (%zu B > %zu B); use "
                "syg_stft_mel_pow2_f32 + syg_logmel_dct_f32", n_mels, mf.tp, lds, LDS_LIMIT);
    auto kern = stft_mel_w1024_kernel<true>;
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) { set_error("stft_mfcc_pow2: cannot reserve %zu B LDS: %s", lds, hipGetErrorString(e)); return SYG_E_LAUNCH; }
    hipLaunchKernelGGL(kern, dim3((unsigned)B), dim3(512), lds, (hipStream_t)stream, y, L, ldy, hop, center ? 512 : 0, T,
                       window, (const float2*)twiddle, basis_p, n_mels, 2, mel_out, tiles, mf);
    SYG_CHECK_LAUNCH("stft_mfcc_pow2");
    return SYG_OK;
  }
  const size_t lds = lds_bytes(n_fft, Fp, n_mels, mf.tp);
  SYG_REQUIRE(lds <= LDS_LIMIT, "stft_mfcc_pow2: the clip's mel matrix (%d x %d) does not fit the LDS (%zu B > %zu B); use "
              "syg_stft_mel_pow2_f32 + syg_logmel_dct_f32", n_mels, mf.tp, lds, LDS_LIMIT);
  auto kern = stft_mel_pow2_kernel<NWAVES, true>;
  hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) { set_error("stft_mfcc_pow2: cannot reserve %zu B LDS: %s", lds, hipGetErrorString(e)); return SYG_E_LAUNCH; }
  hipLaunchKernelGGL(kern, dim3((unsigned)B), dim3(NWAVES * 64), lds, (hipStream_t)stream, y, L, ldy, n_fft, hop,
                     center ? n_fft / 2 : 0, T, window, (const float2*)twiddle, basis_p, Fp, n_mels, 2, mel_out, tiles, mf);
  SYG_CHECK_LAUNCH("stft_mfcc_pow2");
  return SYG_OK;
}
