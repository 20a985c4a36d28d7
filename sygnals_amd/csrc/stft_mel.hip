// Fused STFT(2048) -> power -> mel kernel for gfx950 (MI355X).
//
// One workgroup (8 waves) owns a tile of 16 consecutive frames of one clip.
//   phase 1  each wave computes two frames: coalesced float2 loads of the (overlapped)
//            frame straight from HBM/L2, Hann (any) window, a 1024-point complex FFT
//            held 16 points per lane (radix 16 x 16 x 4, two LDS exchanges, XOR-swizzled
//            so both exchanges are bank-conflict free on write and on exchange-1 read),
//            real-FFT split on mirror pairs, |X|^2 written as one 1025-float row in LDS;
//   phase 2  the 16 power rows are the B operand of v_mfma_f32_16x16x4_f32: the mel
//            filterbank is stored block-sparse (per 16-mel tile only its non-zero bin
//            range) and split over the 8 waves; partial 16x16 tiles are combined in a
//            fixed order (deterministic) and stored as mel[b, m, t];
//   phase 2b (optional) per-frame spectral statistics and contrast peak/valley means
//            are reduced from the same LDS rows with wave-level scans.
// Nothing but the input samples and the mel / stats outputs touches HBM.
//
// Reference behaviour reproduced: librosa.stft (center zero padding, periodic window,
// rfft) -> np.abs -> **2 -> melspectrogram, as called from
// sygnals/core/features/manager.py:184-187, 198, 219-222; per-frame statistics follow
// sygnals/core/features/frequency_domain.py:24-386.
#include "common.h"

namespace syg {
namespace {

constexpr int NFFT = 2048;
constexpr int MC = 1024;        // complex points per frame
constexpr int NBIN = 1025;
constexpr int TILE_T = 16;      // frames per workgroup
constexpr int WAVES = 8;
constexpr int NTHREADS = WAVES * 64;
constexpr int P_STRIDE = 1090;  // == 2 (mod 32): conflict-free MFMA B-operand reads; rows are skewed, see ppos()
constexpr int P_FLOATS = TILE_T * P_STRIDE + 16;
constexpr int XPL = 260;         // exchange-2 plane stride (complex): 256 + 4 skew
constexpr int SC_COMPLEX = 4 * XPL;  // per-wave exchange scratch (1040 complex, 8320 B)
constexpr int SCRATCH_FLOATS = WAVES * SC_COMPLEX * 2;
constexpr int SLAB_FLOATS = WAVES * 256;
constexpr int TW2_STRIDE = 18;         // complex entries per lane class (16 + 2 pad: distinct banks)
constexpr int TW2_FLOATS = 4 * TW2_STRIDE * 2;  // W_64^(b'*c') table
constexpr int TW1_FLOATS = 15 * 64 * 2;  // W_1024^(lane*c) table, [15][64] complex

struct MelPlan {
  int n_tiles;
  int tile[WAVES];
  int k0[WAVES];
  int nsteps[WAVES];
  int woff[WAVES];
};

struct ContrastPlan {
  int n_rows;
  int lo[SYG_MAX_BANDS];
  int hi[SYG_MAX_BANDS];
  int k[SYG_MAX_BANDS];
};

typedef float v4f __attribute__((ext_vector_type(4)));

// LDS complex-index swizzles (validated by tools/wave_fft_model.py)
__device__ __forceinline__ int swz1(int c, int b) { return c * 64 + (b ^ (4 * (c & 7))); }
// exchange 2 is planar: element b' of group (c, c') lives at b'*XPL + 16*c' + (c ^ c')  -- conflict free for
// the ds_write_b64 of pass 2 and for the four ds_read_b64 per group of pass 3
__device__ __forceinline__ int swz2g(int c, int cp) { return cp * 16 + (c ^ cp); }
// position of bin k inside an LDS power row: one pad word every 16 bins turns the stride-16 bin pattern
// of the pass-3 output into a conflict-free store while 4-aligned bin quads stay contiguous for the MFMA
__device__ __forceinline__ int ppos(int k) { return k + (k >> 4); }

// unit u (0..127) -> primary group (c, c') and mirror group (cm, cm')
__device__ __forceinline__ void unit_groups(int u, int& c, int& cp, int& cm, int& cmp) {
  if (u < 112) { c = 1 + (u >> 4); cp = u & 15; cm = 16 - c; cmp = 15 - cp; }
  else if (u < 120) { c = 8; cp = u - 112; cm = 8; cmp = 15 - cp; }
  else if (u < 127) { c = 0; cp = u - 119; cm = 0; cmp = 16 - cp; }
  else { c = 0; cp = 0; cm = 0; cmp = 8; }
}

struct LaneConst {
  float2 twp[2][4]; // W_2048^k for the 4 mirror pairs of each unit
  int kk[2][4];     // output bin k of each pair (its mirror is 1024 - k)
  int pk[2][4];     // ppos(k): position of bin k in an LDS power row
  int pm[2][4];     // ppos(1024 - k)
  int g0[2], g1[2]; // LDS complex index of primary / mirror group of each unit
};

__device__ __forceinline__ void init_lane_const(LaneConst& lc, int lane, const float2* __restrict__ twid) {
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    int u = lane + 64 * j, c, cp, cm, cmp;
    unit_groups(u, c, cp, cm, cmp);
    lc.g0[j] = swz2g(c, cp);
    lc.g1[j] = swz2g(cm, cmp);
    const int kb = c + 16 * cp;
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      int k = kb + 256 * d;
      if (u == 127) k = (d == 0) ? 0 : (d == 1) ? 256 : (d == 2) ? 128 : 384;
      lc.kk[j][d] = k;
      lc.pk[j][d] = ppos(k);
      lc.pm[j][d] = ppos(MC - k);
      lc.twp[j][d] = twid[k];
    }
  }
}

// 1024-point complex forward FFT of the windowed frame + real split.
// v[a] holds z[64a + lane] on entry.  On exit Xa/Xb hold the 16 (+1) spectrum values:
// pair (j, d): X[kk[j][d]] -> xs[j][d], X[1024 - kk[j][d]] -> xm[j][d]; lane 63 also
// returns X[512] in x512.
__device__ __forceinline__ void wave_rfft2048(float2 (&v)[16], const LaneConst& lc, float2* __restrict__ sc,
                                             const float2* __restrict__ tw1l, const float2* __restrict__ tw2l,
                                             int lane, float2 (&xs)[2][4], float2 (&xm)[2][4], float2& x512) {
  // ---- pass 1: radix-16 over a (stride 64), twiddle W_1024^(b*c)
  dft16(v);
#pragma unroll
  for (int c = 1; c < 16; ++c) v[c] = cmul(v[c], tw1l[(c - 1) * 64 + lane]);   // W_1024^(lane*c), LDS table
#pragma unroll
  for (int c = 0; c < 16; ++c) sc[swz1(c, lane)] = v[c];
  // ---- pass 2: lane = (c = lane>>2, b' = lane&3); radix-16 over a'
  const int cl = lane >> 2, bp = lane & 3;
#pragma unroll
  for (int a = 0; a < 16; ++a) v[a] = sc[swz1(cl, 4 * a + bp)];
  dft16(v);
  {
    // W_64^(b'*c') from a 64-entry LDS table (4 lane classes): two twiddles per 16-byte read
    const float4* t4 = reinterpret_cast<const float4*>(tw2l + bp * TW2_STRIDE);
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      const float4 tt = t4[m];
      if (m > 0) v[2 * m] = cmul(v[2 * m], make_float2(tt.x, tt.y));
      v[2 * m + 1] = cmul(v[2 * m + 1], make_float2(tt.z, tt.w));
    }
  }
  {
    const int base = bp * XPL;
#pragma unroll
    for (int cp = 0; cp < 16; ++cp) sc[base + cp * 16 + (cl ^ cp)] = v[cp];
  }
  // ---- pass 3: radix-4 over b' for two mirror-paired units, then the real split
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const float2* p0 = sc + lc.g0[j];
    const float2* p1 = sc + lc.g1[j];
    float2 G[4], H[4];
    bfly4(p0[0], p0[XPL], p0[2 * XPL], p0[3 * XPL], G[0], G[1], G[2], G[3]);
    bfly4(p1[0], p1[XPL], p1[2 * XPL], p1[3 * XPL], H[0], H[1], H[2], H[3]);
    float2 zk[4] = {G[0], G[1], G[2], G[3]};
    float2 zm[4] = {H[3], H[2], H[1], H[0]};
    if (j == 1) {
      // unit 127 (lane 63) pairs the self-mirrored groups (0,0) and (0,8) differently
      const bool sp = (lane == 63);
      x512 = G[2];
      zk[2] = sp ? H[0] : zk[2];
      zk[3] = sp ? H[1] : zk[3];
      zm[0] = sp ? G[0] : zm[0];
      zm[1] = sp ? G[3] : zm[1];
      zm[2] = sp ? H[3] : zm[2];
      zm[3] = sp ? H[2] : zm[3];
    }
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      // E2 = zk + conj(zm), O2 = -i (zk - conj(zm));  X[k] = (E2 + w O2)/2, X[1024-k] = conj(E2 - w O2)/2
      float2 E = make_float2(zk[d].x + zm[d].x, zk[d].y - zm[d].y);
      float2 O = make_float2(zk[d].y + zm[d].y, zm[d].x - zk[d].x);
      float2 wO = cmul(lc.twp[j][d], O);
      xs[j][d] = make_float2(0.5f * (E.x + wO.x), 0.5f * (E.y + wO.y));
      xm[j][d] = make_float2(0.5f * (E.x - wO.x), -0.5f * (E.y - wO.y));
    }
  }
  x512 = make_float2(x512.x, -x512.y);
}

// Raw (unwindowed) samples of one frame: element n = 64a + lane of the packed complex frame covers
// samples s0 + 2n, s0 + 2n + 1; samples outside [0, L) are the zero padding of center=True.
template <bool VEC2>
__device__ __forceinline__ void load_frame_raw(float2 (&r)[16], const float* __restrict__ yb, int64_t L, int64_t s0,
                                               int lane, bool valid) {
  const bool interior = valid && (s0 >= 0) && (s0 + NFFT <= L);
  if (interior) {
#pragma unroll
    for (int a = 0; a < 16; ++a) {
      const int n = 64 * a + lane;
      if (VEC2) r[a] = *reinterpret_cast<const float2*>(yb + s0 + 2 * n);
      else r[a] = make_float2(yb[s0 + 2 * n], yb[s0 + 2 * n + 1]);
    }
  } else {
#pragma unroll
    for (int a = 0; a < 16; ++a) {
      const int64_t s = s0 + 2 * (64 * a + lane);
      const float x0 = (valid && s >= 0 && s < L) ? yb[s] : 0.f;
      const float x1 = (valid && s + 1 >= 0 && s + 1 < L) ? yb[s + 1] : 0.f;
      r[a] = make_float2(x0, x1);
    }
  }
}

// ----------------------------------------------------------------------------------
// per-row statistics from an LDS power row (one wave per row)
// ----------------------------------------------------------------------------------
__device__ __forceinline__ float wave_excl_scan(float v, int lane) {
  float s = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    float t = __shfl_up(s, o, 64);
    if (lane >= o) s += t;
  }
  return s - v;
}

__device__ void row_stats(const float* __restrict__ prow, int lane, float binhz, float roll_percent, float bw_p,
                          float* __restrict__ out, int64_t ostride) {
  // lane owns the contiguous bins [17*lane, 17*lane+17) (64*17 = 1088 >= 1025)
  constexpr int CH = 17;
  const int b0 = lane * CH;
  float msum = 0.f, fsum = 0.f, psum = 0.f, lsum = 0.f, mmax = -1.f;
  int amax = 0;
  float pl[CH];
#pragma unroll
  for (int i = 0; i < CH; ++i) {
    const int k = b0 + i;
    const float p = (k < NBIN) ? prow[ppos(k)] : 0.f;
    pl[i] = p;
    const float m = sqrtf(p);
    if (k < NBIN) {
      msum += m;
      fsum = fmaf(m, (float)k, fsum);
      psum += p;
      lsum += logf(m + 2.220446049250313e-16f);
      if (m > mmax) { mmax = m; amax = k; }
    }
  }
  const float tot_m = wave_sum(msum), tot_f = wave_sum(fsum), tot_p = wave_sum(psum), tot_l = wave_sum(lsum);
  // argmax (first occurrence)
  float gm = wave_max(mmax);
  int cand = (mmax == gm) ? amax : 0x7fffffff;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) cand = min(cand, __shfl_xor(cand, o, 64));
  const float EPS = 2.220446049250313e-16f;
  const bool live = tot_m >= EPS;
  const float cen_bin = live ? tot_f / tot_m : 0.f;
  // bandwidth: (sum m |f - c|^p / sum m)^(1/p), in bins then scaled
  float dsum = 0.f;
#pragma unroll
  for (int i = 0; i < CH; ++i) {
    const int k = b0 + i;
    if (k < NBIN) {
      const float d = fabsf((float)k - cen_bin) * binhz;
      const float m = sqrtf(pl[i]);
      dsum = fmaf(m, (bw_p == 2.f) ? d * d : powf(d, bw_p), dsum);
    }
  }
  const float tot_d = wave_sum(dsum);
  // rolloff: first bin with cumsum(power) >= roll * total
  const float excl = wave_excl_scan(psum, lane);
  const float thr = roll_percent * tot_p;
  int rb = 0x7fffffff;
  float margin = 3.4e38f;
  {
    float c = excl;
#pragma unroll
    for (int i = 0; i < CH; ++i) {
      const int k = b0 + i;
      if (k < NBIN) {
        const float cprev = c;
        c += pl[i];
        if (c >= thr && rb == 0x7fffffff) {
          rb = k;
          margin = fminf(c - thr, (k > 0) ? thr - cprev : 3.4e38f);
        }
      }
    }
  }
  int rbmin = rb;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) rbmin = min(rbmin, __shfl_xor(rbmin, o, 64));
  float mg = (rb == rbmin && rb != 0x7fffffff) ? margin : 3.4e38f;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mg = fminf(mg, __shfl_xor(mg, o, 64));
  if (rbmin == 0x7fffffff || tot_p < EPS) rbmin = NBIN - 1;
  if (lane == 0) {
    const float am = tot_m / (float)NBIN;
    float flat = 0.f;
    if (am >= EPS) flat = fminf(fmaxf(expf(tot_l / (float)NBIN) / am, 0.f), 1.f);
    float bw = 0.f;
    if (live) bw = (bw_p == 2.f) ? sqrtf(fmaxf(tot_d / tot_m, 0.f)) : powf(fmaxf(tot_d / tot_m, 0.f), 1.f / bw_p);
    out[SYG_STAT_CENTROID * ostride] = cen_bin * binhz;
    out[SYG_STAT_BANDWIDTH * ostride] = bw;
    out[SYG_STAT_FLATNESS * ostride] = flat;
    out[SYG_STAT_ROLLOFF_BIN * ostride] = (float)rbmin;
    out[SYG_STAT_DOMINANT_BIN * ostride] = (float)cand;
    out[SYG_STAT_MAG_SUM * ostride] = tot_m;
    out[SYG_STAT_POWER_SUM * ostride] = tot_p;
    out[SYG_STAT_ROLLOFF_MARGIN * ostride] = (tot_p > 0.f) ? mg / tot_p : 0.f;
  }
}

// mean of the k smallest and k largest magnitudes of bins [lo, hi) of one LDS power row.
// Values are non-negative so their float bit patterns order like unsigned integers:
// a 32-step bitwise radix select finds the k-th order statistic exactly, then the tail
// sum is closed with the tie count (identical to sorting, as librosa does).
__device__ void row_contrast(const float* __restrict__ prow, int lane, int lo, int hi, int k, float& peak,
                             float& valley) {
  const int n = hi - lo;
  // ---- k-th smallest (1-based k) threshold on power bits
  auto kth = [&](int kk, bool largest) -> uint32_t {
    uint32_t prefix = 0;
    int remaining = kk;
    for (int bit = 31; bit >= 0; --bit) {
      const uint32_t mask = ~((1u << bit) - 1u);        // bits above and including `bit`
      const uint32_t want = largest ? (prefix | (1u << bit)) : prefix;
      int cnt = 0;
      for (int i = lane; i < n; i += 64) {
        const uint32_t u = __float_as_uint(prow[ppos(lo + i)]);
        cnt += ((u & mask) == want) ? 1 : 0;
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o, 64);
      if (largest) {
        if (cnt >= remaining) prefix |= (1u << bit); else remaining -= cnt;
      } else {
        if (cnt >= remaining) { /* stay in the 0 branch */ } else { remaining -= cnt; prefix |= (1u << bit); }
      }
    }
    return prefix;
  };
  const uint32_t tlo = kth(k, false), thi = kth(k, true);
  float slo = 0.f, shi = 0.f;
  int clo = 0, chi = 0;
  for (int i = lane; i < n; i += 64) {
    const float p = prow[ppos(lo + i)];
    const uint32_t u = __float_as_uint(p);
    const float m = sqrtf(p);
    if (u < tlo) { slo += m; ++clo; }
    if (u > thi) { shi += m; ++chi; }
  }
  slo = wave_sum(slo); shi = wave_sum(shi);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { clo += __shfl_xor(clo, o, 64); chi += __shfl_xor(chi, o, 64); }
  valley = (slo + (float)(k - clo) * sqrtf(__uint_as_float(tlo))) / (float)k;
  peak = (shi + (float)(k - chi) * sqrtf(__uint_as_float(thi))) / (float)k;
}

// ----------------------------------------------------------------------------------
// MODE 0: mel only   MODE 1: mel + per-frame statistics / contrast   MODE 2: complex STFT output
template <bool VEC2, int MODE>
__global__ __launch_bounds__(NTHREADS) void stft2048_kernel(
    const float* __restrict__ y, int64_t L, int64_t ldy, int hop, int pad, int64_t T, int tiles_per_clip,
    int64_t total_tiles, int tiles_per_wg, const float2* __restrict__ win2, const float2* __restrict__ twid,
    const float* __restrict__ wpacked, MelPlan plan, int n_mels, float* __restrict__ mel_out, float binhz,
    float roll_percent, float bw_p, float* __restrict__ stats_out, ContrastPlan cplan,
    float* __restrict__ contrast_out, float2* __restrict__ cout) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float2* scratch = reinterpret_cast<float2*>(lds);                 // [WAVES][1024] complex
  float* Pbuf = lds + SCRATCH_FLOATS;                               // [16][P_STRIDE] (+16)
  float* slab = Pbuf + P_FLOATS;                                    // [WAVES][16][16]
  float2* tw2l = reinterpret_cast<float2*>(slab + SLAB_FLOATS);     // [4][16] complex
  float2* tw1l = tw2l + TW2_FLOATS / 2;                             // [15][64] complex
  int* cpl = reinterpret_cast<int*>(slab + SLAB_FLOATS + TW2_FLOATS + TW1_FLOATS);  // contrast plan

  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  constexpr int FPW = TILE_T / WAVES;                               // frames per wave per tile (2)
  constexpr bool COMPLEX_OUT = (MODE == 2);

  // persistent workgroup: a contiguous chunk of tiles (consecutive tiles of a clip share 3/4 of
  // their samples, so the re-reads of the frame overlap stay in this CU's L1 / this XCD's L2)
  const int64_t tile_begin = (int64_t)blockIdx.x * tiles_per_wg;
  const int64_t tile_end = (tile_begin + tiles_per_wg < total_tiles) ? tile_begin + tiles_per_wg : total_tiles;
  if (tile_begin >= tile_end) return;

  LaneConst lc;
  init_lane_const(lc, lane, twid);
  float2* sc = scratch + w * SC_COMPLEX;
  if (tid < 64) tw2l[(tid >> 4) * TW2_STRIDE + (tid & 15)] = twid[32 * (tid >> 4) * (tid & 15)];
  for (int i = tid; i < 15 * 64; i += NTHREADS) tw1l[i] = twid[2 * (i & 63) * ((i >> 6) + 1)];
  __syncthreads();

  int ns = 0, woff = 0, k0 = 0;
  if (!COMPLEX_OUT) {
    // zero the row pads / slack once (read by the MFMA B operand against zero weights)
    // pad words of the skewed rows, the row tails and the slack are read against zero weights: they must
    // hold finite values, so the whole buffer (and the slab behind it) is cleared once
    for (int i = tid; i < P_FLOATS + SLAB_FLOATS; i += NTHREADS) Pbuf[i] = 0.f;
#pragma unroll
    for (int r = 0; r < SYG_MAX_BANDS; ++r)
      if (tid == r) { cpl[r] = cplan.lo[r]; cpl[SYG_MAX_BANDS + r] = cplan.hi[r]; cpl[2 * SYG_MAX_BANDS + r] = cplan.k[r]; }
#pragma unroll
    for (int ww = 0; ww < WAVES; ++ww)
      if (w == ww) { ns = plan.nsteps[ww]; woff = plan.woff[ww]; k0 = plan.k0[ww]; }
  }

  auto tile_coords = [&](int64_t tile, int64_t& b, int64_t& t0) {
    const uint32_t q = (uint32_t)tile / (uint32_t)tiles_per_clip;   // total_tiles < 2^31 (checked on the host)
    b = q;
    t0 = (int64_t)((uint32_t)tile - q * (uint32_t)tiles_per_clip) * TILE_T;
  };

  // software pipeline, one frame deep: while frame q of this wave's sequence (frames 2w, 2w+1 of
  // tile 0, then of tile 1, ...) is transformed, the raw samples of frame q+1 are in flight
  float2 raw[16];
  {
    int64_t b, t0;
    tile_coords(tile_begin, b, t0);
    const int64_t t = t0 + w * FPW;
    load_frame_raw<VEC2>(raw, y + b * ldy, L, t * (int64_t)hop - pad, lane, t < T);
  }

#pragma unroll 1
  for (int64_t tile = tile_begin; tile < tile_end; ++tile) {
    int64_t b, t0;
    tile_coords(tile, b, t0);
#pragma unroll 1
    for (int j = 0; j < FPW; ++j) {
      const int fs = w * FPW + j;
      const int64_t t = t0 + fs;
      float* prow = Pbuf + fs * P_STRIDE;
      float2 v[16];
#pragma unroll
      for (int a = 0; a < 16; ++a) {
        const float2 wv = win2[64 * a + lane];
        v[a] = make_float2(raw[a].x * wv.x, raw[a].y * wv.y);
      }
      // issue the loads of the next frame of this tile (the first frame of the NEXT tile is requested
      // after the mel phase, when the A-operand registers are free again)
      if (j < FPW - 1 || COMPLEX_OUT) {
        int64_t nb = b, nt = t + 1;
        bool more = true;
        if (j == FPW - 1) {
          more = tile + 1 < tile_end;
          if (more) {
            int64_t nt0;
            tile_coords(tile + 1, nb, nt0);
            nt = nt0 + w * FPW;
          }
        }
        load_frame_raw<VEC2>(raw, y + nb * ldy, L, nt * (int64_t)hop - pad, lane, more && nt < T);
      }
      if (t < T) {
        float2 xs[2][4], xm[2][4], x512;
        wave_rfft2048(v, lc, sc, tw1l, tw2l, lane, xs, xm, x512);
        if (COMPLEX_OUT) {
          float2* o = cout + (b * T + t) * NBIN;
#pragma unroll
          for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int d = 0; d < 4; ++d) {
              o[lc.kk[u][d]] = xs[u][d];
              o[MC - lc.kk[u][d]] = xm[u][d];
            }
          if (lane == 63) o[512] = x512;
        } else {
#pragma unroll
          for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int d = 0; d < 4; ++d) {
              prow[lc.pk[u][d]] = fmaf(xs[u][d].x, xs[u][d].x, xs[u][d].y * xs[u][d].y);
              prow[lc.pm[u][d]] = fmaf(xm[u][d].x, xm[u][d].x, xm[u][d].y * xm[u][d].y);
            }
          if (lane == 63) prow[ppos(512)] = fmaf(x512.x, x512.x, x512.y * x512.y);
        }
      } else if (!COMPLEX_OUT) {
        for (int k = lane; k < P_STRIDE; k += 64) prow[k] = 0.f;
      }
    }
    if (COMPLEX_OUT) continue;

    // ---- phase 2: block-sparse mel projection on the matrix cores.  The wave's A operands (its
    // slice of the packed filterbank, L2 resident) are requested BEFORE the barrier so that their
    // round trip overlaps the wait for the slowest FFT wave; they sit in the registers the FFT freed.
    __syncthreads();
    {
      const int f = lane & 15, g = lane >> 4;
      // k0 is a multiple of 16, so ppos(k0 + 4i) = ppos(k0) + 4i + (i >> 2): one base register and
      // compile-time offsets.  The A operands are packed four steps per lane (one 16-byte load feeds
      // four MFMAs); the step count of a segment is a multiple of 4 (zero-weight padding).
      const float* pq = Pbuf + f * P_STRIDE + g + ppos(k0);
      const float4* wp4 = reinterpret_cast<const float4*>(wpacked) + (int64_t)(woff >> 2) * 64 + lane;
      v4f acc = {0.f, 0.f, 0.f, 0.f};
      const int ng = ns >> 2;
#pragma unroll 4
      for (int q = 0; q < ng; ++q) {
        const float4 a4 = wp4[q * 64];
        const float* pb = pq + 17 * q;          // 16 bins + 1 pad word per group of four steps
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.x, pb[0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.y, pb[4], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.z, pb[8], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.w, pb[12], acc, 0, 0, 0);
      }
      float* sl = slab + w * 256;
#pragma unroll
      for (int r = 0; r < 4; ++r) sl[(4 * g + r) * 16 + f] = acc[r];
    }
    if (tile + 1 < tile_end) {   // first frame of the next tile: in flight across the slab barrier and the reduce
      int64_t nb, nt0;
      tile_coords(tile + 1, nb, nt0);
      const int64_t nt = nt0 + w * FPW;
      load_frame_raw<VEC2>(raw, y + nb * ldy, L, nt * (int64_t)hop - pad, lane, nt < T);
    }
    __syncthreads();
    for (int i = tid; i < plan.n_tiles * 256; i += NTHREADS) {
      const int mt = i >> 8, m = (i >> 4) & 15, tt = i & 15;
      float sum = 0.f;
#pragma unroll
      for (int ww = 0; ww < WAVES; ++ww)
        if (plan.tile[ww] == mt) sum += slab[ww * 256 + m * 16 + tt];
      const int mel = mt * 16 + m;
      if (mel < n_mels && t0 + tt < T) mel_out[(b * n_mels + mel) * T + t0 + tt] = sum;
    }

    // ---- phase 2b: per-frame statistics / contrast means from the same LDS rows
    if (MODE == 1 && (stats_out != nullptr || contrast_out != nullptr)) {
#pragma unroll 1
      for (int j = 0; j < FPW; ++j) {
        const int fs = w * FPW + j;
        const int64_t t = t0 + fs;
        if (t >= T) continue;
        const float* prow = Pbuf + fs * P_STRIDE;
        if (stats_out != nullptr)
          row_stats(prow, lane, binhz, roll_percent, bw_p, stats_out + (b * SYG_NSTAT) * T + t, T);
        if (contrast_out != nullptr) {
          for (int r = 0; r < cplan.n_rows; ++r) {
            float pk, vl;
            row_contrast(prow, lane, cpl[r], cpl[SYG_MAX_BANDS + r], cpl[2 * SYG_MAX_BANDS + r], pk, vl);
            if (lane == 0) {
              contrast_out[((b * 2 + 0) * cplan.n_rows + r) * T + t] = pk;
              contrast_out[((b * 2 + 1) * cplan.n_rows + r) * T + t] = vl;
            }
          }
        }
      }
      __syncthreads();   // the rows are overwritten by the next tile's FFT phase
    }
  }
}

constexpr size_t LDS_BYTES_MEL = (size_t)(SCRATCH_FLOATS + P_FLOATS + SLAB_FLOATS + TW2_FLOATS + TW1_FLOATS + 3 * SYG_MAX_BANDS) * sizeof(float);
constexpr size_t LDS_BYTES_C2C = LDS_BYTES_MEL;   // same carve-up (the mel rows are simply unused)

// One workgroup per CU (the LDS footprint admits one); each takes a contiguous chunk of tiles.
void persistent_grid(int64_t total_tiles, int& wgs, int& per) {
  static int n_cu = 0;
  if (n_cu == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess)
      n_cu = prop.multiProcessorCount;
    if (n_cu <= 0) n_cu = 256;
  }
  int64_t p = (total_tiles + n_cu - 1) / n_cu;
  if (p < 1) p = 1;
  per = (int)p;
  wgs = (int)((total_tiles + p - 1) / p);
}

int check_common(const float* y, int64_t B, int64_t L, int64_t ldy, int hop, int center, int64_t T,
                 const float* window, const float* twiddle) {
  SYG_REQUIRE(y && window && twiddle, "stft2048: null pointer argument");
  SYG_REQUIRE(B >= 1 && L >= 1 && ldy >= L, "stft2048: need B >= 1, L >= 1, ldy >= L (B=%lld L=%lld ldy=%lld)",
              (long long)B, (long long)L, (long long)ldy);
  SYG_REQUIRE(hop >= 1, "stft2048: hop must be >= 1 (got %d)", hop);
  const int64_t Texp = center ? 1 + L / hop : (L >= NFFT ? 1 + (L - NFFT) / hop : 0);
  SYG_REQUIRE(T >= 1 && T == Texp, "stft2048: T=%lld does not match the framing rule (%lld)", (long long)T,
              (long long)Texp);
  SYG_REQUIRE(B * ((T + TILE_T - 1) / TILE_T) < (int64_t)0x7fffffff, "stft2048: grid too large");
  return SYG_OK;
}

}  // namespace
}  // namespace syg

using namespace syg;

extern "C" int syg_stft2048_mel_f32(const float* y, int64_t B, int64_t L, int64_t ldy, int hop, int center,
                                    int64_t T, const float* window, const float* twiddle, const float* wpacked,
                                    const int32_t* plan_host, int n_mels, float* mel_out, float sr,
                                    float roll_percent, float bw_p, float* stats_out, const int32_t* cplan_host,
                                    float* contrast_out, void* stream) {
  int rc = check_common(y, B, L, ldy, hop, center, T, window, twiddle);
  if (rc) return rc;
  SYG_REQUIRE(wpacked && plan_host && mel_out, "stft2048_mel: null pointer argument");
  SYG_REQUIRE(n_mels >= 1 && n_mels <= 16 * WAVES, "stft2048_mel: n_mels must be in [1, %d] (got %d)", 16 * WAVES,
              n_mels);
  MelPlan plan;
  plan.n_tiles = plan_host[0];
  SYG_REQUIRE(plan.n_tiles == (n_mels + 15) / 16, "stft2048_mel: plan has %d tiles, n_mels=%d needs %d",
              plan.n_tiles, n_mels, (n_mels + 15) / 16);
  for (int w = 0; w < WAVES; ++w) {
    plan.tile[w] = plan_host[1 + w];
    plan.k0[w] = plan_host[1 + WAVES + w];
    plan.nsteps[w] = plan_host[1 + 2 * WAVES + w];
    plan.woff[w] = plan_host[1 + 3 * WAVES + w];
    SYG_REQUIRE(plan.tile[w] >= -1 && plan.tile[w] < plan.n_tiles, "stft2048_mel: bad tile in plan");
    SYG_REQUIRE(plan.nsteps[w] >= 0 && plan.k0[w] >= 0 && plan.k0[w] + 4 * plan.nsteps[w] <= NBIN + 15 &&
                    plan.woff[w] >= 0 && plan.k0[w] % 16 == 0 && plan.nsteps[w] % 4 == 0 && plan.woff[w] % 4 == 0,
                "stft2048_mel: plan segment %d out of range (k0=%d nsteps=%d)", w, plan.k0[w], plan.nsteps[w]);
  }
  ContrastPlan cp;
  cp.n_rows = 0;
  if (contrast_out) {
    SYG_REQUIRE(cplan_host, "stft2048_mel: contrast_out given without cplan_host");
    cp.n_rows = cplan_host[0];
    SYG_REQUIRE(cp.n_rows >= 1 && cp.n_rows <= SYG_MAX_BANDS, "stft2048_mel: contrast rows must be in [1, %d]",
                SYG_MAX_BANDS);
    for (int r = 0; r < cp.n_rows; ++r) {
      cp.lo[r] = cplan_host[1 + r];
      cp.hi[r] = cplan_host[1 + SYG_MAX_BANDS + r];
      cp.k[r] = cplan_host[1 + 2 * SYG_MAX_BANDS + r];
      SYG_REQUIRE(cp.lo[r] >= 0 && cp.hi[r] <= NBIN && cp.lo[r] < cp.hi[r] && cp.k[r] >= 1 &&
                      cp.k[r] <= cp.hi[r] - cp.lo[r],
                  "stft2048_mel: contrast band %d invalid (lo=%d hi=%d k=%d)", r, cp.lo[r], cp.hi[r], cp.k[r]);
    }
  }
  if (stats_out) SYG_REQUIRE(sr > 0.f && roll_percent >= 0.f && roll_percent <= 1.f && bw_p > 0.f,
                             "stft2048_mel: invalid statistics parameters");
  const int pad = center ? NFFT / 2 : 0;
  const int tiles = (int)((T + TILE_T - 1) / TILE_T);
  const bool vec2 = (hop % 2 == 0) && (ldy % 2 == 0) && (((uintptr_t)y) % 8 == 0);
  const int64_t total_tiles = B * tiles;
  int wgs = 0, per = 0;
  persistent_grid(total_tiles, wgs, per);
  dim3 grid((unsigned)wgs), block(NTHREADS);
  hipStream_t st = (hipStream_t)stream;
  const float binhz = sr / (float)NFFT;
  const bool extra = (stats_out != nullptr) || (contrast_out != nullptr);
  auto kern = extra ? (vec2 ? stft2048_kernel<true, 1> : stft2048_kernel<false, 1>)
                    : (vec2 ? stft2048_kernel<true, 0> : stft2048_kernel<false, 0>);
  static bool attr_set[4] = {false, false, false, false};
  const int ai = (extra ? 2 : 0) + (vec2 ? 1 : 0);
  if (!attr_set[ai]) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)LDS_BYTES_MEL);
    if (e != hipSuccess) { set_error("stft2048_mel: cannot reserve %zu B LDS: %s", LDS_BYTES_MEL, hipGetErrorString(e)); return SYG_E_LAUNCH; }
    attr_set[ai] = true;
  }
  hipLaunchKernelGGL(kern, grid, block, LDS_BYTES_MEL, st, y, L, ldy, hop, pad, T, tiles, total_tiles, per,
                     (const float2*)window, (const float2*)twiddle, wpacked, plan, n_mels, mel_out, binhz,
                     roll_percent, bw_p, stats_out, cp, contrast_out, (float2*)nullptr);
  SYG_CHECK_LAUNCH("stft2048_mel");
  return SYG_OK;
}

extern "C" int syg_stft2048_c2c_f32(const float* y, int64_t B, int64_t L, int64_t ldy, int hop, int center,
                                    int64_t T, const float* window, const float* twiddle, float* out,
                                    void* stream) {
  int rc = check_common(y, B, L, ldy, hop, center, T, window, twiddle);
  if (rc) return rc;
  SYG_REQUIRE(out, "stft2048_c2c: null output");
  const int pad = center ? NFFT / 2 : 0;
  const int tiles = (int)((T + TILE_T - 1) / TILE_T);
  const bool vec2 = (hop % 2 == 0) && (ldy % 2 == 0) && (((uintptr_t)y) % 8 == 0);
  const int64_t total_tiles = B * tiles;
  int wgs = 0, per = 0;
  persistent_grid(total_tiles, wgs, per);
  dim3 grid((unsigned)wgs), block(NTHREADS);
  MelPlan plan = {};
  ContrastPlan cp = {};
  auto kern = vec2 ? stft2048_kernel<true, 2> : stft2048_kernel<false, 2>;
  static bool attr_set[2] = {false, false};
  if (!attr_set[vec2]) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)LDS_BYTES_C2C);
    if (e != hipSuccess) { set_error("stft2048_c2c: cannot reserve LDS: %s", hipGetErrorString(e)); return SYG_E_LAUNCH; }
    attr_set[vec2] = true;
  }
  hipLaunchKernelGGL(kern, grid, block, LDS_BYTES_C2C, (hipStream_t)stream, y, L, ldy, hop, pad, T, tiles,
                     total_tiles, per, (const float2*)window, (const float2*)twiddle, (const float*)nullptr, plan, 0,
                     (float*)nullptr, 0.f, 0.f, 0.f, (float*)nullptr, cp, (float*)nullptr, (float2*)out);
  SYG_CHECK_LAUNCH("stft2048_c2c");
  return SYG_OK;
}
