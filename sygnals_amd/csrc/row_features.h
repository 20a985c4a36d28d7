// Per-frame row functions on an LDS power row, one wave per row: the spectral statistics of
// sygnals/core/features/frequency_domain.py:24-386 (centroid, bandwidth, flatness, rolloff, dominant frequency; driven per
// frame by manager.py:289-316) and the tail means of librosa.feature.spectral_contrast (frequency_domain.py:147-212,
// manager.py:318-343).  Shared by the fused kernels of every frame length (stft_mel.hip: 1025 bins; stft_mel_w1024_seg.hip:
// 513; stft_mel_wseg_small.hip: 257 / 129; stft_mel_w4096.hip: 2049): NBIN = bins per row (16 NL + 1), PS: the row holds 4^PS |X|^2 (the kernels that
// pack 2 / 4 / 8 real frames into one complex transform leave the halvings of the split out; the factor -- a power of two
// -- is taken back here, exactly).
// Rows are skewed: bin k sits at word ppos(k) = k + k / 16.  Included inside namespace syg { namespace { ... } } after common.h.
#pragma once

// position of bin k inside an LDS power row: one pad word every 16 bins turns the stride-16 bin pattern
// of the pass-3 output into a conflict-free store while 16-aligned runs of bins stay contiguous for the MFMA
__device__ __forceinline__ int ppos(int k) { return k + (k >> 4); }

// ----------------------------------------------------------------------------------
// per-row statistics from an LDS power row (one wave per row)
// ----------------------------------------------------------------------------------
// Hardware transcendental forms (v_sqrt / v_log / v_exp / v_rcp_f32, <= 1 ulp): the IEEE-exact library
// versions expand to 15-200 instructions each, and 17 inlined copies overflow the instruction cache.
__device__ __forceinline__ float fsqrt(float x) { return __builtin_amdgcn_sqrtf(x); }
__device__ __forceinline__ float frcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float flog(float x) { return __builtin_amdgcn_logf(x) * 0.69314718055994531f; }  // ln
__device__ __forceinline__ float fexp(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504088896341f); }
__device__ __forceinline__ float fpow(float x, float p) {    // x >= 0
  return (x > 0.f) ? __builtin_amdgcn_exp2f(p * __builtin_amdgcn_logf(x)) : 0.f;
}

// The row functions stay out of line (inlined, their registers would spill the FFT loop).  Their pointers
// carry the address space: a generic pointer would turn every row read into a flat_load.
typedef const __attribute__((address_space(3))) float* lds_row;
constexpr int CONTRAST_SELBITS = 16;   // bits of the selection threshold that are decided (sign, exponent, 7 mantissa bits)

// smask bits: 1 centroid, 2 bandwidth, 4 flatness, 8 rolloff, 16 dominant (only the requested rows are
// computed and written; MAG_SUM / POWER_SUM / margin ride along with centroid / rolloff)
// Returns the statistics in the lanes of one register: lane SYG_STAT_x holds row x (every value is wave-uniform when it
// is formed, so any lane can keep it).  The function does NOT store: a later out-of-line call would wait for the stores
// at its entry (s_waitcnt vmcnt(0)), the caller writes the rows behind its last call (stats_row_mask() says which).
__device__ __forceinline__ int stats_row_mask(int smask) {
  return ((smask & 1) ? (1 << SYG_STAT_CENTROID) | (1 << SYG_STAT_MAG_SUM) : 0) | ((smask & 2) ? (1 << SYG_STAT_BANDWIDTH) : 0) |
         ((smask & 4) ? (1 << SYG_STAT_FLATNESS) : 0) | ((smask & 16) ? (1 << SYG_STAT_DOMINANT_BIN) : 0) |
         ((smask & 8) ? (1 << SYG_STAT_ROLLOFF_BIN) | (1 << SYG_STAT_POWER_SUM) | ((smask & 32) ? 0 : (1 << SYG_STAT_ROLLOFF_MARGIN)) : 0);
}
template <int NBIN, int PS>
__device__ __forceinline__ float row_stats_body(lds_row prow, int lane, float binhz, float roll_percent, float bw_p, int smask) {
  float res = 0.f;
#define SYG_PUT(row, val) res = (lane == (row)) ? (val) : res
  // lane owns the 16 contiguous bins [16 lane, 16 lane + 16) -- 16 consecutive words at 17 lane of the skewed row:
  // immediate offsets, no bank conflicts -- and the last owning lane (63 of a 1025-bin row) also the Nyquist bin as a 17th
  // value (0 in the other lanes).  Rows of 513 / 257 / 129 bins: the lanes 32 / 16 / 8 ... 63 own nothing (zeros: no
  // contribution to any sum, never an extreme unless the row is all zero -- lane 0 then wins, as numpy's argmax does).
  // Rows of 2049 bins (frame length 4096): G = 2 blocks of 16 bins per lane, 34 words apart per lane, Nyquist the 33rd value.
  // The powers are read ONCE and every statistic works on the registers (round 2 re-read the row per pass to
  // stay inside the caller-saved registers; 17 + 17 values still do).
  constexpr int NG = (NBIN - 1) / 16;            // 16-bin blocks of the row
  constexpr int G = NG > 64 ? 2 : 1;             // blocks per lane
  constexpr int NL = NG / G;                     // lanes that own bins
  constexpr int NV = 16 * G;                     // bins per lane (+ the Nyquist slot)
  static_assert(NBIN == 16 * G * NL + 1 && NL >= 1 && NL <= 64, "rows of 16 NL + 1 bins (at most 1025), or 2049");
  constexpr bool FULL = (NL == 64);
  const float EPS = 2.220446049250313e-16f;
  const bool last = (lane == NL - 1);
  const bool own = FULL || lane < NL;
  float p[NV + 1];
  {
    lds_row pr = prow + 17 * G * (own ? lane : 0);
#pragma unroll
    for (int i = 0; i < NV; ++i) p[i] = pr[i + (i >> 4)];
    const float nyq = pr[17 * G];                  // ppos(16 G NL) = 17 G (NL - 1) + 17 G for the last owning lane (inside the row's slack elsewhere)
    if (!FULL) {
#pragma unroll
      for (int i = 0; i < NV; ++i) p[i] = own ? p[i] : 0.f;
    }
    p[NV] = last ? nyq : 0.f;
    if (PS > 0) {
      constexpr float PSC = 1.f / (float)(1 << (2 * PS));
#pragma unroll
      for (int i = 0; i < NV + 1; ++i) p[i] *= PSC;
    }
  }
  float psum = 0.f;
#pragma unroll
  for (int i = 0; i < NV + 1; ++i) psum += p[i];
  const float tot_p = wave_sum(psum);
  const float kb = (float)(NV * lane);
  float tot_m = 0.f, cen_bin = 0.f;
  bool live = false;
  if (smask & (1 | 2 | 4)) {        // magnitude sums
    float m[NV + 1];
    float msum = 0.f, fl = 0.f;
#pragma unroll
    for (int i = 0; i < NV + 1; ++i) {
      m[i] = fsqrt(p[i]);
      msum += m[i];
      fl = fmaf(m[i], (float)i, fl);               // sum m (k - 16 lane): the lane's base enters once below
    }
    tot_m = wave_sum(msum);
    live = tot_m >= EPS;
    const float tot_f = wave_sum(fmaf(kb, msum, fl));
    cen_bin = live ? tot_f * frcp(tot_m) : 0.f;
    SYG_PUT(SYG_STAT_CENTROID, cen_bin * binhz);
    SYG_PUT(SYG_STAT_MAG_SUM, tot_m);
    if (smask & 4) {    // flatness: exp(mean log(m + eps)) / mean(m)
      float lsum = 0.f;
#pragma unroll
      for (int i = 0; i < NV; ++i) lsum += __builtin_amdgcn_logf(m[i] + EPS);
      const float l16 = __builtin_amdgcn_logf(m[NV] + EPS);
      if (!FULL) lsum = own ? lsum : 0.f;
      lsum += last ? l16 : 0.f;
      const float tot_l = wave_sum(lsum) * 0.69314718055994531f;
      const float am = tot_m * (1.f / (float)NBIN);
      SYG_PUT(SYG_STAT_FLATNESS, (am >= EPS) ? fminf(fmaxf(fexp(tot_l * (1.f / (float)NBIN)) * frcp(am), 0.f), 1.f) : 0.f);
    }
    if (smask & 2) {    // bandwidth: (sum m |f - c|^p / sum m)^(1/p);  (m[NV] = 0 outside the last owning lane)
      const int pmode = (bw_p == 2.f) ? 2 : (bw_p == 1.f) ? 1 : 0;
      const float d0 = kb - cen_bin;
      float dsum = 0.f;
      if (pmode == 2) {
#pragma unroll
        for (int i = 0; i < NV + 1; ++i) { const float d = (d0 + (float)i) * binhz; dsum = fmaf(m[i], d * d, dsum); }
      } else {
#pragma unroll
        for (int i = 0; i < NV + 1; ++i) {
          const float d = fabsf(d0 + (float)i) * binhz;
          dsum = fmaf(m[i], pmode == 1 ? d : fpow(d, bw_p), dsum);
        }
      }
      const float tot_d = wave_sum(dsum);
      const float r = live ? fmaxf(tot_d * frcp(tot_m), 0.f) : 0.f;
      SYG_PUT(SYG_STAT_BANDWIDTH, pmode == 2 ? fsqrt(r) : pmode == 1 ? r : fpow(r, frcp(bw_p)));
    }
  }
  if (smask & 16) {   // argmax of the magnitude == argmax of the power (first occurrence)
    float pmax = p[0];
    int amax = 0;
#pragma unroll
    for (int i = 1; i < NV + 1; ++i) {
      const bool up = (i < NV || last) && p[i] > pmax;
      pmax = up ? p[i] : pmax; amax = up ? i : amax;
    }
    const float gm = wave_max(pmax);
    const int cand = wave_min_i((pmax == gm) ? NV * lane + amax : 0x7fffffff);
    SYG_PUT(SYG_STAT_DOMINANT_BIN, (float)cand);
  }
  if (smask & 8) {    // rolloff: first bin with cumsum(power) >= roll * total
    // The running sum never decreases (powers are >= 0), so the number of a lane's sums below the threshold IS the
    // position of its first hit; the decision margin is the distance of the threshold to the nearest running sum on
    // either side (the sum in front of bin 0 excepted).
    // (SYG_SM_NO_MARGIN: callers that do not read the margin row -- the C4 block -- skip its three instructions per bin.)
    const float thr = roll_percent * tot_p;
    float c = wave_excl_scan(psum, lane);
    int below = 0;
    float mgw = 0.f;
    if (smask & 32) {
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        c += p[i];
        below += (c < thr) ? 1 : 0;
      }
      c += p[NV];
      below += (last && c < thr) ? 1 : 0;
    } else {
      float mg = (lane > 0) ? fabsf(c - thr) : 3.4e38f;
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        c += p[i];
        below += (c < thr) ? 1 : 0;
        mg = fminf(mg, fabsf(c - thr));
      }
      c += p[NV];
      below += (last && c < thr) ? 1 : 0;
      mg = fminf(mg, fabsf(c - thr));
      mgw = wave_min(mg);
    }
    const int rb = (below < (last ? NV + 1 : NV)) ? NV * lane + below : 0x7fffffff;
    int rbmin = wave_min_i(rb);
    if (rbmin >= NBIN || tot_p < EPS) rbmin = NBIN - 1;      // (no hit: 0x7fffffff, or a lane that owns nothing)
    SYG_PUT(SYG_STAT_ROLLOFF_BIN, (float)rbmin);
    SYG_PUT(SYG_STAT_POWER_SUM, tot_p);
    SYG_PUT(SYG_STAT_ROLLOFF_MARGIN, (tot_p > 0.f) ? mgw * frcp(tot_p) : 0.f);
  }
#undef SYG_PUT
  return res;
}

// k-th order statistic of the powers of bins [lo, lo + n) by a 32-step radix select on the float bit patterns
// (fallback for long bands / large k)
__device__ __forceinline__ uint32_t row_kth(lds_row prow, int lane, int lo, int n, int kk, bool largest) {
  uint32_t prefix = 0;
  int remaining = kk;
  for (int bit = 31; bit >= 0; --bit) {
    const uint32_t mask = ~((1u << bit) - 1u);
    const uint32_t want = largest ? (prefix | (1u << bit)) : prefix;
    int cnt = 0;
    for (int i = lane; i < n; i += 64) cnt += ((__float_as_uint(prow[ppos(lo + i)]) & mask) == want) ? 1 : 0;
    cnt = wave_sum_i(cnt);
    if (largest) {
      if (cnt >= remaining) prefix |= (1u << bit); else remaining -= cnt;
    } else {
      if (cnt < remaining) { remaining -= cnt; prefix |= (1u << bit); }
    }
  }
  return prefix;
}

// the kk-th largest over the lanes of two 32-bit values per lane (two independent selections in one loop): bisection
// from the top bit, per bit and value one vector compare and a scalar popcount.  kk is made scalar here (it arrives in
// a vector register when the caller is an out-of-line function): thresholds and counts then live on the scalar unit.
// Only bits 31 .. LOWBIT are decided: the result is the k-th largest ROUNDED DOWN to that precision -- still a value
// with at least kk lane values at or above it, which is all the selection below needs (a lower threshold only lets a
// few more candidates through).
template <int LOWBIT>
__device__ __forceinline__ void wave_kth_largest2_u32(uint32_t x, uint32_t y, int kk, uint32_t& tx, uint32_t& ty) {
  const int ks = __builtin_amdgcn_readfirstlane(kk);
  uint32_t a = 0, b = 0;
#pragma unroll 4
  for (int bit = 31; bit >= LOWBIT; --bit) {
    const uint32_t ca = a | (1u << bit), cb = b | (1u << bit);
    const int na = __popcll(__ballot(x >= ca)), nb = __popcll(__ballot(y >= cb));
    a = (na >= ks) ? ca : a;
    b = (nb >= ks) ? cb : b;
  }
  tx = a; ty = b;
}

// The same over TWO values per lane and side (the kk-th largest of the 128 values x1, x2 / y1, y2): a tighter threshold
// for callers whose lanes hold sorted lists -- the kk-th largest of the lanes' two top values is much closer to the
// kk-th largest of everything than the kk-th largest lane MAXIMUM is (a lane with two of the top kk values is common,
// one with three is rare), so that fewer candidates pass it and have to be taken back one by one.
// x*: non-negative floats as bits (bit 31 clear), y*: complements of such (bit 31 set): the top bit is known.
template <int LOWBIT>
__device__ __forceinline__ void wave_kth_largest2x2_u32(uint32_t x1, uint32_t x2, uint32_t y1, uint32_t y2, int kk, uint32_t& tx,
                                                        uint32_t& ty) {
  const int ks = __builtin_amdgcn_readfirstlane(kk);
  uint32_t a = 0, b = 0x80000000u;
#pragma unroll 5
  for (int bit = 30; bit >= LOWBIT; --bit) {
    const uint32_t ca = a | (1u << bit), cb = b | (1u << bit);
    const int na = __popcll(__ballot(x1 >= ca)) + __popcll(__ballot(x2 >= ca));
    const int nb = __popcll(__ballot(y1 >= cb)) + __popcll(__ballot(y2 >= cb));
    a = (na >= ks) ? ca : a;
    b = (nb >= ks) ? cb : b;
  }
  tx = a; ty = b;
}

// Data-oblivious sorting networks for the R values a lane holds (Batcher's odd-even merge sort pruned to R wires;
// checked with the 0-1 principle, tools/sortnet.py).
template <int R> struct SortNet;
template <> struct SortNet<1> { static constexpr int N = 0; static constexpr int P[1][2] = {{0, 0}}; };
template <> struct SortNet<2> { static constexpr int N = 1; static constexpr int P[1][2] = {{0, 1}}; };
template <> struct SortNet<4> {
  static constexpr int N = 5;
  static constexpr int P[5][2] = {{0, 1}, {2, 3}, {0, 2}, {1, 3}, {1, 2}};
};
template <> struct SortNet<5> {
  static constexpr int N = 9;
  static constexpr int P[9][2] = {{0, 1}, {2, 3}, {0, 2}, {1, 3}, {1, 2}, {0, 4}, {2, 4}, {1, 2}, {3, 4}};
};
template <> struct SortNet<7> {
  static constexpr int N = 16;
  static constexpr int P[16][2] = {{0, 1}, {2, 3}, {0, 2}, {1, 3}, {1, 2}, {4, 5}, {4, 6}, {5, 6},
                                   {0, 4}, {2, 6}, {2, 4}, {1, 5}, {3, 5}, {1, 2}, {3, 4}, {5, 6}};
};
template <> struct SortNet<8> {
  static constexpr int N = 19;
  static constexpr int P[19][2] = {{0, 1}, {2, 3}, {0, 2}, {1, 3}, {1, 2}, {4, 5}, {6, 7}, {4, 6}, {5, 7}, {5, 6},
                                   {0, 4}, {2, 6}, {2, 4}, {1, 5}, {3, 7}, {3, 5}, {1, 2}, {3, 4}, {5, 6}};
};
template <> struct SortNet<10> {
  static constexpr int N = 32;
  static constexpr int P[32][2] = {{0, 1}, {2, 3}, {0, 2}, {1, 3}, {1, 2}, {4, 5}, {6, 7}, {4, 6}, {5, 7}, {5, 6}, {0, 4},
                                   {2, 6}, {2, 4}, {1, 5}, {3, 7}, {3, 5}, {1, 2}, {3, 4}, {5, 6}, {8, 9}, {0, 8}, {4, 8},
                                   {2, 4}, {6, 8}, {1, 9}, {5, 9}, {3, 5}, {7, 9}, {1, 2}, {3, 4}, {5, 6}, {7, 8}};
};
template <> struct SortNet<12> {
  static constexpr int N = 41;
  static constexpr int P[41][2] = {{0, 1}, {2, 3}, {0, 2}, {1, 3}, {1, 2}, {4, 5}, {6, 7}, {4, 6}, {5, 7}, {5, 6}, {0, 4},
                                   {2, 6}, {2, 4}, {1, 5}, {3, 7}, {3, 5}, {1, 2}, {3, 4}, {5, 6}, {8, 9}, {10, 11},
                                   {8, 10}, {9, 11}, {9, 10}, {0, 8}, {4, 8}, {2, 10}, {6, 10}, {2, 4}, {6, 8}, {1, 9},
                                   {5, 9}, {3, 11}, {7, 11}, {3, 5}, {7, 9}, {1, 2}, {3, 4}, {5, 6}, {7, 8}, {9, 10}};
};

// The band sits in R registers per lane (bin lo + 64 r + lane in register r).  Each lane first sorts its own R
// values (two copies: `up` ascending with -1 in the unused slots, `dn` descending with +huge), so that a lane's
// candidate for the next largest / smallest is always in its last register.  One extraction is then a wave max / min
// over those heads, and the first owning lane shifts its list by one: 2 R selects per step instead of the
// 8 R compare / select operations of a search through unsorted registers.
template <int R>
__device__ __forceinline__ void contrast_extract(lds_row prow, int lane, int lo, int n, int k, float& spk, float& svl) {
  float up[R], dn[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int i = r * 64 + lane;
    const float p = prow[ppos(lo + (i < n ? i : 0))];
    up[r] = (i < n) ? p : -1.f;
    dn[r] = (i < n) ? p : 3.4e38f;
  }
#pragma unroll
  for (int c = 0; c < SortNet<R>::N; ++c) {
    constexpr auto& P = SortNet<R>::P;
    const int i = P[c][0], j = P[c][1];
    const float ua = up[i], ub = up[j], da = dn[i], db = dn[j];
    up[i] = fminf(ua, ub); up[j] = fmaxf(ua, ub);
    dn[i] = fmaxf(da, db); dn[j] = fminf(da, db);
  }
  spk = 0.f; svl = 0.f;
  for (int it = 0; it < k; ++it) {
    float MH = up[R - 1], ML = dn[R - 1];
    wave_maxmin(MH, ML);
    const int fh = __ffsll((long long)__ballot(up[R - 1] == MH)) - 1;
    const int fl = __ffsll((long long)__ballot(dn[R - 1] == ML)) - 1;
    const bool mh = lane == fh, ml = lane == fl;
#pragma unroll
    for (int r = R - 1; r > 0; --r) {
      up[r] = mh ? up[r - 1] : up[r];
      dn[r] = ml ? dn[r - 1] : dn[r];
    }
    up[0] = mh ? -1.f : up[0];
    dn[0] = ml ? 3.4e38f : dn[0];
    spk += fsqrt(MH);
    svl += fsqrt(ML);
  }
}

// Wide bands (R = 12 registers per lane: 449..768 bins, config C4's 751-bin top band with k = 15).  The register form
// above pays 2 R selects per extraction to shift two sorted lists; here a lane sorts its values ONCE into one
// ascending list, parks it transposed in a dead part of its own row -- words [0, 64 R) of the row: the bands below
// this one are finished and this band's values are in registers (the caller guarantees ascending band order and
// 64 R <= ppos(hi)) -- and an extraction moves a head index and re-reads one word: the largest values are consumed
// from the top of the list, the smallest from the bottom, independently (as two sorted copies would be).
template <int R>
__device__ __forceinline__ void contrast_extract_lds(lds_row prow, int lane, int lo, int n, int k, float& spk, float& svl) {
  typedef __attribute__((address_space(3))) float* lds_wrow;
  lds_wrow wrow = (lds_wrow)prow;
  float v[R];
  // bin lo + lane + 64 r sits at ppos(lo + lane) + 68 r (64 r / 16 = 4 r pad words, no carry): one base, immediate
  // offsets.  Lanes past the band's end read on (still inside the LDS allocation) and are replaced by the pad.
  lds_row pr = prow + ppos(lo + lane);
  const int nrem = n - lane;                      // this lane holds the values r with 64 r < nrem
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const float p = pr[68 * r];
    v[r] = (64 * r < nrem) ? p : 3.4e38f;         // pads sort to the top of the list and are never a head
  }
  const int nvu = (nrem + 63) >> 6;
  const int nv = nvu < 0 ? 0 : (nvu > R ? R : nvu);   // valid values of this lane
#pragma unroll
  for (int c = 0; c < SortNet<R>::N; ++c) {
    constexpr auto& P = SortNet<R>::P;
    const int i = P[c][0], j = P[c][1];
    const float a = v[i], b = v[j];
    v[i] = fminf(a, b); v[j] = fmaxf(a, b);
  }
  wave_lds_sync();                                // every lane has read its band values: the row may be overwritten
#pragma unroll
  for (int r = 0; r < R; ++r) wrow[r * 64 + lane] = v[r];
  wave_lds_sync();
  int ht = nv - 1, hb = 0;                        // head indices: next largest / next smallest of this lane
  float MHl = (nv > 0) ? wrow[(nv > 0 ? ht : 0) * 64 + lane] : -1.f, MLl = (nv > 0) ? v[0] : 3.4e38f;
  // Selection instead of k extractions (k >= 8).  The k-th largest of the 64 lane maxima, T, has at least k values of
  // the band at or above it, all of them in the lanes whose maximum reaches it: count them (C), sum their magnitudes, and
  // take back the C - k smallest of them -- two or three wave-wide rounds instead of k (the lane maxima of 12 values
  // each are the top of the band: C - k is small; when it is not, ties or a constant band, the k rounds below run).
  // T by bisection on the bit patterns (non-negative floats order like their bits): 32 compares + scalar popcounts.
  if (k >= 8 && n >= 64) {                      // (wave-uniform; every lane holds at least one value)
    uint32_t Tu, Bu;                              // Bu: k-th smallest of the lane minima = ~(k-th largest of their complements)
    // 16 of the 32 bits (sign, exponent, 7 mantissa bits: the threshold is within 1 % of the exact order statistic)
    wave_kth_largest2_u32<16>(__float_as_uint(MHl), ~__float_as_uint(MLl), k, Tu, Bu);
    Bu = ~Bu;
    const float Th = __uint_as_float(Tu), Tl = __uint_as_float(Bu);
    int ch = nv - R, cl = 0;                      // (the pads, +huge, count as >= Th: taken off up front)
#pragma unroll
    for (int r = 0; r < R; ++r) { ch += (v[r] >= Th) ? 1 : 0; cl += (v[r] <= Tl) ? 1 : 0; }
    const int Eh = wave_sum_i(ch) - k, El = wave_sum_i(cl) - k;
    if (Eh <= 12 && El <= 12) {
      const int cmax = (int)wave_max((float)(ch > cl ? ch : cl));
      float ah = 0.f, al = 0.f;
      for (int t = 0; t < cmax; ++t) {
        const int ih = nv - 1 - t;
        const float vh = wrow[(ih > 0 ? ih : 0) * 64 + lane], vl = wrow[(t < R ? t : 0) * 64 + lane];
        ah += (t < ch) ? fsqrt(vh) : 0.f;
        al += (t < cl) ? fsqrt(vl) : 0.f;
      }
      float Sh = wave_sum(ah), Sl = wave_sum(al);
      // the extras: the smallest of the upper candidates, the largest of the lower ones
      const int emax = Eh > El ? Eh : El;
      for (int e = 0; e < emax; ++e) {
        float lo_c = (ch > 0) ? wrow[(nv - ch) * 64 + lane] : 3.4e38f;       // this lane's smallest upper candidate
        float hi_c = (cl > 0) ? wrow[(cl - 1) * 64 + lane] : -1.f;           // its largest lower candidate
        float MH = hi_c, ML = lo_c;
        wave_maxmin(MH, ML);
        const int fh = __ffsll((long long)__ballot(hi_c == MH)) - 1;
        const int fl = __ffsll((long long)__ballot(lo_c == ML)) - 1;
        if (e < Eh) { Sh -= fsqrt(ML); ch -= (lane == fl) ? 1 : 0; }
        if (e < El) { Sl -= fsqrt(MH); cl -= (lane == fh) ? 1 : 0; }
      }
      spk = Sh; svl = Sl;
      return;
    }
  }
  spk = 0.f; svl = 0.f;
  for (int it = 0; it < k; ++it) {
    float MH = MHl, ML = MLl;
    wave_maxmin(MH, ML);
    const int fh = __ffsll((long long)__ballot(MHl == MH)) - 1;
    const int fl = __ffsll((long long)__ballot(MLl == ML)) - 1;
    ht -= (lane == fh) ? 1 : 0;
    hb += (lane == fl) ? 1 : 0;
    // (a lane re-reads its heads every round: the address only moves in the two winning lanes)
    const float nh = wrow[(ht >= 0 ? ht : 0) * 64 + lane], nl = wrow[(hb < nv ? hb : 0) * 64 + lane];
    MHl = (ht >= 0) ? nh : -1.f;
    MLl = (hb < nv) ? nl : 3.4e38f;
    spk += fsqrt(MH);
    svl += fsqrt(ML);
  }
}

// k = 1: the band's largest and smallest power.  Lanes past the band's end re-read its last bin (a duplicate changes
// neither extreme): no masks.
__device__ __forceinline__ void contrast_minmax(lds_row prow, int lane, int lo, int n, float& spk, float& svl) {
  float hi = 0.f, lw = 3.4e38f;
  for (int r0 = 0; r0 < n; r0 += 64) {
    const int i = r0 + lane;
    const float p = prow[ppos(lo + (i < n ? i : n - 1))];
    hi = fmaxf(hi, p);
    lw = fminf(lw, p);
  }
  wave_maxmin(hi, lw);
  spk = fsqrt(hi); svl = fsqrt(lw);
}

// k <= 3 on bands of up to 192 bins: every lane sorts its (up to) three values, then the sorted triples are merged over
// the wave by a DPP butterfly -- the three largest of the union of two descending triples a, b are
//   c1 = max(a1, b1)   c2 = max(a2, b2, min(a1, b1))   c3 = max(a3, b3, min(a2, b1), min(a1, b2))
// (and the mirror image for the three smallest): six steps of ten instructions, no scalar round trip, no loop over k.
// A merge of a triple with ITSELF is wrong (elements would count twice): the row-broadcast steps leave garbage in the
// rows they do not write, which no later step reads -- the result is taken from lane 63.
template <int CTRL, int ROWMASK, bool TOP>
__device__ __forceinline__ void merge3_step(float& a1, float& a2, float& a3) {
  float b1, b2, b3;
  if (ROWMASK == 0xF) { b1 = dpp_f<CTRL>(a1); b2 = dpp_f<CTRL>(a2); b3 = dpp_f<CTRL>(a3); }
  else { b1 = dpp_rows_f<CTRL, ROWMASK>(a1); b2 = dpp_rows_f<CTRL, ROWMASK>(a2); b3 = dpp_rows_f<CTRL, ROWMASK>(a3); }
  if (TOP) {
    const float c3 = fmaxf(fmaxf(a3, b3), fmaxf(fminf(a2, b1), fminf(a1, b2)));
    const float c2 = fmaxf(fmaxf(a2, b2), fminf(a1, b1));
    a1 = fmaxf(a1, b1); a2 = c2; a3 = c3;
  } else {
    const float c3 = fminf(fminf(a3, b3), fminf(fmaxf(a2, b1), fmaxf(a1, b2)));
    const float c2 = fminf(fminf(a2, b2), fmaxf(a1, b1));
    a1 = fminf(a1, b1); a2 = c2; a3 = c3;
  }
}
template <bool TOP>
__device__ __forceinline__ void wave_merge3(float& a1, float& a2, float& a3) {
  merge3_step<DPP_QP_1032, 0xF, TOP>(a1, a2, a3);
  merge3_step<DPP_QP_2301, 0xF, TOP>(a1, a2, a3);
  merge3_step<DPP_ROW_HALF_MIRROR, 0xF, TOP>(a1, a2, a3);
  merge3_step<DPP_ROW_MIRROR, 0xF, TOP>(a1, a2, a3);
  merge3_step<DPP_ROW_BCAST15, 0xA, TOP>(a1, a2, a3);
  merge3_step<DPP_ROW_BCAST31, 0xC, TOP>(a1, a2, a3);
  a1 = rl_f(a1, 63); a2 = rl_f(a2, 63); a3 = rl_f(a3, 63);
}
__device__ __forceinline__ void contrast_top3(lds_row prow, int lane, int lo, int n, int k, float& spk, float& svl) {
  float t[3], u[3];                                  // descending with -1 pads / ascending with +huge pads
  lds_row pr = prow + ppos(lo + lane);              // (bin lo + lane + 64 r at ppos(lo + lane) + 68 r)
  const int nrem = n - lane;
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    const float p = pr[68 * r];
    t[r] = (64 * r < nrem) ? p : -1.f;
    u[r] = (64 * r < nrem) ? p : 3.4e38f;
  }
  auto cx = [](float& hi, float& lw) { const float a = hi, b = lw; hi = fmaxf(a, b); lw = fminf(a, b); };
  cx(t[0], t[1]); cx(t[1], t[2]); cx(t[0], t[1]);    // t0 >= t1 >= t2
  cx(u[1], u[0]); cx(u[2], u[1]); cx(u[1], u[0]);    // u0 <= u1 <= u2
  wave_merge3<true>(t[0], t[1], t[2]);
  wave_merge3<false>(u[0], u[1], u[2]);
  spk = fsqrt(t[0]) + (k >= 2 ? fsqrt(t[1]) : 0.f) + (k >= 3 ? fsqrt(t[2]) : 0.f);
  svl = fsqrt(u[0]) + (k >= 2 ? fsqrt(u[1]) : 0.f) + (k >= 3 ? fsqrt(u[2]) : 0.f);
}

// Wide bands whose last register is the only partly filled one (64 (R - 1) < n <= 64 R; C4's 751-bin band at 48 kHz,
// the 728-bin band at 44.1 kHz: R = 12), 4 <= k <= 16: selection on STATIC registers, no parked lists, no re-reads.
// A lane sorts its R values once (pads +huge on top); its four largest are then v[R-1 .. R-4], one register lower in
// the lanes that hold a pad, its four smallest v[0 .. 3].  The threshold Th = k-th largest lane maximum (rounded down:
// wave_kth_largest2_u32) has at least k values at or above it, all of them among the lanes' top values; the first
// three of each lane are counted and summed, the few extras (count - k) are taken back smallest first, one wave-wide
// round each.  A lane whose FOURTH value still reaches the threshold might hide a fifth: the function then reports
// failure and the caller runs the general form (contrast_extract_lds) -- as it does for many extras (ties, constant
// bands).  Mirror image for the k smallest.
template <int R>
__device__ __forceinline__ bool contrast_select(lds_row prow, int lane, int lo, int n, int k, float& spk, float& svl) {
  float v[R];
  lds_row pr = prow + ppos(lo + lane);            // bin lo + lane + 64 r at ppos(lo + lane) + 68 r
#pragma unroll
  for (int r = 0; r < R; ++r) v[r] = pr[68 * r];
  const bool hp = lane >= n - 64 * (R - 1);       // no value of this lane in the last register
  v[R - 1] = hp ? 3.4e38f : v[R - 1];
#pragma unroll
  for (int c = 0; c < SortNet<R>::N; ++c) {
    constexpr auto& P = SortNet<R>::P;
    const int i = P[c][0], j = P[c][1];
    const float a = v[i], b = v[j];
    v[i] = fminf(a, b); v[j] = fmaxf(a, b);
  }
  const float t1 = hp ? v[R - 2] : v[R - 1], t2 = hp ? v[R - 3] : v[R - 2], t3 = hp ? v[R - 4] : v[R - 3],
              t4 = hp ? v[R - 5] : v[R - 4];
  const float b1 = v[0], b2 = v[1], b3 = v[2], b4 = v[3];
  uint32_t Tu, Bu;
  wave_kth_largest2x2_u32<CONTRAST_SELBITS>(__float_as_uint(t1), __float_as_uint(t2), ~__float_as_uint(b1), ~__float_as_uint(b2), k, Tu, Bu);
  const float Th = __uint_as_float(Tu), Tl = __uint_as_float(~Bu);
  if (__ballot(t4 >= Th || b4 <= Tl) != 0) return false;
  int ch = (t1 >= Th ? 1 : 0) + (t2 >= Th ? 1 : 0) + (t3 >= Th ? 1 : 0);
  int cl = (b1 <= Tl ? 1 : 0) + (b2 <= Tl ? 1 : 0) + (b3 <= Tl ? 1 : 0);
  const int Eh = wave_sum_i(ch) - k, El = wave_sum_i(cl) - k;
  if (Eh > 10 || El > 10) return false;
  const float q1 = fsqrt(t1), q2 = fsqrt(t2), q3 = fsqrt(t3), r1 = fsqrt(b1), r2 = fsqrt(b2), r3 = fsqrt(b3);
  float Sh = wave_sum((ch >= 1 ? q1 : 0.f) + (ch >= 2 ? q2 : 0.f) + (ch >= 3 ? q3 : 0.f));
  float Sl = wave_sum((cl >= 1 ? r1 : 0.f) + (cl >= 2 ? r2 : 0.f) + (cl >= 3 ? r3 : 0.f));
  const int emax = Eh > El ? Eh : El;
  for (int e = 0; e < emax; ++e) {
    // this lane's smallest upper / largest lower candidate (as magnitudes: the order is the same)
    const float lo_c = ch == 3 ? q3 : ch == 2 ? q2 : ch == 1 ? q1 : 3.4e38f;
    const float hi_c = cl == 3 ? r3 : cl == 2 ? r2 : cl == 1 ? r1 : -1.f;
    float MH = hi_c, ML = lo_c;
    wave_maxmin(MH, ML);
    const int fh = __ffsll((long long)__ballot(hi_c == MH)) - 1;
    const int fl = __ffsll((long long)__ballot(lo_c == ML)) - 1;
    if (e < Eh) { Sh -= ML; ch -= (lane == fl) ? 1 : 0; }
    if (e < El) { Sl -= MH; cl -= (lane == fh) ? 1 : 0; }
  }
  spk = Sh; svl = Sl;
  return true;
}

// Bands of 769 ... 1536 bins (the upper octave bands of frame length 4096: 1502 bins with k = 30 at 48 kHz): the band
// sits in R = 24 registers per lane as bit patterns (non-negative floats order like their bits).  The k-th largest lane
// MAXIMUM, rounded down to 16 bits, has at least k values of the band at or above it -- a few dozen candidates out of
// 1500; they are compacted (lane counts, exclusive scan, predicated LDS writes) into words [0, 128) of the row -- the
// bands below are finished and this band's values are in registers: the caller guarantees ascending band order and
// ppos(lo) >= 256 -- so that the exact order statistic is a bisection over TWO values per lane (31 bits, two compares
// each) instead of 24, and the tail sum is closed with the tie count.  The k smallest the same way on the complements, in
// words [128, 256).  More than 128 candidates on a side (ties, a constant band): false, the caller's radix select runs.
template <int R>
__device__ __forceinline__ bool contrast_compact(lds_row prow, int lane, int lo, int n, int k, float& spk, float& svl) {
  typedef __attribute__((address_space(3))) uint32_t* lds_urow;
  lds_urow wrow = (lds_urow)prow;
  constexpr uint32_t PAD = 0xffffffffu;
  uint32_t v[R];
  lds_row pr = prow + ppos(lo + lane);            // bin lo + lane + 64 r at ppos(lo + lane) + 68 r
  const int nrem = n - lane;
  uint32_t mx = 0, mn = PAD;
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const bool in = 64 * r < nrem;
    const uint32_t u = __float_as_uint(pr[in ? 68 * r : 0]);
    v[r] = in ? u : PAD;
    mx = in ? (u > mx ? u : mx) : mx;
    mn = v[r] < mn ? v[r] : mn;
  }
  uint32_t Tu, Bu;
  wave_kth_largest2_u32<16>(mx, ~mn, k, Tu, Bu);
  Bu = ~Bu;                                       // at least k lane minima at or below it
  int ch = 0, cl = 0;
#pragma unroll
  for (int r = 0; r < R; ++r) { ch += (v[r] >= Tu && v[r] != PAD) ? 1 : 0; cl += (v[r] <= Bu) ? 1 : 0; }
  if (wave_sum_i(ch) > 128 || wave_sum_i(cl) > 128) return false;
  int oh = (int)wave_excl_scan((float)ch, lane), ol = 128 + (int)wave_excl_scan((float)cl, lane);
  wave_lds_sync();                                // every lane has read its band values: the row's head may be overwritten
  wrow[lane] = 0u; wrow[64 + lane] = 0u; wrow[128 + lane] = PAD; wrow[192 + lane] = PAD;
  wave_lds_sync();
#pragma unroll
  for (int r = 0; r < R; ++r) {
    if (v[r] >= Tu && v[r] != PAD) { wrow[oh] = v[r]; ++oh; }
    if (v[r] <= Bu) { wrow[ol] = v[r]; ++ol; }
  }
  wave_lds_sync();
  const uint32_t x1 = wrow[lane], x2 = wrow[64 + lane], y1 = ~wrow[128 + lane], y2 = ~wrow[192 + lane];   // (pads: 0 on both sides)
  uint32_t tx, ty;
  wave_kth_largest2x2_u32<0>(x1, x2, y1, y2, k, tx, ty);
  const float sh = (x1 > tx ? fsqrt(__uint_as_float(x1)) : 0.f) + (x2 > tx ? fsqrt(__uint_as_float(x2)) : 0.f);
  const float sl = (y1 > ty ? fsqrt(__uint_as_float(~y1)) : 0.f) + (y2 > ty ? fsqrt(__uint_as_float(~y2)) : 0.f);
  const int nh = (x1 > tx ? 1 : 0) + (x2 > tx ? 1 : 0), nl = (y1 > ty ? 1 : 0) + (y2 > ty ? 1 : 0);
  spk = wave_sum(sh) + (float)(k - wave_sum_i(nh)) * fsqrt(__uint_as_float(tx));
  svl = wave_sum(sl) + (float)(k - wave_sum_i(nl)) * fsqrt(__uint_as_float(~ty));
  return true;
}

// mean of the k smallest and k largest MAGNITUDES of bins [lo, hi) of one LDS power row (identical to sorting,
// as librosa does: values are non-negative, selection on power == selection on magnitude).
//   bands of <= 768 bins with k <= 16 : register extraction, specialised by registers per lane;
//   otherwise                         : radix select of the k-th order statistic + tail sum closed with the
//                                       tie count.
// may_park: the bands come in ascending order and the row's statistics are done, so a wide band may park its sorted
// lists in the part of the row below its own end (contrast_extract_lds)
// WIDE: rows of more than 1025 bins (the 24-register form of bands of 769 ... 1536 bins is only built into their kernels:
// inside the row functions of the 2048-sample kernels it would cost registers on a path their rows never take)
template <bool WIDE>
__device__ __forceinline__ float2 band_contrast(lds_row prow, int lane, int lo, int hi, int k, int may_park) {   // (peak, valley)
  const int n = hi - lo;
  if (n <= 768 && k <= 16) {
    float spk, svl;
    if (k == 1) contrast_minmax(prow, lane, lo, n, spk, svl);
    else if (k <= 3 && n <= 192) contrast_top3(prow, lane, lo, n, k, spk, svl);
    else if (n <= 64) contrast_extract<1>(prow, lane, lo, n, k, spk, svl);
    else if (n <= 128) contrast_extract<2>(prow, lane, lo, n, k, spk, svl);
    else if (n <= 256) contrast_extract<4>(prow, lane, lo, n, k, spk, svl);
    else {
      // Bands of more than 256 bins with k >= 4 whose registers are all full but the last (64 (R - 1) < n <= 64 R for
      // R = 5, 7, 8, 10, 12 -- the wide bands of the usual sample rates: 298 / 431 bins at 22.05 kHz, 479 at 24 kHz, 616 at
      // 32 kHz, 728 at 44.1 kHz, 751 at 48 kHz) are taken by selection; whatever it refuses (ties, a lane with more than
      // three candidates) and every other width by extraction rounds on 7 or 12 registers.
      bool done = false;
      if (k >= 4) {
        if (n > 704) done = contrast_select<12>(prow, lane, lo, n, k, spk, svl);
        else if (n > 576 && n <= 640) done = contrast_select<10>(prow, lane, lo, n, k, spk, svl);
        else if (n > 448 && n <= 512) done = contrast_select<8>(prow, lane, lo, n, k, spk, svl);
        else if (n > 384 && n <= 448) done = contrast_select<7>(prow, lane, lo, n, k, spk, svl);
        else if (n > 256 && n <= 320) done = contrast_select<5>(prow, lane, lo, n, k, spk, svl);
      }
      if (!done) {
        if (n <= 448) contrast_extract<7>(prow, lane, lo, n, k, spk, svl);
        else if (may_park && ppos(hi - 1) >= 64 * 12) contrast_extract_lds<12>(prow, lane, lo, n, k, spk, svl);
        else contrast_extract<12>(prow, lane, lo, n, k, spk, svl);
      }
    }
    const float rk = frcp((float)k);
    return make_float2(spk * rk, svl * rk);
  }
  if (WIDE && n <= 64 * 24 && n >= 64 && k <= 64 && may_park && ppos(lo) >= 256) {
    float spk, svl;
    if (contrast_compact<24>(prow, lane, lo, n, k, spk, svl)) {
      const float rk = frcp((float)k);
      return make_float2(spk * rk, svl * rk);
    }
  }
  const uint32_t tlo = row_kth(prow, lane, lo, n, k, false), thi = row_kth(prow, lane, lo, n, k, true);
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
  clo = wave_sum_i(clo); chi = wave_sum_i(chi);
  return make_float2((shi + (float)(k - chi) * sqrtf(__uint_as_float(thi))) / (float)k,
                     (slo + (float)(k - clo) * sqrtf(__uint_as_float(tlo))) / (float)k);
}

// Arguments of an out-of-line device function travel in VGPRs; these put the wave-uniform ones back into
// SGPRs so that the callee's loops and addresses stay scalar.
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ float uni(float v) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); }
template <typename P>
__device__ __forceinline__ P* uni(P* p) {
  const uint64_t v = (uint64_t)p;
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
  return (P*)(((uint64_t)hi << 32) | lo);
}


// The LEADING bands with k = 1 that end at or below bin 192 (C4: five of the seven bands, bins 0 ... 135) in ONE pass: the
// bins sit in three strided registers (bin lane + 64 r); band b's largest power goes to slot 2 b, the negative of its
// smallest to slot 2 b + 1 (so that all sixteen slots are MAX reductions), and the sixteen slots are reduced over the wave
// together by a butterfly that halves the number of live slots at every step: at the step with lane distance d the
// lanes with bit d clear keep the even slot of a pair and hand the odd one to their partner, and vice versa -- 56
// instructions for sixteen wave-wide reductions instead of sixteen times six.  Afterwards every lane holds the wave's
// result of slot (lane & 15); lane b fetches its band's two slots through the LDS crossbar.
// Returns the number of bands taken (0: fewer than two such bands, the caller's loop does everything).
template <int CTRL>
__device__ __forceinline__ float dpp_partner(float v) { return dpp_f<CTRL>(v); }
__device__ __forceinline__ float xor4_partner(float v) {       // lane ^ 4 inside a row: two bank-masked row shifts
  int t = __builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), 0x114 /* row_shr:4 */, 0xF, 0xA, false);
  t = __builtin_amdgcn_update_dpp(t, __float_as_int(v), 0x104 /* row_shl:4 */, 0xF, 0x5, false);
  return __int_as_float(t);
}
__device__ __forceinline__ int contrast_narrow_group(lds_row prow, int lane, int plo, int phi, int pk, int n_rows,
                                                     float& rp, float& rv) {
  // leading run of bands with k == 1 and hi <= 192 (lane = band in plo / phi / pk)
  const uint64_t okm = __ballot(lane < n_rows && pk == 1 && phi <= 192 && phi > plo);
  int nb = __ffsll((long long)~okm) - 1;                      // (~okm is never zero: lanes >= 16 are clear)
  nb = nb > 8 ? 8 : nb;
  if (nb < 2) return 0;
  lds_row pr = prow + ppos(lane);                              // bin lane + 64 r at ppos(lane) + 68 r
  const float q0 = pr[0], q1 = pr[68], q2 = pr[136];
  float v[16];
#pragma unroll
  for (int b = 0; b < 8; ++b) {
    float mx = -3.4e38f, mn = -3.4e38f;
    if (b < nb) {                                              // (wave-uniform)
      const int lo = __builtin_amdgcn_readlane(plo, b), n = __builtin_amdgcn_readlane(phi, b) - lo;
      const bool i0 = (unsigned)(lane - lo) < (unsigned)n, i1 = (unsigned)(lane + 64 - lo) < (unsigned)n,
                 i2 = (unsigned)(lane + 128 - lo) < (unsigned)n;
      mx = fmaxf(fmaxf(i0 ? q0 : mx, i1 ? q1 : mx), i2 ? q2 : mx);
      mn = fmaxf(fmaxf(i0 ? -q0 : mn, i1 ? -q1 : mn), i2 ? -q2 : mn);
    }
    v[2 * b] = mx; v[2 * b + 1] = mn;
  }
  const bool b0 = lane & 1, b1 = lane & 2, b2 = lane & 4, b3 = lane & 8;
  float w[8], x[4], y[2];
#pragma unroll
  for (int j = 0; j < 8; ++j)
    w[j] = fmaxf(b0 ? v[2 * j + 1] : v[2 * j], dpp_partner<DPP_QP_1032>(b0 ? v[2 * j] : v[2 * j + 1]));
#pragma unroll
  for (int j = 0; j < 4; ++j)
    x[j] = fmaxf(b1 ? w[2 * j + 1] : w[2 * j], dpp_partner<DPP_QP_2301>(b1 ? w[2 * j] : w[2 * j + 1]));
#pragma unroll
  for (int j = 0; j < 2; ++j) y[j] = fmaxf(b2 ? x[2 * j + 1] : x[2 * j], xor4_partner(b2 ? x[2 * j] : x[2 * j + 1]));
  float z = fmaxf(b3 ? y[1] : y[0], dpp_partner<0x128 /* row_ror:8 = lane ^ 8 */>(b3 ? y[0] : y[1]));
  // the four rows: lane ^ 16, then lane ^ 32
  {
    const auto r16 = __builtin_amdgcn_permlane16_swap(__float_as_uint(z), __float_as_uint(z), false, false);
    z = fmaxf(z, __uint_as_float((lane & 16) ? r16[0] : r16[1]));
    const auto r32 = __builtin_amdgcn_permlane32_swap(__float_as_uint(z), __float_as_uint(z), false, false);
    z = fmaxf(z, __uint_as_float((lane & 32) ? r32[0] : r32[1]));
  }
  // slot s sits in every lane with (lane & 15) == s: lane b takes slots 2 b and 2 b + 1
  const int src = (2 * lane) & 15;
  const float pmax = __int_as_float(__builtin_amdgcn_ds_bpermute(4 * src, __float_as_int(z)));
  const float nmin = __int_as_float(__builtin_amdgcn_ds_bpermute(4 * (src + 1), __float_as_int(z)));
  rp = (lane < nb) ? fsqrt(pmax) : rp;
  rv = (lane < nb) ? fsqrt(-nmin) : rv;
  return nb;
}

// All contrast bands of one row in ONE out-of-line call (a call per band paid the entry / exit sequence and the
// argument traffic seven times).  Band r's (peak, valley) tail means come back in lane r of the two result registers;
// the caller stores them.  The plan (lo, hi, k per band) is read from its LDS copy: one read per array, lane = band.
typedef const __attribute__((address_space(3))) int* lds_iptr;
template <int PS, bool WIDE>
__device__ __forceinline__ float2 row_contrast_body(lds_row prow, int lane, lds_iptr cpl, int n_rows_v, int may_park_v) {
  const int n_rows = uni(n_rows_v), may_park = uni(may_park_v);
  const int lb = lane & (SYG_MAX_BANDS - 1);
  const int plo = cpl[lb], phi = cpl[SYG_MAX_BANDS + lb], pk = cpl[2 * SYG_MAX_BANDS + lb];
  float rp = 0.f, rv = 0.f;
  const int r0 = contrast_narrow_group(prow, lane, plo, phi, pk, n_rows, rp, rv);
  for (int r = r0; r < n_rows; ++r) {
    const int lo = __builtin_amdgcn_readlane(plo, r), hi = __builtin_amdgcn_readlane(phi, r),
              k = __builtin_amdgcn_readlane(pk, r);
    const float2 pv = band_contrast<WIDE>(prow, lane, lo, hi, k, may_park);
    rp = (lane == r) ? pv.x : rp;
    rv = (lane == r) ? pv.y : rv;
  }
  if (PS > 0) {                                  // (the row holds 4^PS |X|^2: every magnitude came out 2^PS times too large)
    constexpr float MSC = 1.f / (float)(1 << PS);
    rp *= MSC; rv *= MSC;
  }
  return make_float2(rp, rv);
}
template <int NBIN, int PS>
__device__ __noinline__ float row_stats(lds_row prow, int lane, float binhz, float roll_percent, float bw_p, int smask) {
  return row_stats_body<NBIN, PS>(prow, lane, binhz, roll_percent, bw_p, smask);
}
template <int PS, bool WIDE = false>
__device__ __noinline__ float2 row_contrast_all(lds_row prow, int lane, lds_iptr cpl, int n_rows_v, int may_park_v) {
  return row_contrast_body<PS, WIDE>(prow, lane, cpl, n_rows_v, may_park_v);
}
// Statistics AND contrast of one row in one call (the C4 block asks for both: one entry / exit sequence, one wait for
// the outstanding memory operations, instead of two).  x: the statistics register of row_stats, y / z: peak / valley.
template <int NBIN, int PS>
__device__ __noinline__ float3 row_features(lds_row prow, int lane, float binhz, float roll_percent, float bw_p, int smask,
                                            lds_iptr cpl, int n_rows_v, int may_park_v) {
  const float s = row_stats_body<NBIN, PS>(prow, lane, binhz, roll_percent, bw_p, smask);
  const float2 pv = row_contrast_body<PS, (NBIN > 1025)>(prow, lane, cpl, n_rows_v, may_park_v);
  return make_float3(s, pv.x, pv.y);
}
