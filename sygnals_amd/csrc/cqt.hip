// Constant-Q transform building blocks (librosa.cqt as called by compute_cqt, sygnals/core/dsp.py:276-284):
//   decimate2_kernel   y[n] = scale * sum_j h[j] x[2n + half - j]   (the FIR decimation between octaves; taps are
//                      supplied by the caller -- see DESIGN.md for the resampler deviation)
//   cqt_octave_kernel  one octave: rectangular-window STFT frame (n_fft = 2^k) -> rfft in LDS -> n_filt complex dot
//                      products with the frequency-domain constant-Q basis -> out[b, row0 + f, t]
// One 64-lane wave per frame, FRAMES_PER_WG frames per workgroup (their outputs are adjacent in t).
#include "common.h"
#include <stdlib.h>

namespace syg {
namespace {

constexpr int FPW = 8;            // frames per workgroup
constexpr int MAXFILT = 24;       // filters per octave (bins_per_octave <= 24 handled in registers)

// FIR decimation by 2:  y[n] = scale * sum_j h[j] x[2n + half - j],  half = (ntaps - 1) / 2.
// A workgroup produces DEC_NBO consecutive outputs: the input run it needs is staged once in LDS, split into its
// even and odd samples (E[m] = x[2m], O[m] = x[2m+1]: the stride-2 reads of the filter become contiguous,
// conflict-free LDS reads; samples outside [0, L) are staged as zeros), the taps sit in LDS too, and every thread
// accumulates DEC_OUTS outputs per tap read.  Taps are applied in index order (same sum order as a direct loop).
constexpr int DEC_NT = 256, DEC_OUTS = 8, DEC_NBO = DEC_NT * DEC_OUTS;

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
  // the fast path stages sample pairs with 8-byte loads: rows and the base pointer must be 8-byte aligned
  const bool pair_ok = ((ldx & 1) == 0) && ((((uintptr_t)x) & 7) == 0);
  for (int j = tid; j < ntaps; j += DEC_NT) hs[j] = taps[j];
  for (int64_t n0 = (int64_t)blockIdx.x * DEC_NBO; n0 < Lout; n0 += (int64_t)gridDim.x * DEC_NBO) {
    // lowest input index used: 2 n0 + half - (ntaps - 1) = 2 n0 - half; mbase = floor(that / 2)
    const int64_t ilo = 2 * n0 - half;
    const int64_t mbase = (ilo >= 0) ? ilo / 2 : -((-ilo + 1) / 2);
    __syncthreads();                                  // the previous round's reads are done
    // staging: sample pairs (x[2m], x[2m+1]) as one 8-byte load per lane where the whole run lies inside the signal
    // and rows are 8-byte aligned (interior workgroups: no bounds checks); element-wise with zero fill otherwise
    const bool fast = pair_ok && mbase >= 0 && 2 * (mbase + nstage) <= L;
    if (fast) {
      const float2* xp = reinterpret_cast<const float2*>(xb) + mbase;
      for (int u = tid; u < nstage; u += DEC_NT) {
        const float2 v = xp[u];
        E[u] = v.x;
        O[u] = v.y;
      }
    } else {
      for (int u = tid; u < nstage; u += DEC_NT) {
        const int64_t i0 = 2 * (mbase + u);
        E[u] = (i0 >= 0 && i0 < L) ? xb[i0] : 0.f;
        O[u] = (i0 + 1 >= 0 && i0 + 1 < L) ? xb[i0 + 1] : 0.f;
      }
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

// The same filter for a fixed tap count (NT = 41: the CQT's decimator), two ADJACENT outputs per lane per round: the
// windows E[n - H2 .. n + H2 + 1], O[n - H2 .. n + H2] the pair needs are read as 8-byte words (lane stride 8 B:
// conflict-free) and kept in registers -- 12 to 22 LDS reads per two outputs instead of one read per tap and output.
// A half-band filter (every tap at an even offset from the centre is zero, the centre excepted) needs only the odd
// samples and the two centre ones: detected once per workgroup from the taps themselves.  The odd-offset taps are
// applied first (ascending index), then the even-offset ones.
template <int NT>
__global__ __launch_bounds__(DEC_NT, 8) void decimate2_fixed_kernel(const float* __restrict__ x, int64_t L, int64_t ldx,
                                                                 const float* __restrict__ taps, float scale,
                                                                 float* __restrict__ y, int64_t Lout, int64_t ldy) {
  constexpr int HALF = (NT - 1) / 2;                 // 20: even
  constexpr int H2 = HALF / 2;                       // 10
  static_assert(HALF % 2 == 0, "specialised for an even half length");
  constexpr int PAIRS = DEC_OUTS / 2;                // adjacent output pairs per thread
  constexpr int NSTAGE = DEC_NBO + HALF + 2;
  __shared__ __attribute__((aligned(16))) float hs[(NT + 3) & ~3];
  __shared__ __attribute__((aligned(16))) float E[NSTAGE + 2];
  __shared__ __attribute__((aligned(16))) float O[NSTAGE + 2];
  __shared__ int hb_flag;
  const int64_t b = blockIdx.y;
  const float* xb = x + b * ldx;
  const int tid = threadIdx.x;
  const bool pair_ok = ((ldx & 1) == 0) && ((((uintptr_t)x) & 7) == 0);
  for (int j = tid; j < NT; j += DEC_NT) hs[j] = taps[j];
  __syncthreads();
  if (tid == 0) {
    int hb = 1;
    for (int j = 0; j < NT; ++j)
      if (((HALF - j) & 1) == 0 && j != HALF && hs[j] != 0.f) hb = 0;
    hb_flag = hb;
  }
  __syncthreads();
  const bool halfband = hb_flag != 0;
  // the taps live in scalar registers for the whole kernel (wave-uniform values: one LDS read each, once)
  float hreg[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) hreg[j] = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(hs[j])));
  for (int64_t n0 = (int64_t)blockIdx.x * DEC_NBO; n0 < Lout; n0 += (int64_t)gridDim.x * DEC_NBO) {
    const int64_t mbase = n0 - H2;                   // staged pair u holds (x[2 (mbase + u)], x[2 (mbase + u) + 1])
    __syncthreads();
    const bool fast = pair_ok && mbase >= 0 && 2 * (mbase + NSTAGE) <= L;
    if (fast) {
      const float2* xp = reinterpret_cast<const float2*>(xb) + mbase;
      for (int u = tid; u < NSTAGE; u += DEC_NT) {
        const float2 v = xp[u];
        E[u] = v.x;
        O[u] = v.y;
      }
    } else {
      for (int u = tid; u < NSTAGE; u += DEC_NT) {
        const int64_t i0 = 2 * (mbase + u);
        E[u] = (i0 >= 0 && i0 < L) ? xb[i0] : 0.f;
        O[u] = (i0 + 1 >= 0 && i0 + 1 < L) ? xb[i0 + 1] : 0.f;
      }
    }
    __syncthreads();
#pragma unroll
    for (int p = 0; p < PAIRS; ++p) {
      const int nl = 2 * tid + p * 2 * DEC_NT;       // local index of the pair's first output; its window starts at nl
      // x[2 n + c], c = HALF - j:  c even -> E[n + c/2] (staged at nl + H2 + c/2),  c odd -> O[n + (c-1)/2]
      // one register window serves both parities in turn (the odd samples first, then the even ones): the kernel then
      // fits 64 VGPRs and can share a CU with the octave products running beside it on another stream
      float wv[2 * H2 + 2];
      const float2* o2 = reinterpret_cast<const float2*>(O + nl);
      const float2* e2 = reinterpret_cast<const float2*>(E + nl);
#pragma unroll
      for (int k = 0; k <= H2; ++k) { const float2 v = o2[k]; wv[2 * k] = v.x; wv[2 * k + 1] = v.y; }
      float a0 = 0.f, a1 = 0.f;
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const int c = HALF - j;
        if ((c & 1) != 0) {
          const int q = H2 + (c - 1) / 2;            // (c - 1) / 2 = floor(c / 2) for odd c of either sign
          a0 = fmaf(hreg[j], wv[q], a0);
          a1 = fmaf(hreg[j], wv[q + 1], a1);
        }
      }
      if (halfband) {
        static_assert(H2 % 2 == 0, "centre pair must be 8-byte aligned");
        const float2 v = e2[H2 / 2];                 // E[nl + H2], E[nl + H2 + 1]: the two centre samples
        a0 = fmaf(hreg[HALF], v.x, a0);
        a1 = fmaf(hreg[HALF], v.y, a1);
      } else {
#pragma unroll
        for (int k = 0; k <= H2; ++k) { const float2 v = e2[k]; wv[2 * k] = v.x; wv[2 * k + 1] = v.y; }
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          const int c = HALF - j;
          if ((c & 1) == 0) {
            const int q = H2 + c / 2;                // window index of E[n + c/2]
            a0 = fmaf(hreg[j], wv[q], a0);
            a1 = fmaf(hreg[j], wv[q + 1], a1);
          }
        }
      }
      const int64_t n = n0 + nl;
      if (n < Lout) y[b * ldy + n] = a0 * scale;
      if (n + 1 < Lout) y[b * ldy + n + 1] = a1 * scale;
    }
  }
}

// Two to four decimations in one pass (the CQT walks down an octave per decimation and needs every level): the chain
// x -> y1 -> y2 -> y3 moves 2 x the bytes of its largest member when every level is a launch of its own, because each
// level is written and read back; here a workgroup carries a tile through all levels in LDS and writes each level once.
// A tile owns NF outputs of the last level and the 2 NF / 4 NF outputs of the levels above; what the next level's
// filter needs beyond them (20 samples per side and level) is recomputed from a wider input run (+7 % reads at NF = 478,
// three levels; +20 % at four, used where the input is already small).  Every value is produced by the same instructions in the same order as by decimate2_fixed_kernel and
// passes to the next level as the float it would have been stored as; positions outside [0, L_level) are zero, as the
// zero padding of a level-by-level chain makes them: identical bits.
// Level s (s = 1 .. NL) computes CNT_s outputs from the level below, which sits in LDS split into even / odd samples
// (pair u = samples 2 (a_s - 10 + u), + 1; a_s = first output index of the tile at level s, always even):
//   CNT_NL = NF,   CNT_{s-1} = 2 CNT_s + 44  (the 8-byte window reads of the last pair touch two pairs more than the taps)
__device__ __forceinline__ int64_t uniform64(int64_t v) {          // a wave-uniform value back into scalar registers
  const int lo = __builtin_amdgcn_readfirstlane((int)(v & 0xffffffff)), hi = __builtin_amdgcn_readfirstlane((int)(v >> 32));
  return ((int64_t)hi << 32) | (uint32_t)lo;
}

template <int NL>
struct DecChain {
  static constexpr int NF = (NL == 4) ? 206 : (NL == 3) ? 478 : 990;   // CNT_1 / 2 = 978 / 1022 / 1012 pairs: four rounds of 256 lanes
  static constexpr int cnt(int s) { return s == NL ? NF : 2 * cnt(s + 1) + 44; }
  static constexpr int lds_floats() { int t = 0; for (int s = 0; s < NL; ++s) t += cnt(s) + 4; return t; }
};

struct DecOut {
  float* y[4];
  int64_t ld[4];
  int64_t len[5];                                // len[0] = L (input), len[s] = length of level s
};

template <int NT, int NL>
#ifndef SYG_DEC_WAVES
#define SYG_DEC_WAVES 4
#endif
__global__ __launch_bounds__(DEC_NT, SYG_DEC_WAVES) void decimate2_chain_kernel(const float* __restrict__ x, int64_t ldx,
                                                                const float* __restrict__ taps, float scale, DecOut o,
                                                                int64_t ntiles) {
  constexpr int HALF = (NT - 1) / 2, H2 = HALF / 2;
  static_assert(HALF % 4 == 0, "specialised for a half length that is a multiple of four");
  using DC = DecChain<NL>;
  __shared__ __attribute__((aligned(16))) float hs[(NT + 3) & ~3];
  __shared__ __attribute__((aligned(16))) float buf[DC::lds_floats()];
  __shared__ int hb_flag;
  __shared__ DecOut ol;                          // the per-level arguments, fetched where a level starts (they would
                                                 // otherwise sit in ~20 scalar registers for the whole kernel and spill)
  const int64_t b = blockIdx.y;
  const float* xb = x + b * ldx;
  const int tid = threadIdx.x;
  // the fast path moves the input run with 16-byte loads (an 8-byte load per lane runs at 0.54-0.70 of that rate --
  // MI355X_MICROARCH.md, table of load flavours -- and this pass is bound by its 1.3 GB of traffic): tile starts are
  // multiples of four samples by construction (NF << NL and HALF (2^NL - 1) both are), so every row start must be
  // 16-byte aligned too: ldx % 4 == 0 and a 16-byte aligned base (rows with ldx % 4 == 2, e.g. 22 050-sample clips,
  // take the element-wise path)
  const bool pair_ok = ((ldx & 3) == 0) && ((((uintptr_t)x) & 15) == 0);
  for (int j = tid; j < NT; j += DEC_NT) hs[j] = taps[j];
  if (tid == 0) ol = o;
  __syncthreads();
  if (tid == 0) {
    int hb = 1;
    for (int j = 0; j < NT; ++j)
      if (((HALF - j) & 1) == 0 && j != HALF && hs[j] != 0.f) hb = 0;
    hb_flag = hb;
  }
  __syncthreads();
  const bool halfband = hb_flag != 0;
  const int64_t len0 = o.len[0];
  // the taps at odd offsets from the centre, and the centre, live in scalar registers (all a half-band filter has);
  // the other even-offset taps of a general filter are read from LDS where they are used
  float hodd[HALF];
#pragma unroll
  for (int m = 0; m < HALF; ++m) hodd[m] = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(hs[2 * m + 1])));
  const float hcen = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(hs[HALF])));

  // consecutive tiles go to the same XCD (workgroup i runs on XCD i % 8) at about the same time: the input samples two
  // neighbours both need then come from that XCD's L2 the second time
  const int64_t per_x = (ntiles + 7) / 8;
  constexpr int NP0 = DC::cnt(0) / 2;
  static_assert(NP0 % 2 == 0 && ((DC::NF << NL) % 4) == 0 && ((HALF * ((1 << NL) - 1)) % 4) == 0, "16-byte input runs");
  constexpr int NQ0 = NP0 / 2;                    // 16-byte pieces (two pairs) of the input run
  constexpr int NR0 = (NQ0 + DEC_NT - 1) / DEC_NT;
  constexpr int A0OFF = HALF * ((1 << NL) - 1);  // a[0] = 2^NL a[NL] - HALF (2^NL - 1)
#ifdef SYG_DEC_NOXCD
  auto tile_of = [&](int64_t wg) { return wg; };
#else
  auto tile_of = [&](int64_t wg) { return (wg & 7) * per_x + (wg >> 3); };
#endif
  auto is_fast = [&](int64_t tile) {
    const int64_t a0 = tile * ((int64_t)DC::NF << NL) - A0OFF;
    return pair_ok && a0 >= 0 && a0 + 2 * NP0 <= len0;
  };
  // the input run of the NEXT tile is requested before the levels of the current one are computed (a workgroup
  // without loads in flight for three levels' worth of arithmetic leaves the memory system idle)
  float4 v[NR0];
  auto request = [&](int64_t tile) {
    const float4* xp = reinterpret_cast<const float4*>(xb + (tile * ((int64_t)DC::NF << NL) - A0OFF));
#pragma unroll
    for (int r = 0; r < NR0; ++r) {
      const int u = tid + r * DEC_NT;
      if (r + 1 < NR0 || u < NQ0) v[r] = xp[u];
    }
  };
  int64_t wg = blockIdx.x;
  while (wg < per_x * 8 && tile_of(wg) >= ntiles) wg += gridDim.x;
  if (wg < per_x * 8 && is_fast(tile_of(wg))) request(tile_of(wg));
  while (wg < per_x * 8) {
    const int64_t tile = tile_of(wg);
    // first output index of this tile at every level
    int64_t a[NL + 1];
    a[NL] = tile * DC::NF;
#pragma unroll
    for (int s = NL; s >= 1; --s) a[s - 1] = 2 * a[s] - HALF;
    __syncthreads();                             // the previous tile's readers are done
    {   // level 0: CNT_0 samples from a[0] as pairs
      float* E = buf;
      float* O = buf + NP0 + 2;
      if (is_fast(tile)) {
#pragma unroll
        for (int r = 0; r < NR0; ++r) {
          const int u = tid + r * DEC_NT;               // piece u = pairs 2 u, 2 u + 1
          if (r + 1 < NR0 || u < NQ0) {
            reinterpret_cast<float2*>(E)[u] = make_float2(v[r].x, v[r].z);
            reinterpret_cast<float2*>(O)[u] = make_float2(v[r].y, v[r].w);
          }
        }
      } else {                                     // a tile at either end of the signal
#pragma unroll 1
        for (int u = tid; u < NP0; u += DEC_NT) {
          const int64_t i0 = a[0] + 2 * u;
          E[u] = (i0 >= 0 && i0 < len0) ? xb[i0] : 0.f;
          O[u] = (i0 + 1 >= 0 && i0 + 1 < len0) ? xb[i0 + 1] : 0.f;
        }
      }
    }
    wg += gridDim.x;
    while (wg < per_x * 8 && tile_of(wg) >= ntiles) wg += gridDim.x;
    if (wg < per_x * 8 && is_fast(tile_of(wg))) request(tile_of(wg));
    int off = 0;
#pragma unroll
    for (int s = 1; s <= NL; ++s) {
      __syncthreads();
      const int npin = DC::cnt(s - 1) / 2;         // staged pairs of the level below
      const float* E = buf + off;
      const float* O = buf + off + npin + 2;
      off += DC::cnt(s - 1) + 4;
      const int npout = DC::cnt(s) / 2;            // output pairs of this level
      float* En = buf + off;                       // next level's even / odd arrays (unused at the last level)
      float* On = buf + off + npout + 2;
      // everything about this tile and level as wave-uniform 32-bit numbers relative to a[s]
      const int64_t Ls = uniform64(ol.len[s]);
      const int64_t own0 = tile * ((int64_t)DC::NF << (NL - s));
      const int span = DC::NF << (NL - s);
      const int64_t lo64 = -a[s], hi64 = Ls - a[s];           // outputs with local index in [vlo, vhi) exist
      const int vlo = lo64 > 0 ? (int)(lo64 < 2 * npout ? lo64 : 2 * npout) : 0;
      const int vhi = hi64 < 0 ? 0 : (int)(hi64 < 2 * npout ? hi64 : 2 * npout);
      const int w0 = (int)(own0 - a[s]);           // owned (written) local range [w0, w0 + span), cut to [vlo, vhi)
      const int wlo = w0 > vlo ? w0 : vlo, whi = w0 + span < vhi ? w0 + span : vhi;
      float* ybase = reinterpret_cast<float*>(uniform64((int64_t)(uintptr_t)ol.y[s - 1]));
      float* yo = ybase ? ybase + b * uniform64(ol.ld[s - 1]) + a[s] : nullptr;   // (only owned, existing indices are touched)
      const bool st2 = (((uintptr_t)yo) & 7) == 0;    // (a[s] is even: 8-byte aligned whenever the level's rows are)
#pragma unroll 1
      for (int pr = tid; pr < npout; pr += DEC_NT) {
        const int nl = 2 * pr;                     // local index of the pair's first output (window starts at pair nl)
        float wv[2 * H2 + 2];
        const float2* o2 = reinterpret_cast<const float2*>(O + nl);
#pragma unroll
        for (int k = 0; k <= H2; ++k) { const float2 v = o2[k]; wv[2 * k] = v.x; wv[2 * k + 1] = v.y; }
        float a0 = 0.f, a1 = 0.f;
#pragma unroll
        for (int m = 0; m < HALF; ++m) {           // taps j = 2 m + 1 (odd offset c = HALF - j from the centre), ascending
          const int q = H2 + (HALF - (2 * m + 1) - 1) / 2;
          a0 = fmaf(hodd[m], wv[q], a0);
          a1 = fmaf(hodd[m], wv[q + 1], a1);
        }
        if (halfband) {
          const float2 v = *reinterpret_cast<const float2*>(E + nl + H2);
          a0 = fmaf(hcen, v.x, a0);
          a1 = fmaf(hcen, v.y, a1);
        } else {                                   // a general filter: the even-offset taps, ascending, from LDS
#pragma unroll 1
          for (int j = 0; j < NT; j += 2) {
            const int q = nl + H2 + (HALF - j) / 2;
            const float h = hs[j];
            a0 = fmaf(h, E[q], a0);
            a1 = fmaf(h, E[q + 1], a1);
          }
        }
        const float v0 = (nl >= vlo && nl < vhi) ? a0 * scale : 0.f;
        const float v1 = (nl + 1 >= vlo && nl + 1 < vhi) ? a1 * scale : 0.f;
        if (s < NL) { En[pr] = v0; On[pr] = v1; }
        if (yo) {
          if (st2 && nl >= wlo && nl + 1 < whi) {          // both outputs owned: one 8-byte store
            *reinterpret_cast<float2*>(yo + nl) = make_float2(v0, v1);
          } else {
            if (nl >= wlo && nl < whi) yo[nl] = v0;
            if (nl + 1 >= wlo && nl + 1 < whi) yo[nl + 1] = v1;
          }
        }
      }
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

// ---------------------------------------------------------------------------------------------------------------
// One octave as a framed matrix product on the matrix cores.  The octave's response is linear in the frame:
//   out[f, t] = sum_k basis[f, k] * rfft(frame_t)[k] = sum_n frame_t[n] * g_f[n],   g_f[n] = sum_k basis[f, k] W_N^(k n)
// (g: the sparsified frequency-domain rows taken back to the time domain on the host, in float64), i.e.
//   OUT [2 n_filt, T] = G^T [2 n_filt, n_fft] x FRAMES [n_fft, T],   FRAMES[n, t] = y[t hop - n_fft/2 + n]
// with the real and imaginary parts of a filter as two rows.  v_mfma_f32_16x16x4_f32 (exact fp32): A = a 16-row tile
// of G^T, held in registers for the whole launch (n_fft / 4 VGPRs); B = 16 frames, each lane loading 16 bytes of its
// frame per four k-steps straight from global memory -- the k order inside a group of 16 samples is permuted so that a
// lane's float4 feeds four consecutive steps (A is packed with the same permutation on the host); frames overlap, so
// the re-reads hit L1 / L2.  No LDS, no barrier; a wave owns (row tile, frame tiles wave_id, wave_id + n_waves, ...),
// and the loads of the next frame tile are issued as soon as the MFMAs that read a register have been issued.
// RT row tiles per wave (n_rowtiles is a multiple of RT): with RT = 2 the two accumulator chains share every frame load
// and cover each other's MFMA latency.
template <int NFFT, int RT>
__global__ __launch_bounds__(256) void cqt_gemm_kernel(const float* __restrict__ ysig, int64_t L, int64_t ldy, int hop,
                                                       int64_t T, const float* __restrict__ gpacked, int n_rowtiles,
                                                       int n_filt, float2* __restrict__ out, int64_t out_bstride,
                                                       int row0, int waves_per_rowtile) {
  constexpr int S = NFFT / 16;                 // groups of 16 samples = 4 MFMA steps each
  typedef float v4f __attribute__((ext_vector_type(4)));
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4 + (threadIdx.x >> 6)));
  const int ngroups = n_rowtiles / RT;
  const int mt = (wid % ngroups) * RT, slot = wid / ngroups;
  if (slot >= waves_per_rowtile) return;
  const int64_t b = blockIdx.y;
  const float* yb = ysig + b * ldy;
  float a[RT][S * 4];
#pragma unroll
  for (int r = 0; r < RT; ++r) {
    const float* gp = gpacked + ((int64_t)(mt + r) * S * 4) * 64 + lane;
#pragma unroll
    for (int i = 0; i < S * 4; ++i) a[r][i] = gp[i * 64];
  }
  const int n = lane & 15, kk = lane >> 4;
  const int64_t ntiles = (T + 15) >> 4;
  // first output row (filter) of this lane: rows 16 mt + 4 kk + {0, 1} are (re, im) of filter f0, + {2, 3} of f0 + 1
  const int f0 = 8 * mt + 2 * kk;
  // groups [s0, s1) of frame tile `tile` into q: plain 16-byte loads when every frame of the tile lies inside the signal
  // (wave-uniform test), element-wise with zero fill otherwise (centre padding at both ends)
  auto load_part = [&](int64_t tile, float4 (&q)[S], int s0, int s1) {
    const int64_t t = tile * 16 + n;
    const int64_t base = t * (int64_t)hop - NFFT / 2 + 4 * kk;
    const int64_t lo = tile * 16 * (int64_t)hop - NFFT / 2, hi = (tile * 16 + 15) * (int64_t)hop + NFFT / 2;
    if (lo >= 0 && hi <= L) {
#pragma unroll
      for (int s = 0; s < S; ++s)
        if (s >= s0 && s < s1) q[s] = *reinterpret_cast<const float4*>(yb + base + 16 * s);
    } else {
#pragma unroll
      for (int s = 0; s < S; ++s)
        if (s >= s0 && s < s1) {
          const int64_t i0 = base + 16 * s;
          q[s].x = (i0 >= 0 && i0 < L) ? yb[i0] : 0.f;
          q[s].y = (i0 + 1 >= 0 && i0 + 1 < L) ? yb[i0 + 1] : 0.f;
          q[s].z = (i0 + 2 >= 0 && i0 + 2 < L) ? yb[i0 + 2] : 0.f;
          q[s].w = (i0 + 3 >= 0 && i0 + 3 < L) ? yb[i0 + 3] : 0.f;
        }
    }
  };
  float4 q[S];
  int64_t tile = slot;
  if (tile < ntiles) load_part(tile, q, 0, S);
  for (; tile < ntiles; tile += waves_per_rowtile) {
    v4f acc[RT];
#pragma unroll
    for (int r = 0; r < RT; ++r) acc[r] = v4f{0.f, 0.f, 0.f, 0.f};
    const int64_t nxt = tile + waves_per_rowtile;
    // two halves: as soon as the MFMAs of a half have been issued its registers are reloaded with the next tile's
    // samples, which then travel while the other half is multiplied -- no second buffer
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
      for (int s = h * (S / 2); s < (h + 1) * (S / 2); ++s) {
#pragma unroll
        for (int r = 0; r < RT; ++r) acc[r] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[r][4 * s + 0], q[s].x, acc[r], 0, 0, 0);
#pragma unroll
        for (int r = 0; r < RT; ++r) acc[r] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[r][4 * s + 1], q[s].y, acc[r], 0, 0, 0);
#pragma unroll
        for (int r = 0; r < RT; ++r) acc[r] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[r][4 * s + 2], q[s].z, acc[r], 0, 0, 0);
#pragma unroll
        for (int r = 0; r < RT; ++r) acc[r] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[r][4 * s + 3], q[s].w, acc[r], 0, 0, 0);
      }
      if (nxt < ntiles) load_part(nxt, q, h * (S / 2), (h + 1) * (S / 2));
    }
    const int64_t t = tile * 16 + n;
    if (t < T) {
#pragma unroll
      for (int r = 0; r < RT; ++r) {
        const int f = f0 + 8 * r;
        float2* o = out + b * out_bstride + (int64_t)(row0 + f) * T + t;
        if (f < n_filt) o[0] = make_float2(acc[r][0], acc[r][1]);
        if (f + 1 < n_filt) o[T] = make_float2(acc[r][2], acc[r][3]);
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// The same framed product with both operands split into three bfloat16 terms (x = hi + mid + lo, 3 x 8 mantissa bits
// = the 24 of a float32) and the six products that carry more than 2^-24 of the result accumulated in fp32 on
// v_mfma_f32_16x16x32_bf16: a_hi b_hi + a_hi b_mid + a_mid b_hi + a_hi b_lo + a_lo b_hi + a_mid b_mid (the dropped
// terms are <= 3 x 2^-24 |a b|, the size of the fp32 rounding the single-instruction form makes).  Why: the fp32-input
// MFMA runs at 1/16 of the bf16 rate and, dense, holds the chip at ~1.5 GHz; six bf16 instructions of depth 32 do the
// work of sixteen fp32 instructions of depth 4 in 3/8 of the matrix-pipe time, and the splitting of the frame samples
// (11 vector instructions per two samples) runs on the otherwise idle vector pipe.  G^T is split on the host and sits
// in LDS (3 x RT x n_fft/32 x 64 lanes x 16 B = 48 KiB at n_fft 256), frames come straight from global memory as
// before; lane l of a step holds the 8 consecutive k = 8 (l >> 4) .. + 7 of its row / frame.
typedef __bf16 v8bf __attribute__((ext_vector_type(8)));
typedef float cq_v4f __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void split3_bf16(const float (&x)[8], v8bf& hi, v8bf& mid, v8bf& lo) {
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const __bf16 h = (__bf16)x[j];
    const float r = x[j] - (float)h;              // exact: h keeps the leading 8 bits of x
    const __bf16 m = (__bf16)r;
    hi[j] = h; mid[j] = m; lo[j] = (__bf16)(r - (float)m);
  }
}

template <int NFFT, int RT>
__global__ __launch_bounds__(256) void cqt_bf16x3_kernel(const float* __restrict__ ysig, int64_t L, int64_t ldy, int hop,
                                                         int64_t T, const uint4* __restrict__ gsplit, int n_filt,
                                                         float2* __restrict__ out, int64_t out_bstride, int row0,
                                                         int n_waves) {
  constexpr int S = NFFT / 32;                 // MFMA k-steps of 32 samples
  __shared__ uint4 atab[3 * RT * S * 64];       // [term][row tile][step][lane]: 8 bf16 each
  for (int i = threadIdx.x; i < 3 * RT * S * 64; i += 256) atab[i] = gsplit[i];
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4 + (threadIdx.x >> 6)));
  const int64_t b = blockIdx.y;
  const float* yb = ysig + b * ldy;
  const int n = lane & 15, kk = lane >> 4;
  const int64_t ntiles = (T + 15) >> 4;
  const int f0 = 2 * kk;                         // rows 4 kk + {0, 1}: (re, im) of filter f0, + {2, 3}: of f0 + 1
  auto load_part = [&](int64_t tile, float4 (&q)[2 * S], int s0, int s1) {
    const int64_t t = tile * 16 + n;
    const int64_t base = t * (int64_t)hop - NFFT / 2 + 8 * kk;
    const int64_t lo = tile * 16 * (int64_t)hop - NFFT / 2, hi = (tile * 16 + 15) * (int64_t)hop + NFFT / 2;
    if (lo >= 0 && hi <= L) {
#pragma unroll
      for (int s = 0; s < S; ++s)
        if (s >= s0 && s < s1) {
          q[2 * s] = *reinterpret_cast<const float4*>(yb + base + 32 * s);
          q[2 * s + 1] = *reinterpret_cast<const float4*>(yb + base + 32 * s + 4);
        }
    } else {
#pragma unroll
      for (int s = 0; s < S; ++s)
        if (s >= s0 && s < s1) {
          float v[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const int64_t i0 = base + 32 * s + j;
            v[j] = (i0 >= 0 && i0 < L) ? yb[i0] : 0.f;
          }
          q[2 * s] = make_float4(v[0], v[1], v[2], v[3]);
          q[2 * s + 1] = make_float4(v[4], v[5], v[6], v[7]);
        }
    }
  };
  float4 q[2 * S];
  int64_t tile = wid;
  if (tile < ntiles) load_part(tile, q, 0, S);
  for (; tile < ntiles; tile += n_waves) {
    cq_v4f acc[RT];
#pragma unroll
    for (int r = 0; r < RT; ++r) acc[r] = cq_v4f{0.f, 0.f, 0.f, 0.f};
    const int64_t nxt = tile + n_waves;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
      for (int s = h * (S / 2); s < (h + 1) * (S / 2); ++s) {
        const float x[8] = {q[2 * s].x, q[2 * s].y, q[2 * s].z, q[2 * s].w, q[2 * s + 1].x, q[2 * s + 1].y, q[2 * s + 1].z,
                            q[2 * s + 1].w};
        v8bf bh, bm, bl;
        split3_bf16(x, bh, bm, bl);
#pragma unroll
        for (int r = 0; r < RT; ++r) {
          const v8bf ah = *reinterpret_cast<const v8bf*>(&atab[((0 * RT + r) * S + s) * 64 + lane]);
          const v8bf am = *reinterpret_cast<const v8bf*>(&atab[((1 * RT + r) * S + s) * 64 + lane]);
          const v8bf al = *reinterpret_cast<const v8bf*>(&atab[((2 * RT + r) * S + s) * 64 + lane]);
          // small terms first
          acc[r] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh, acc[r], 0, 0, 0);
          acc[r] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl, acc[r], 0, 0, 0);
          acc[r] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bm, acc[r], 0, 0, 0);
          acc[r] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bh, acc[r], 0, 0, 0);
          acc[r] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bm, acc[r], 0, 0, 0);
          acc[r] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, acc[r], 0, 0, 0);
        }
        // one k-step at a time: without this fence the scheduler hoists the operand reads and splits of all eight
        // steps to the top (308 VGPRs); with it a step's temporaries die before the next step's are born
        __builtin_amdgcn_sched_barrier(0);
      }
      if (nxt < ntiles) load_part(nxt, q, h * (S / 2), (h + 1) * (S / 2));
    }
    const int64_t t = tile * 16 + n;
    if (t < T) {
#pragma unroll
      for (int r = 0; r < RT; ++r) {
        const int f = f0 + 8 * r;
        float2* o = out + b * out_bstride + (int64_t)(row0 + f) * T + t;
        if (f < n_filt) o[0] = make_float2(acc[r][0], acc[r][1]);
        if (f + 1 < n_filt) o[T] = make_float2(acc[r][2], acc[r][3]);
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Overlapping frames (hop <= n_fft / 2, the lower octaves: hop halves with every decimation while the filters keep
// their length, so a sample sits in 2 .. 64 frames): the kernel above splits a sample once per frame that holds it,
// and that splitting is most of its time.  Here a WAVE splits the contiguous sample run of its 16-frame tile
// ((16 - 1) hop + n_fft samples) once into three bfloat16 planes in an LDS region of its own and reads its frames' B
// operands from the planes with ds_read_b128 (8 consecutive samples = 16 B; hop % 8 == 4 keeps a second copy of the
// planes shifted by four samples, so that every frame start is 16-byte aligned in one of the two).  The 16-byte chunk i
// of the run sits at slot i + (i >> sh): lane (n, kk) of step s reads chunk n hop/8 + kk + 4 s, and the skew
// (sh = log2(hop / 8) - 1 for hop >= 32) spreads the 16 lanes of every ds_read_b128 lane group over the 16 bank groups
// (checked by enumeration for hop / 8 in {1, 2, 4, 8, 16, 32}; other hops work with 2-way conflicts).  The waves of a
// workgroup share only the operand table: no barrier after the first, each wave requests the samples of its next tile
// before the products of the current one and splits them afterwards.  As many waves per workgroup (one workgroup per
// CU) as the LDS holds regions.  Same operands, same MFMA order as cqt_bf16x3_kernel: the two produce identical bits.
#ifdef SYG_CQT_STAMP
__device__ unsigned long long cqs_stamp[4 * 4096];       // development build: per-wave wall-clock stamps (100 MHz)
#define CQS_STAMP(k) do { if (lane == 0 && blockIdx.x * nwv + w < 4096) { cqs_stamp[(blockIdx.x * nwv + w) * 4 + (k)] = wall_clock64(); if ((k) != 1) cqs_stamp[(blockIdx.x * nwv + w) * 4 + ((k) == 0 ? 1 : 3)] = clock64(); } } while (0)
#else
#define CQS_STAMP(k) do { } while (0)
#endif
// MAXC: 16-byte chunks per lane and tile (hop 128: (15 * 128 + 256) / 8 = 272 chunks <= 5 * 64)
template <int NFFT, int RT, int MAXC>
__global__ __launch_bounds__(MAXC == 5 ? 512 : MAXC == 3 ? 768 : 1024) void cqt_bf16x3_staged_kernel(const float* __restrict__ ysig, int64_t L, int64_t ldy,
                                                                 int hop, int64_t T, const uint4* __restrict__ gsplit,
                                                                 int n_filt, float2* __restrict__ out,
                                                                 int64_t out_bstride, int row0, int nchunks, int cplane,
                                                                 int ncopy, int sh) {
  constexpr int S = NFFT / 32;
  constexpr int NA = 3 * RT * S * 64;
  extern __shared__ __attribute__((aligned(16))) uint4 cqs_lds[];
  const int tid = threadIdx.x, nt = blockDim.x;
  uint4* atab = cqs_lds;                         // [term][row tile][step][lane]
  const int lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nwv = nt >> 6;
  CQS_STAMP(0);
  for (int i = tid; i < NA; i += nt) atab[i] = gsplit[i];
  const int plane = cplane * ncopy;
  const int ntot = nchunks * ncopy;
  uint4* stage = cqs_lds + NA + w * 3 * plane;   // this wave's [term][copy][slot]
  const int64_t b = blockIdx.y;
  const float* yb = ysig + b * ldy;
  const int n = lane & 15, kk = lane >> 4;
  const int64_t ntiles = (T + 15) >> 4;
  const int64_t tstride = (int64_t)gridDim.x * nwv;
  const int f0 = 2 * kk;

  float4 pre[2 * MAXC];
  auto fetch = [&](int64_t tile) {
    const int64_t s0 = tile * 16 * (int64_t)hop - NFFT / 2;
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
      const int idx = lane + 64 * c;
      if (idx < ntot) {
        const int cp = idx >= nchunks ? 1 : 0, i = idx - cp * nchunks;
        const int64_t g = s0 + 4 * cp + 8 * (int64_t)i;
        if (g >= 0 && g + 8 <= L) {
          pre[2 * c] = *reinterpret_cast<const float4*>(yb + g);
          pre[2 * c + 1] = *reinterpret_cast<const float4*>(yb + g + 4);
        } else {
          float v[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] = (g + j >= 0 && g + j < L) ? yb[g + j] : 0.f;
          pre[2 * c] = make_float4(v[0], v[1], v[2], v[3]);
          pre[2 * c + 1] = make_float4(v[4], v[5], v[6], v[7]);
        }
      }
    }
  };
  auto park = [&]() {
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
      if (c > 0) __builtin_amdgcn_sched_barrier(0);             // one chunk's temporaries at a time
      const int idx = lane + 64 * c;
      if (idx < ntot) {
        const int cp = idx >= nchunks ? 1 : 0, i = idx - cp * nchunks;
        const float x[8] = {pre[2 * c].x, pre[2 * c].y, pre[2 * c].z, pre[2 * c].w,
                            pre[2 * c + 1].x, pre[2 * c + 1].y, pre[2 * c + 1].z, pre[2 * c + 1].w};
        v8bf bh, bm, bl;
        split3_bf16(x, bh, bm, bl);
        const int slot = i + (i >> sh) + cp * cplane;
        *reinterpret_cast<v8bf*>(&stage[slot]) = bh;
        *reinterpret_cast<v8bf*>(&stage[plane + slot]) = bm;
        *reinterpret_cast<v8bf*>(&stage[2 * plane + slot]) = bl;
      }
    }
  };

  // wave w of workgroup g is wave w * gridDim.x + g of the grid: the waves that get one tile more than the others
  // (the first ntiles mod n_waves of them) are then spread over all CUs instead of filling the first few
  int64_t tile = (int64_t)w * gridDim.x + blockIdx.x;
  if (tile < ntiles) fetch(tile);
  __syncthreads();                               // the operand table is in place (the only workgroup barrier)
  const int fo = n * hop;                        // the frame's first sample inside the tile's run
  const int cp = ncopy == 2 ? (fo >> 2) & 1 : 0;
  const int i0 = ((fo - 4 * cp) >> 3) + kk;
  const int coff = cp * cplane;
  for (; tile < ntiles; tile += tstride) {
    park();
    wave_lds_sync();
    if (tile + tstride < ntiles) fetch(tile + tstride);
    cq_v4f acc[RT];
#pragma unroll
    for (int r = 0; r < RT; ++r) acc[r] = cq_v4f{0.f, 0.f, 0.f, 0.f};
    // operands of step s + 1 are requested from LDS before the matrix instructions of step s issue (two register
    // sets); within a step the two row tiles alternate, so consecutive matrix instructions are independent
    v8bf Ah[2][RT], Am[2][RT], Al[2][RT], Bh[2], Bm[2], Bl[2];
    auto operands = [&](int s, int q) {          // (read in the order the matrix instructions consume them)
      const int i = i0 + 4 * s;
      const int slot = i + (i >> sh) + coff;
      Bh[q] = *reinterpret_cast<const v8bf*>(&stage[slot]);
#pragma unroll
      for (int r = 0; r < RT; ++r) Al[q][r] = *reinterpret_cast<const v8bf*>(&atab[((2 * RT + r) * S + s) * 64 + lane]);
      Bl[q] = *reinterpret_cast<const v8bf*>(&stage[2 * plane + slot]);
#pragma unroll
      for (int r = 0; r < RT; ++r) Ah[q][r] = *reinterpret_cast<const v8bf*>(&atab[((0 * RT + r) * S + s) * 64 + lane]);
      Bm[q] = *reinterpret_cast<const v8bf*>(&stage[plane + slot]);
#pragma unroll
      for (int r = 0; r < RT; ++r) Am[q][r] = *reinterpret_cast<const v8bf*>(&atab[((1 * RT + r) * S + s) * 64 + lane]);
    };
    operands(0, 0);
#pragma unroll
    for (int s = 0; s < S; ++s) {
      const int q = s & 1;
      if (s + 1 < S) operands(s + 1, q ^ 1);
      // small terms first (per accumulator the same order as cqt_bf16x3_kernel)
#pragma unroll
      for (int r = 0; r < RT; ++r) acc[r] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Al[q][r], Bh[q], acc[r], 0, 0, 0);
#pragma unroll
      for (int r = 0; r < RT; ++r) acc[r] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Ah[q][r], Bl[q], acc[r], 0, 0, 0);
#pragma unroll
      for (int r = 0; r < RT; ++r) acc[r] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Am[q][r], Bm[q], acc[r], 0, 0, 0);
#pragma unroll
      for (int r = 0; r < RT; ++r) acc[r] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Am[q][r], Bh[q], acc[r], 0, 0, 0);
#pragma unroll
      for (int r = 0; r < RT; ++r) acc[r] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Ah[q][r], Bm[q], acc[r], 0, 0, 0);
#pragma unroll
      for (int r = 0; r < RT; ++r) acc[r] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Ah[q][r], Bh[q], acc[r], 0, 0, 0);
      // issue order of the step: one LDS read of the next step behind each of the first matrix instructions
      if (s + 1 < S) {
#pragma unroll
        for (int k = 0; k < 3 + 3 * RT; ++k) {
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, 6 * RT - (3 + 3 * RT), 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    wave_lds_sync();                             // the planes have been read before the next tile's are written
    const int64_t t = tile * 16 + n;
    if (t < T) {
#pragma unroll
      for (int r = 0; r < RT; ++r) {
        const int f = f0 + 8 * r;
        float2* o = out + b * out_bstride + (int64_t)(row0 + f) * T + t;
        if (f < n_filt) o[0] = make_float2(acc[r][0], acc[r][1]);
        if (f + 1 < n_filt) o[T] = make_float2(acc[r][2], acc[r][3]);
      }
    }
  }
  CQS_STAMP(2);
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
  if (ntaps == 41)
    hipLaunchKernelGGL(decimate2_fixed_kernel<41>, dim3((unsigned)blocks, (unsigned)B), dim3(DEC_NT), 0, (hipStream_t)stream,
                       x, L, ldx, taps, scale, y, Lout, ldy);
  else
    hipLaunchKernelGGL(decimate2_kernel, dim3((unsigned)blocks, (unsigned)B), dim3(DEC_NT), lds, (hipStream_t)stream, x, L,
                       ldx, taps, ntaps, scale, y, Lout, ldy);
  SYG_CHECK_LAUNCH("decimate2");
  return SYG_OK;
}

extern "C" int syg_decimate2_chain_f32(const float* x, int64_t B, int64_t L, int64_t ldx, const float* taps, int ntaps,
                                       float scale, int levels, float* const* y, const int64_t* ldy, void* stream) {
  SYG_REQUIRE(x && taps && y && ldy, "decimate2_chain: null pointer argument");
  SYG_REQUIRE(B >= 1 && B <= 65535 && L >= 1 && ldx >= L, "decimate2_chain: bad B/L/ldx");
  SYG_REQUIRE(levels >= 1 && levels <= 4, "decimate2_chain: levels must be 1 ... 4 (got %d)", levels);
  SYG_REQUIRE(ntaps >= 1 && (ntaps & 1) == 1 && ntaps <= 1025, "decimate2_chain: ntaps must be odd and <= 1025");
  SYG_REQUIRE(y[levels - 1], "decimate2_chain: the last level needs an output buffer");
  DecOut o;
  o.len[0] = L;
  for (int s = 0; s < 4; ++s) { o.y[s] = nullptr; o.ld[s] = 0; o.len[s + 1] = 0; }
  for (int s = 0; s < levels; ++s) {
    o.len[s + 1] = (o.len[s] + 1) / 2;
    o.y[s] = y[s];
    o.ld[s] = ldy[s];
    SYG_REQUIRE(!y[s] || ldy[s] >= o.len[s + 1], "decimate2_chain: ldy[%d] too small", s);
  }
  if (ntaps != 41 || levels == 1) {               // level by level (every level then needs its buffer)
    const float* src = x;
    int64_t ls = ldx;
    for (int s = 0; s < levels; ++s) {
      SYG_REQUIRE(y[s], "decimate2_chain: with %d taps every level needs an output buffer", ntaps);
      const int rc = syg_decimate2_f32(src, B, o.len[s], ls, taps, ntaps, scale, y[s], ldy[s], stream);
      if (rc != SYG_OK) return rc;
      src = y[s];
      ls = ldy[s];
    }
    return SYG_OK;
  }
  int dev = 0, n_cu = 0;
  if (hipGetDevice(&dev) != hipSuccess ||
      hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n_cu <= 0) {
    set_error("decimate2_chain: cannot query the device");
    return SYG_E_LAUNCH;
  }
  const int nf = levels == 4 ? DecChain<4>::NF : levels == 3 ? DecChain<3>::NF : DecChain<2>::NF;
  const int64_t ntiles = (o.len[levels] + nf - 1) / nf;
  int64_t blocks = ((ntiles + 7) / 8) * 8;                      // a multiple of 8: a workgroup keeps its XCD
  const int64_t cap = (((int64_t)n_cu * SYG_DEC_WAVES + B - 1) / B + 7) / 8 * 8;
  if (blocks > cap) blocks = cap;
  const dim3 grid((unsigned)blocks, (unsigned)B), block(DEC_NT);
  if (levels == 4)
    hipLaunchKernelGGL((decimate2_chain_kernel<41, 4>), grid, block, 0, (hipStream_t)stream, x, ldx, taps, scale, o, ntiles);
  else if (levels == 3)
    hipLaunchKernelGGL((decimate2_chain_kernel<41, 3>), grid, block, 0, (hipStream_t)stream, x, ldx, taps, scale, o, ntiles);
  else
    hipLaunchKernelGGL((decimate2_chain_kernel<41, 2>), grid, block, 0, (hipStream_t)stream, x, ldx, taps, scale, o, ntiles);
  SYG_CHECK_LAUNCH("decimate2_chain");
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


extern "C" int syg_cqt_octave_gemm_f32(const float* y, int64_t B, int64_t L, int64_t ldy, int n_fft, int hop, int64_t T,
                                       const float* gpacked, int n_filt, float* out, int64_t out_bstride, int row0,
                                       void* stream) {
  SYG_REQUIRE(y && gpacked && out, "cqt_octave_gemm: null pointer argument");
  SYG_REQUIRE(n_fft == 128 || n_fft == 256 || n_fft == 512, "cqt_octave_gemm: n_fft must be 128, 256 or 512 (got %d)", n_fft);
  SYG_REQUIRE(B >= 1 && B <= 65535 && L >= 1 && ldy >= L && hop >= 1, "cqt_octave_gemm: bad B/L/ldy/hop");
  SYG_REQUIRE(T >= 1 && T <= 1 + L / hop, "cqt_octave_gemm: T=%lld exceeds the centred frame count %lld", (long long)T,
              (long long)(1 + L / hop));
  SYG_REQUIRE(n_filt >= 1 && n_filt <= 64, "cqt_octave_gemm: n_filt must be in [1, 64]");
  SYG_REQUIRE(row0 >= 0 && out_bstride >= (int64_t)(row0 + n_filt) * T, "cqt_octave_gemm: output rows out of range");
  const int n_rowtiles = (2 * n_filt + 15) / 16;
  const int64_t ntiles = (T + 15) / 16;
  // two row tiles per wave when they come in pairs and the operands fit the registers (n_fft <= 256: 128 + 64 VGPRs)
  const int rt = (n_rowtiles % 2 == 0 && n_fft <= 256) ? 2 : 1;
  const int ngroups = n_rowtiles / rt;
  // persistent waves: about 8 (rt = 2) / 12 per CU over the batch, never more than there are frame tiles
  int64_t wpr = (256 * (rt == 2 ? 8 : 12)) / ((int64_t)ngroups * B);
  if (wpr < 4) wpr = 4;
  if (wpr > ntiles) wpr = ntiles;
  const int64_t waves = wpr * ngroups;
  const dim3 grid((unsigned)((waves + 3) / 4), (unsigned)B), block(256);
  hipStream_t st = (hipStream_t)stream;
#define SYG_CQT_GEMM(N, R)                                                                                           \
  hipLaunchKernelGGL((cqt_gemm_kernel<N, R>), grid, block, 0, st, y, L, ldy, hop, T, gpacked, n_rowtiles, n_filt,    \
                     (float2*)out, out_bstride, row0, (int)wpr)
  if (n_fft == 128) { if (rt == 2) SYG_CQT_GEMM(128, 2); else SYG_CQT_GEMM(128, 1); }
  else if (n_fft == 256) { if (rt == 2) SYG_CQT_GEMM(256, 2); else SYG_CQT_GEMM(256, 1); }
  else SYG_CQT_GEMM(512, 1);
#undef SYG_CQT_GEMM
  SYG_CHECK_LAUNCH("cqt_octave_gemm");
  return SYG_OK;
}


extern "C" int syg_cqt_octave_bf16x3_f32(const float* y, int64_t B, int64_t L, int64_t ldy, int n_fft, int hop, int64_t T,
                                         const void* gsplit, int n_filt, float* out, int64_t out_bstride, int row0,
                                         void* stream) {
  SYG_REQUIRE(y && gsplit && out, "cqt_octave_bf16x3: null pointer argument");
  SYG_REQUIRE(n_fft == 128 || n_fft == 256, "cqt_octave_bf16x3: n_fft must be 128 or 256 (got %d)", n_fft);
  SYG_REQUIRE(B >= 1 && B <= 65535 && L >= 1 && ldy >= L && hop >= 1, "cqt_octave_bf16x3: bad B/L/ldy/hop");
  SYG_REQUIRE(T >= 1 && T <= 1 + L / hop, "cqt_octave_bf16x3: T=%lld exceeds the centred frame count %lld", (long long)T,
              (long long)(1 + L / hop));
  SYG_REQUIRE(n_filt >= 1 && n_filt <= 16, "cqt_octave_bf16x3: n_filt must be in [1, 16] (two 16-row tiles)");
  SYG_REQUIRE(row0 >= 0 && out_bstride >= (int64_t)(row0 + n_filt) * T, "cqt_octave_bf16x3: output rows out of range");
  SYG_REQUIRE(((uintptr_t)gsplit) % 16 == 0, "cqt_octave_bf16x3: operand table must be 16-byte aligned");
  const int rt = (2 * n_filt + 15) / 16;
  const int64_t ntiles = (T + 15) / 16;
  hipStream_t st = (hipStream_t)stream;
  // overlapping frames: every wave splits the sample run of its 16-frame tile once (cqt_bf16x3_staged_kernel)
  {
    const int staged_opt = option(SYG_OPT_CQT_STAGED);   // -1 default, 0 off, 2: also where hop = n_fft / 2 (for the tests)
    const bool off = staged_opt == 0;
    const bool shape_ok = hop % 4 == 0 && hop <= n_fft / 2;
    const int ncopy = hop % 8 == 0 ? 1 : 2;
    const int nchunks = shape_ok ? (15 * hop + n_fft) / 8 : 0;
    int sh = 31;                                   // slot skew: floor(log2(hop / 8)) - 1 for hop >= 32, none below
    if (hop >= 32) { const int h8 = hop >> 3; int lg = 0; while ((2 << lg) <= h8) ++lg; sh = lg - 1; }
    int cplane = sh < 31 ? nchunks + (nchunks >> sh) + 1 : nchunks;
    if (ncopy == 2) cplane = ((cplane + 7) & ~15) + 8;          // the second copy starts 8 bank groups over
    const size_t atab_bytes = (size_t)3 * rt * (n_fft / 32) * 64 * 16;
    const size_t wave_bytes = (size_t)3 * ncopy * cplane * 16;
    int nwv = shape_ok ? (int)((160 * 1024 - 512 - atab_bytes) / wave_bytes) : 0;       // regions the LDS holds
    const int maxc = (ncopy * nchunks + 63) / 64;
    // (hop = n_fft / 2, maxc 5: a sample sits in two frames only and the LDS holds 7 regions -- measured slower than the
    //  per-frame kernel, 78 vs 66 us per C5 octave; SYG_OPT_CQT_STAGED = 2 forces it for the tests)
    const bool worth = maxc <= 3 || staged_opt == 2;
    const int wcap = maxc > 3 ? 8 : maxc == 3 ? 12 : 16;       // the kernels' launch bounds (longer runs hold more registers)
    if (nwv > wcap) nwv = wcap;
    if (!off && worth && shape_ok && ((uintptr_t)y) % 16 == 0 && (B == 1 || ldy % 4 == 0) && ncopy * nchunks <= 5 * 64 &&
        nwv >= 4) {
      int dev = 0, n_cu = 0;
      if (hipGetDevice(&dev) != hipSuccess ||
          hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n_cu <= 0) {
        set_error("cqt_octave_bf16x3: cannot query the device");
        return SYG_E_LAUNCH;
      }
      const size_t lds = atab_bytes + (size_t)nwv * wave_bytes;
      int64_t gx = ((int64_t)n_cu + B - 1) / B;    // one workgroup per CU over the batch
      if (gx * nwv > ntiles) gx = (ntiles + nwv - 1) / nwv;
      if (gx < 1) gx = 1;
      const dim3 grid((unsigned)gx, (unsigned)B), block(nwv * 64);
#define SYG_CQT_ST(N, R, C)                                                                                          \
  do {                                                                                                               \
    hipError_t e2 = hipFuncSetAttribute((const void*)cqt_bf16x3_staged_kernel<N, R, C>,                              \
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                       \
    if (e2 != hipSuccess) { set_error("cqt_octave_bf16x3: cannot reserve LDS: %s", hipGetErrorString(e2)); return SYG_E_LAUNCH; } \
    hipLaunchKernelGGL((cqt_bf16x3_staged_kernel<N, R, C>), grid, block, lds, st, y, L, ldy, hop, T,                 \
                       (const uint4*)gsplit, n_filt, (float2*)out, out_bstride, row0, nchunks, cplane, ncopy, sh);   \
  } while (0)
#define SYG_CQT_ST2(N, R) do { if (maxc <= 2) SYG_CQT_ST(N, R, 2); else if (maxc == 3) SYG_CQT_ST(N, R, 3); else SYG_CQT_ST(N, R, 5); } while (0)
      if (n_fft == 128) { if (rt == 2) SYG_CQT_ST2(128, 2); else SYG_CQT_ST2(128, 1); }
      else { if (rt == 2) SYG_CQT_ST2(256, 2); else SYG_CQT_ST2(256, 1); }
#undef SYG_CQT_ST2
#undef SYG_CQT_ST
      SYG_CHECK_LAUNCH("cqt_octave_bf16x3 (staged)");
      return SYG_OK;
    }
  }
  int64_t waves = (256 * 12) / B;                  // three workgroups of four waves per CU (48 KiB of LDS each)
  if (waves < 4) waves = 4;
  if (waves > ntiles) waves = (ntiles + 3) & ~(int64_t)3;
  waves &= ~(int64_t)3;
  const dim3 grid((unsigned)(waves / 4), (unsigned)B), block(256);
#define SYG_CQT_B3(N, R)                                                                                             \
  hipLaunchKernelGGL((cqt_bf16x3_kernel<N, R>), grid, block, 0, st, y, L, ldy, hop, T, (const uint4*)gsplit, n_filt,  \
                     (float2*)out, out_bstride, row0, (int)waves)
  if (n_fft == 128) { if (rt == 2) SYG_CQT_B3(128, 2); else SYG_CQT_B3(128, 1); }
  else { if (rt == 2) SYG_CQT_B3(256, 2); else SYG_CQT_B3(256, 1); }
#undef SYG_CQT_B3
  SYG_CHECK_LAUNCH("cqt_octave_bf16x3");
  return SYG_OK;
}

#ifdef SYG_CQT_STAMP
extern "C" int syg_debug_cqt_stamps(unsigned long long* host, int n) {
  return hipMemcpyFromSymbol(host, HIP_SYMBOL(syg::cqs_stamp), (size_t)n * sizeof(unsigned long long)) == hipSuccess ? 0 : -1;
}
#endif
