// Fused STFT(2048) -> power -> mel (-> dB -> DCT) kernel for gfx950 (MI355X).
//
// One persistent workgroup of W waves (W = 16: one per CU; W = 8: two) walks a contiguous chunk of tiles; a tile
// is W consecutive frames of one clip, one frame per wave.  Per tile (two barriers):
//   FFT      the tile's contiguous sample run was copied into LDS by LDS-DMA (buffer_load ... lds; the buffer
//            descriptor's range check supplies center=True's zero padding) while the previous tile was being
//            processed; each wave takes its (overlapped) frame from there, applies the analysis window (LDS copy)
//            and runs a 1024-point complex FFT held 16 points per lane (radix 16 x 16 x 4).  The two exchanges go
//            through a 4.1 KiB per-wave scratch that aliases the wave's own power row, in two half-rounds each:
//            full-wave ds_write_b64 / ds_read_b64, conflict-free under the per-instruction LDS banking, the lane
//            pair (L, L + 32) completing each other's rows with v_permlane32_swap.  Real-FFT split on mirror
//            pairs, |X|^2 stored as one skewed 1025-bin row.
//   barrier A
//   project  the W power rows are the B operand of v_mfma_f32_16x16x4_f32: the mel filterbank is stored
//            block-sparse (per 16-mel tile only its non-zero bin range), split over the waves, four k-steps per
//            16-byte load (pre-loaded behind barrier A); then the next frame is fetched LDS -> registers.
//   barrier B
//   reduce   the DMA of the tile after next is started; partial 16 x W tiles are combined in a fixed order
//            (deterministic) into mel[b, m, t] (MODE 0/1) or the clip's LDS mel matrix (MODE 3).
//   MODE 1   per-frame spectral statistics and contrast tail means from the same LDS rows (+ one barrier).
//   MODE 3   at a clip's last tile: power_to_db + DCT-II from the LDS mel matrix inside the next tile's projection
//            phase -- only MFCCs are written.   MODE 2: complex STFT output instead of the projection.
// Nothing but the input samples and the outputs touches HBM.
//
// Reference behaviour reproduced: librosa.stft (center zero padding, periodic window, rfft) -> np.abs ->
// **2 -> melspectrogram -> power_to_db(ref=np.max) -> mfcc, as called from sygnals/core/features/manager.py:184-187,
// 198, 219-223 and cepstral.py:106-115; per-frame statistics follow sygnals/core/features/frequency_domain.py:24-386.
// Index maps and LDS bank behaviour are validated by tools/wave_fft_model_v4.py.
#include "common.h"
#include <stdlib.h>
#include <string.h>

#ifndef SYG_TRIX
#define SYG_TRIX 0   // development experiments on MODE 6 (timing only)
#endif
#ifndef SYG_ABL
#define SYG_ABL 0   // development ablations (tools/ablate.sh); 0 = product build
#endif
// Issue priority falls as a wave advances through its frame: the SIMD's arbiter otherwise favours the oldest
// wave, which then idles at barrier A while the youngest finishes alone with nothing to hide its LDS latency.
#ifndef SYG_NOPRIO
#define SETPRIO(n) __builtin_amdgcn_s_setprio(n)
// level n for the younger waves, one lower for the older ones (experiment SYG_AGEPRIO: the arbiter serves the oldest wave
// first at equal priority)
#define SETPRIO_AGE(n, older) do { if (older) SETPRIO((n) > 0 ? (n) - 1 : 0); else SETPRIO(n); } while (0)
#else
#define SETPRIO(n)
#endif
#if SYG_ABL == 9
// -DSYG_TICK_NOVM: the stamps do not drain vector memory (the DMA / store waits then show up where the product waits)
#ifdef SYG_TICK_NOVM
#define SYG_TICK_WAIT "s_waitcnt lgkmcnt(0)"
#else
#define SYG_TICK_WAIT "s_waitcnt vmcnt(0) lgkmcnt(0)"
#endif
// timeline mode: per-wave cycle accumulators per phase, dumped into stats_out (tools/timeline.py)
#define TICK(slot, reg)                                                                                         \
  do {                                                                                                          \
    unsigned long long _t;                                                                                      \
    asm volatile(SYG_TICK_WAIT "\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(_t), "+v"(reg)::"memory"); \
    tacc[slot] += _t - tprev;                                                                                   \
    tprev = _t;                                                                                                 \
  } while (0)
#define TARGS , unsigned long long (&tacc)[12], unsigned long long& tprev
#define TPASS , tacc, tprev
#else
#define TICK(slot, reg)
#define TARGS
#define TPASS
#endif

namespace syg {
namespace {

#ifndef SYG_NOX2
constexpr bool X2_MEL = true;    // mel-only modes keep 4 |X|^2 in the power rows (see stft2048_kernel)
#else
constexpr bool X2_MEL = false;
#endif
constexpr int NFFT = 2048;
constexpr int MC = 1024;         // complex points per frame
constexpr int NBIN = 1025;
constexpr int MAXW = 16;         // waves per workgroup (8 or 16)
constexpr int P_STRIDE = 1090;   // == 2 (mod 32): conflict-free MFMA B-operand reads; rows are skewed, see ppos()
constexpr int PL2 = 132;         // exchange-2 plane stride (complex): 128 group slots + 4 skew
constexpr int SC_COMPLEX = 4 * PL2;   // per-wave exchange scratch: 528 complex = 4224 B
constexpr int TW2_STRIDE = 18;   // complex entries per lane class (16 + 2 pad: distinct banks)
constexpr int TW2_FLOATS = 4 * TW2_STRIDE * 2;
constexpr int TW1_FLOATS = 15 * 64 * 2;

// Block-sparse filterbank plan for v_mfma_f32_4x4x1_16b_f32 (sygnals_amd/_tables.py: pack_mel_plan): the mel rows are
// taken in groups of four; the non-zero bin range of a group is cut into chunks of at most `steps` bins, one chunk per
// (wave, slot) -- a wave runs four slots side by side, one bin per slot and step.  The tables sit behind the packed
// weights in the same device buffer (int32, 4 x 64 entries from `table_off`, in floats):
//   [0, 64) first row POSITION ppos(bin) of slot (wave * 4 + s)   [64, 128) mel group of the slot (-1: unused)
//   [128, 192) first slot of mel group g          [192, 256) number of slots of group g
struct MelPlan {
  int steps;        // row positions per slot = MFMA steps per wave and tile (a multiple of 4, at least 28)
  int n_groups;     // mel groups of four rows
  int table_off;    // offset (in floats) of the tables inside wpacked
};
constexpr int MTAB_INTS = 256;
constexpr int SEGTAB_WORDS = 2 * 2 * 64 * 4;   // piece table of the segment-sum projection (pack_mel_segments), two passes
constexpr int SEGTAB4_WORDS = 2 * SEGTAB_WORDS; // ... four passes (MODE 8 / 9: filterbanks of up to 256 pieces, e.g. 128 bands)
constexpr int TRI4_ROW_BASE = 4;               // the four-pass tables are built with row_base = 4 (a short first piece needs room
                                               // for its lead): the projection reads from 4 words in front of the power row

struct ContrastPlan {
  int n_rows;
  int ascending;     // 1: lo[] and hi[] are non-decreasing (a band never starts below an earlier band's start)
  int lo[SYG_MAX_BANDS];
  int hi[SYG_MAX_BANDS];
  int k[SYG_MAX_BANDS];
};

typedef float v4f __attribute__((ext_vector_type(4)));

// MODE 3 (clip-resident MFCC): the mel tiles of a clip stay in LDS; after the clip's last tile the workgroup
// converts them to dB (per-clip max reference, top_db floor) and applies the DCT -- only MFCCs reach HBM.
struct MfccArgs {
  const float* dct;      // [n_mfcc, n_mels]
  const float* lifter;   // [n_mfcc] or null
  float* out;            // [B, n_mfcc, T]
  int n_mfcc;
  int ref_is_max;        // 1: reference = max of the clip's mel powers, 0: ref_value
  float ref_value, amin, top_db;   // top_db < 0: no floor
  int tp;                // padded frames per clip (tiles_per_clip * TILE_T): row stride of the LDS mel matrix
  int rows_per_clip;     // rows between two clips of `out` (n_mfcc, or more when the MFCCs are the head of a wider block)
};

// exchange 2 (half buffer by c' & 7, planar in b'): slot of group (c, c') inside a plane
__device__ __forceinline__ int x2g(int c, int cp) { return (cp & 7) * 16 + ((c + 4 * ((cp & 7) >> 1)) & 15); }
// position of bin k inside an LDS power row: one pad word every 16 bins turns the stride-16 bin pattern
// of the pass-3 output into a conflict-free store while 16-aligned runs of bins stay contiguous for the MFMA
__device__ __forceinline__ int ppos(int k) { return k + (k >> 4); }

struct LaneConst {
  float2 twb[2];     // W_2048^kb of each unit (kb = c + 16 c'); pair d uses twb * W_8^d
  float2 cA2, cA3;   // unit-0 multipliers for d = 2, 3: W_8^2, W_8^3 -- except lane 0 (see below)
  int pkb[2];        // ppos(kb): pair d of a regular unit sits at pkb + 272 d (its mirror at pmb - 272 d)
  int pmb[2];        // ppos(1024 - kb)
  int dA2, dA3;      // unit-0 position offsets for d = 2, 3 (544, 816 -- except lane 0)
  int kb[2];         // kb (complex-output mode)
  int g0[2], g1[2];  // exchange-2 slot of the primary (c' < 8) / mirror (c' >= 8) group
};

// unit u = lane + 64 j: primary group (c = u >> 3, c' = u & 7), bins k = kb + 256 d; mirror group
// (16 - c, 15 - c') holds bins 1024 - k, with the c = 0 exceptions (0, 16 - c') and, for u = 0 (lane 0), the
// self-mirrored pair of groups (0,0) / (0,8) whose four pairs are the bins {0, 256, 128, 384} (+ bin 512).
__device__ __forceinline__ void init_lane_const(LaneConst& lc, int lane, const float2* __restrict__ twid) {
  constexpr float C1 = 0.92387953251128673848f, S1 = 0.38268343236508977173f, R = 0.70710678118654752440f;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int u = lane + 64 * j, c = u >> 3, cp = u & 7;
    int cm = 16 - c, cmp = 15 - cp;
    if (c == 0) { cm = 0; cmp = (cp == 0) ? 8 : 16 - cp; }
    lc.g0[j] = x2g(c, cp);
    lc.g1[j] = x2g(cm, cmp);
    const int kb = c + 16 * cp;
    lc.kb[j] = kb;
    lc.pkb[j] = ppos(kb);
    lc.pmb[j] = ppos(MC - kb);
    lc.twb[j] = twid[kb];
  }
  const bool sp = (lane == 0);
  lc.cA2 = sp ? make_float2(C1, -S1) : make_float2(0.f, -1.f);   // W_2048^128 = W_16^1   |  W_8^2
  lc.cA3 = sp ? make_float2(S1, -C1) : make_float2(-R, -R);      // W_2048^384 = W_16^3   |  W_8^3
  lc.dA2 = sp ? ppos(128) : 544;
  lc.dA3 = sp ? ppos(384) : 816;
}

// 1024-point complex forward FFT of the windowed frame + real split.  v[a] holds z[64a + lane] on entry.
// On exit pair (j, d) holds X[k] in xs[j][d] and X[1024 - k] in xm[j][d] (k = kb_j + 256 d; lane 0 / unit 0:
// k = 0, 256, 128, 384); lane 0 also returns X[512].
// NPF > 0: the first NPF 16-byte groups of the wave's filterbank operands (pf_src, 64 float4 apart) are requested
// behind pass 3 -- the transform's register peak is over there -- so that their L2 latency hides behind the real split
// and the row stores instead of standing in front of barrier A.
// PD: priority drop (MODE 1 keeps the top level for its row functions, the transform then runs one level lower)
// X2: return 2 X instead of X (the two halvings of the real split are left out; the caller's powers are then 4 |X|^2 --
// an exact scaling that the mel-only modes take back where the mel values leave the kernel: 32 instructions per frame)
template <int NPF, int PD = 0, bool X2 = false>
__device__ __forceinline__ void wave_rfft2048(float2 (&v)[16], const LaneConst& lc, float2* __restrict__ sc,
                                             const float2* __restrict__ tw1l, const float2* __restrict__ tw2l,
                                             int lane, float2 (&xs)[2][4], float2 (&xm)[2][4], float2& x512,
                                             const float4* __restrict__ pf_src, float4* __restrict__ pf, bool older TARGS) {
  const int cl = lane >> 2, bp = lane & 3;
  // ---- pass 1: radix-16 over a (stride 64), twiddle W_1024^(b*c)
  dft16(v);
#if SYG_ABL == 3 || SYG_ABL == 4
#pragma unroll
  for (int c = 1; c < 16; ++c) v[c] = cmul(v[c], make_float2(0.6f, 0.8f));
#else
#pragma unroll
  for (int c = 1; c < 16; ++c) v[c] = cmul(v[c], tw1l[(c - 1) * 64 + lane]);
#endif
  // ---- exchange 1 in two half-rounds through a 512-complex buffer.  Round h moves the 8 ROWS c = 8h..8h+7:
  // all 64 lanes store (row r = c & 7 at r*64 + (b ^ 4r)) -- an LDS store costs its 6 cycles whatever the EXEC
  // mask, so half-wave stores would pay twice.  A pass-2 lane (c = lane>>2, b' = lane&3) needs the 16 operands
  // y[c][4a + b'] of ONE row, which only one of the rounds holds: in round h the lane pair (L, L + 32) shares
  // the reading of row (L>>2) + 8h -- L takes a = 0..7, L + 32 takes a = 8..15 -- and v_permlane32_swap then
  // hands each lane the half it is missing (L's round-1 operands <-> (L+32)'s round-0 operands): full-wave
  // stores AND full-wave loads for 16 extra VALU instructions.
  TICK(1, v[1].x);
  float2 t[16];
  {
#if SYG_ABL == 2 || SYG_ABL == 4
#pragma unroll
    for (int a = 0; a < 16; ++a) t[a] = v[a];
#else
    const int r7 = cl & 7;
    const int rbase = r7 * 64 + bp + 8 * (cl & 8) / 2;      // + 32 for the upper half-wave (a = 8..15)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
      for (int r = 0; r < 8; ++r) sc[r * 64 + (lane ^ (4 * r))] = v[8 * h + r];
      wave_lds_sync();
#pragma unroll
      for (int i = 0; i < 8; ++i) t[8 * h + i] = sc[rbase + 4 * (i ^ r7)];
      wave_lds_sync();
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const auto sx = __builtin_amdgcn_permlane32_swap(__float_as_uint(t[i].x), __float_as_uint(t[8 + i].x), false, false);
      const auto sy = __builtin_amdgcn_permlane32_swap(__float_as_uint(t[i].y), __float_as_uint(t[8 + i].y), false, false);
      t[i] = make_float2(__uint_as_float(sx[0]), __uint_as_float(sy[0]));
      t[8 + i] = make_float2(__uint_as_float(sx[1]), __uint_as_float(sy[1]));
    }
#endif
  }
  TICK(2, t[0].x);
  // ---- pass 2: lane = (c = lane>>2, b' = lane&3); radix-16 over a'
  SETPRIO_AGE(2 - PD > 0 ? 2 - PD : 0, older);
  dft16(t);
  {
    // W_64^(b'*c') from a 64-entry LDS table (4 lane classes): two twiddles per 16-byte read
    // (read through the LDS address space as one native 4-vector: a generic float4 is split into two 8-byte
    // halves and re-fused into ds_read2_b64, which costs twice the LDS cycles of ds_read_b128)
    typedef __attribute__((address_space(3))) const v4f* lds_v4;
    lds_v4 t4 = (lds_v4)(tw2l + bp * TW2_STRIDE);
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      const v4f tt = t4[m];
      if (m > 0) t[2 * m] = cmul(t[2 * m], make_float2(tt.x, tt.y));
      t[2 * m + 1] = cmul(t[2 * m + 1], make_float2(tt.z, tt.w));
    }
  }
  // ---- exchange 2 in two half-rounds (c' < 8, then c' >= 8); every unit's primary group has c' < 8 and its
  // mirror c' >= 8, so round 0 delivers all primaries and round 1 all mirrors.  Pass 3 = radix-4 over b'.
  TICK(3, t[1].x);
  float2 G[2][4], H[2][4];
  {
    const int wbase = bp * PL2;
#if SYG_ABL == 2 || SYG_ABL == 4
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      bfly4(t[4 * j], t[4 * j + 1], t[4 * j + 2], t[4 * j + 3], G[j][0], G[j][1], G[j][2], G[j][3]);
      bfly4(t[8 + 4 * j], t[9 + 4 * j], t[10 + 4 * j], t[11 + 4 * j], H[j][0], H[j][1], H[j][2], H[j][3]);
    }
#else
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
      for (int r = 0; r < 8; ++r) sc[wbase + r * 16 + ((cl + 4 * (r >> 1)) & 15)] = t[8 * h + r];
      wave_lds_sync();
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const float2* p = sc + (h == 0 ? lc.g0[j] : lc.g1[j]);
        if (h == 0) bfly4(p[0], p[PL2], p[2 * PL2], p[3 * PL2], G[j][0], G[j][1], G[j][2], G[j][3]);
        else bfly4(p[0], p[PL2], p[2 * PL2], p[3 * PL2], H[j][0], H[j][1], H[j][2], H[j][3]);
      }
      wave_lds_sync();
    }
#endif
  }
  TICK(4, G[0][0].x);
  SETPRIO_AGE(1 - PD > 0 ? 1 - PD : 0, older);
  if (NPF > 0) {
    int lp = lane;                     // laundered: the loads may not be hoisted above this point (register peak)
    asm volatile("" : "+v"(lp));
#pragma unroll
    for (int q = 0; q < NPF; ++q) pf[q] = pf_src[(int64_t)q * 64 + lp];
  }
  // ---- real split on mirror pairs
  x512 = X2 ? make_float2(2.f * G[0][2].x, -2.f * G[0][2].y)
            : make_float2(G[0][2].x, -G[0][2].y);      // X[512] = conj(Z[512]) (meaningful in lane 0 only)
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    float2 zk[4] = {G[j][0], G[j][1], G[j][2], G[j][3]};
    float2 zm[4] = {H[j][3], H[j][2], H[j][1], H[j][0]};
    if (j == 0) {
      // unit 0 (lane 0) pairs the self-mirrored groups (0,0) and (0,8) differently
      const bool sp = (lane == 0);
      zk[2] = sp ? H[0][0] : zk[2];
      zk[3] = sp ? H[0][1] : zk[3];
      zm[0] = sp ? G[0][0] : zm[0];
      zm[1] = sp ? G[0][3] : zm[1];
      zm[2] = sp ? H[0][3] : zm[2];
      zm[3] = sp ? H[0][2] : zm[3];
    }
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      // E2 = zk + conj(zm), O2 = -i (zk - conj(zm));  X[k] = (E2 + w O2)/2, X[1024-k] = conj(E2 - w O2)/2
      const float2 E = make_float2(zk[d].x + zm[d].x, zk[d].y - zm[d].y);
      const float2 O = make_float2(zk[d].y + zm[d].y, zm[d].x - zk[d].x);
      // w = twb * W_8^d (unit 0, d >= 2: twb * cA_d, which differs in lane 0 only)
      constexpr float R = 0.70710678118654752440f;
      float2 rO;
      if (d == 0) rO = O;
      else if (d == 1) rO = make_float2(R * (O.x + O.y), R * (O.y - O.x));
      else if (j == 0) rO = cmul(O, d == 2 ? lc.cA2 : lc.cA3);
      else if (d == 2) rO = make_float2(O.y, -O.x);
      else rO = make_float2(R * (O.y - O.x), -R * (O.x + O.y));
      const float2 wO = cmul(lc.twb[j], rO);
      if (X2) {
        xs[j][d] = make_float2(E.x + wO.x, E.y + wO.y);
        xm[j][d] = make_float2(E.x - wO.x, wO.y - E.y);
      } else {
        xs[j][d] = make_float2(0.5f * (E.x + wO.x), 0.5f * (E.y + wO.y));
        xm[j][d] = make_float2(0.5f * (E.x - wO.x), -0.5f * (E.y - wO.y));
      }
    }
  }
}

// Raw samples of one frame: element n = 64a + lane of the packed complex frame covers samples s0 + 2n,
// s0 + 2n + 1; samples outside [0, L) are the zero padding of center=True.  The loads are only issued here;
// the analysis window (LDS copy) is applied by apply_window() when the frame is consumed.
//   LOAD 0 / 1: straight from global memory (scalar / 8-byte loads)
//   LOAD 2    : from the tile's staged sample run in LDS (filled by LDS-DMA, stage_tile())
template <int LOAD>
__device__ __forceinline__ void fetch_frame(float2 (&v)[16], const float* __restrict__ yb, int64_t L, int64_t s0,
                                            const float* __restrict__ stage_frame, int lane) {
  if (LOAD == 2) {
    const float2* sf = reinterpret_cast<const float2*>(stage_frame);
#pragma unroll
    for (int a = 0; a < 16; ++a) v[a] = sf[64 * a + lane];
    return;
  }
  const bool interior = (s0 >= 0) && (s0 + NFFT <= L);
  if (interior) {
#pragma unroll
    for (int a = 0; a < 16; ++a) {
      const int n = 64 * a + lane;
      if (LOAD == 1) v[a] = *reinterpret_cast<const float2*>(yb + s0 + 2 * n);
      else v[a] = make_float2(yb[s0 + 2 * n], yb[s0 + 2 * n + 1]);
    }
  } else {
#pragma unroll
    for (int a = 0; a < 16; ++a) {
      const int64_t s = s0 + 2 * (64 * a + lane);
      v[a].x = (s >= 0 && s < L) ? yb[s] : 0.f;
      v[a].y = (s + 1 >= 0 && s + 1 < L) ? yb[s + 1] : 0.f;
    }
  }
}

__device__ __forceinline__ void apply_window(float2 (&v)[16], const float2* __restrict__ winl, int lane) {
#pragma unroll
  for (int a = 0; a < 16; ++a) {
    const float2 wv = winl[64 * a + lane];
    v[a] = make_float2(v[a].x * wv.x, v[a].y * wv.y);
  }
}

// LDS-DMA of the sample run [s_begin, s_begin + 256 n_chunks) of one clip into the stage buffer: the buffer
// descriptor's range check returns zeros for samples before the clip (negative offsets wrap to huge unsigned
// ones) and past its end -- exactly the zero padding of center=True.  wide: 16 B per lane (needs 16-byte
// aligned runs), else 4 B per lane.
template <int WAVES>
__device__ __forceinline__ void stage_tile(const float* yb, int clip_bytes, float* stage, int s_begin, int span,
                                           bool wide, int w, int lane) {
  typedef __attribute__((address_space(3))) void* lds_ptr;
  const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(yb), 0, clip_bytes, 0x00020000);
  if (wide) {
    const int n = (span + 255) >> 8;
    for (int c = w; c < n; c += WAVES)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr)(stage + c * 256), 16, (s_begin + c * 256 + lane * 4) * 4,
                                               0, 0, 0);
  } else {
    const int n = (span + 63) >> 6;
    for (int c = w; c < n; c += WAVES)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr)(stage + c * 64), 4, (s_begin + c * 64 + lane) * 4, 0, 0,
                                               0);
  }
}

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
#ifndef SYG_R7ABL
#define SYG_R7ABL 0
#endif
#ifndef SYG_SELBITS
#define SYG_SELBITS 16
#endif
#ifndef SYG_SEL2
#define SYG_SEL2 1
#endif
#ifndef SYG_P6SPLIT
#define SYG_P6SPLIT 0      // (experiment, correct, measured without gain: 140.8 vs 140.6 us -- see proj_split below)
#endif
#ifndef SYG_R7PRIO
#define SYG_R7PRIO 3      // issue priority of the row functions of MODE 7 (the transform runs one level below the top)
#endif
#ifndef SYG_R7SHIFT
#define SYG_R7SHIFT 2     // waves w, w + 4, w + 8, w + 12 share a SIMD: two of them early, two late
#endif
#ifndef SYG_R7SPLIT
#define SYG_R7SPLIT 1
#endif
#ifndef SYG_ROWBOTH
#define SYG_ROWBOTH 1
#endif
typedef __attribute__((address_space(1))) float* gptr;

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
__device__ __forceinline__ float row_stats_body(lds_row prow, int lane, float binhz, float roll_percent, float bw_p, int smask) {
  float res = 0.f;
#define SYG_PUT(row, val) res = (lane == (row)) ? (val) : res
  // lane owns the 16 contiguous bins [16 lane, 16 lane + 16) -- 16 consecutive words at 17 lane of the skewed row:
  // immediate offsets, no bank conflicts -- and lane 63 also the Nyquist bin 1024 as a 17th value (0 in the other
  // lanes).  The powers are read ONCE and every statistic works on the registers (round 2 re-read the row per pass to
  // stay inside the caller-saved registers; 17 + 17 values still do).
  const float EPS = 2.220446049250313e-16f;
  const bool last = (lane == 63);
  float p[17];
  {
    lds_row pr = prow + 17 * lane;
#pragma unroll
    for (int i = 0; i < 16; ++i) p[i] = pr[i];
    const float nyq = pr[17];                      // ppos(1024) = 17 * 63 + 17 for lane 63 (inside the row's slack elsewhere)
    p[16] = last ? nyq : 0.f;
  }
  float psum = 0.f;
#pragma unroll
  for (int i = 0; i < 17; ++i) psum += p[i];
  const float tot_p = wave_sum(psum);
  const float kb = (float)(16 * lane);
  float tot_m = 0.f, cen_bin = 0.f;
  bool live = false;
  if (smask & (1 | 2 | 4)) {        // magnitude sums
    float m[17];
    float msum = 0.f, fl = 0.f;
#pragma unroll
    for (int i = 0; i < 17; ++i) {
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
      for (int i = 0; i < 16; ++i) lsum += __builtin_amdgcn_logf(m[i] + EPS);
      const float l16 = __builtin_amdgcn_logf(m[16] + EPS);
      lsum += last ? l16 : 0.f;
      const float tot_l = wave_sum(lsum) * 0.69314718055994531f;
      const float am = tot_m * (1.f / (float)NBIN);
      SYG_PUT(SYG_STAT_FLATNESS, (am >= EPS) ? fminf(fmaxf(fexp(tot_l * (1.f / (float)NBIN)) * frcp(am), 0.f), 1.f) : 0.f);
    }
    if (smask & 2) {    // bandwidth: (sum m |f - c|^p / sum m)^(1/p);  (m[16] = 0 outside lane 63)
      const int pmode = (bw_p == 2.f) ? 2 : (bw_p == 1.f) ? 1 : 0;
      const float d0 = kb - cen_bin;
      float dsum = 0.f;
      if (pmode == 2) {
#pragma unroll
        for (int i = 0; i < 17; ++i) { const float d = (d0 + (float)i) * binhz; dsum = fmaf(m[i], d * d, dsum); }
      } else {
#pragma unroll
        for (int i = 0; i < 17; ++i) {
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
    for (int i = 1; i < 17; ++i) {
      const bool up = (i < 16 || last) && p[i] > pmax;
      pmax = up ? p[i] : pmax; amax = up ? i : amax;
    }
    const float gm = wave_max(pmax);
    const int cand = wave_min_i((pmax == gm) ? 16 * lane + amax : 0x7fffffff);
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
      for (int i = 0; i < 16; ++i) {
        c += p[i];
        below += (c < thr) ? 1 : 0;
      }
      c += p[16];
      below += (last && c < thr) ? 1 : 0;
    } else {
      float mg = (lane > 0) ? fabsf(c - thr) : 3.4e38f;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        c += p[i];
        below += (c < thr) ? 1 : 0;
        mg = fminf(mg, fabsf(c - thr));
      }
      c += p[16];
      below += (last && c < thr) ? 1 : 0;
      mg = fminf(mg, fabsf(c - thr));
      mgw = wave_min(mg);
    }
    const int rb = (below < (last ? 17 : 16)) ? 16 * lane + below : 0x7fffffff;
    int rbmin = wave_min_i(rb);
    if (rbmin == 0x7fffffff || tot_p < EPS) rbmin = NBIN - 1;
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
#if SYG_SEL2
  wave_kth_largest2x2_u32<SYG_SELBITS>(__float_as_uint(t1), __float_as_uint(t2), ~__float_as_uint(b1), ~__float_as_uint(b2), k, Tu, Bu);
#else
  wave_kth_largest2_u32<16>(__float_as_uint(t1), ~__float_as_uint(b1), k, Tu, Bu);
#endif
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

// mean of the k smallest and k largest MAGNITUDES of bins [lo, hi) of one LDS power row (identical to sorting,
// as librosa does: values are non-negative, selection on power == selection on magnitude).
//   bands of <= 768 bins with k <= 16 : register extraction, specialised by registers per lane;
//   otherwise                         : radix select of the k-th order statistic + tail sum closed with the
//                                       tie count.
// may_park: the bands come in ascending order and the row's statistics are done, so a wide band may park its sorted
// lists in the part of the row below its own end (contrast_extract_lds)
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
__device__ __forceinline__ MfccArgs uni(const MfccArgs& a) {
  MfccArgs u;
  u.dct = uni(a.dct); u.lifter = uni(a.lifter); u.out = uni(a.out); u.n_mfcc = uni(a.n_mfcc);
  u.ref_is_max = uni(a.ref_is_max); u.ref_value = uni(a.ref_value); u.amin = uni(a.amin); u.top_db = uni(a.top_db);
  u.tp = uni(a.tp); u.rows_per_clip = uni(a.rows_per_clip);
  return u;
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
#ifdef SYG_NO_NARROW_GROUP
  return 0;                                                    // (timing variant: every band through the loop)
#endif
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
__device__ __forceinline__ float2 row_contrast_body(lds_row prow, int lane, lds_iptr cpl, int n_rows_v, int may_park_v) {
  const int n_rows = uni(n_rows_v), may_park = uni(may_park_v);
  const int lb = lane & (SYG_MAX_BANDS - 1);
  const int plo = cpl[lb], phi = cpl[SYG_MAX_BANDS + lb], pk = cpl[2 * SYG_MAX_BANDS + lb];
  float rp = 0.f, rv = 0.f;
  const int r0 = contrast_narrow_group(prow, lane, plo, phi, pk, n_rows, rp, rv);
  for (int r = r0; r < n_rows; ++r) {
    const int lo = __builtin_amdgcn_readlane(plo, r), hi = __builtin_amdgcn_readlane(phi, r),
              k = __builtin_amdgcn_readlane(pk, r);
    const float2 pv = band_contrast(prow, lane, lo, hi, k, may_park);
    rp = (lane == r) ? pv.x : rp;
    rv = (lane == r) ? pv.y : rv;
  }
  return make_float2(rp, rv);
}
__device__ __noinline__ float row_trivial(lds_row prow, int lane) { return wave_sum(prow[17 * lane]); }
__device__ __noinline__ float row_stats(lds_row prow, int lane, float binhz, float roll_percent, float bw_p, int smask) {
  return row_stats_body(prow, lane, binhz, roll_percent, bw_p, smask);
}
__device__ __noinline__ float2 row_contrast_all(lds_row prow, int lane, lds_iptr cpl, int n_rows_v, int may_park_v) {
  return row_contrast_body(prow, lane, cpl, n_rows_v, may_park_v);
}
// Statistics AND contrast of one row in one call (the C4 block asks for both: one entry / exit sequence, one wait for
// the outstanding memory operations, instead of two).  x: the statistics register of row_stats, y / z: peak / valley.
__device__ __noinline__ float3 row_features(lds_row prow, int lane, float binhz, float roll_percent, float bw_p, int smask,
                                            lds_iptr cpl, int n_rows_v, int may_park_v) {
  const float s = row_stats_body(prow, lane, binhz, roll_percent, bw_p, smask);
  const float2 pv = row_contrast_body(prow, lane, cpl, n_rows_v, may_park_v);
  return make_float3(s, pv.x, pv.y);
}

// MODE 3 clip epilogue (a workgroup's chunk is whole clips): power_to_db + DCT-II (+ lifter) from the LDS mel
// matrix -- librosa.power_to_db(S, ref=np.max) (manager.py:223) -> scipy.fft.dct rows (cepstral.py:106-115).
// LDS behind the mel matrix: red[WAVES] (per-wave maxima of the clip's mel powers, written at the clip's last
// reduce), then the DCT rows [n_mfcc][n_mels] and the lifter [n_mfcc] (copied once at kernel start).
// The waves that own a 16 x 16 output tile run this in the NEXT tile's projection phase: barrier A of that tile
// orders it behind the last reduce of the clip, barrier B ahead of the next reduce that overwrites the matrix.
// The powers are converted to dB on the fly with the hardware log2: dB = 10 log10(2) (log2 x - log2 ref)
// (v_log_f32, 1 ulp: ~6e-6 dB at -100 dB; inputs are >= amin, never denormal; exactly 0 when x == ref).
// Out of line (inlined, its scalars push the tile loop's SGPRs into spills); uni() re-scalarises the arguments.
typedef __attribute__((address_space(3))) float* lds_fptr;     // (a generic pointer would make every access a flat_*)
// ----------------------------------------------------------------------------------
// MODE 6 / 7: the mel projection of ONE power row by ONE wave, by segment sums: tri_project<2>() of mel_segments.h
// ----------------------------------------------------------------------------------
#include "mel_segments.h"

template <int WAVES>
__device__ __noinline__ void clip_dct(int clipmel_addr, int red_addr, int dct_addr, MfccArgs mfv, int n_mels_v, int T_v, int b_v,
                                      int w_v, int lane, int stride_v = WAVES) {
  const MfccArgs mf = uni(mfv);
  const int n_mels = uni(n_mels_v), T = uni(T_v), w = uni(w_v), stride = uni(stride_v);   // output tiles w, w + stride, ...
  const int64_t b = uni(b_v);
  lds_fptr clipmel = (lds_fptr)(uintptr_t)(uint32_t)uni(clipmel_addr);
  lds_fptr red = (lds_fptr)(uintptr_t)(uint32_t)uni(red_addr);          // [WAVES] per-wave maxima of the clip
  lds_fptr dctl = (lds_fptr)(uintptr_t)(uint32_t)uni(dct_addr);         // [n_mfcc][n_mels] DCT rows, then the lifter
  lds_fptr lifl = dctl + mf.n_mfcc * n_mels;
  // clip maximum: one LDS read per lane (the WAVES per-wave maxima, one per lane of a DPP row) + row reduction
  float m = red[lane & (WAVES - 1)];
  m = fmaxf(m, dpp_f<DPP_QP_1032>(m));
  m = fmaxf(m, dpp_f<DPP_QP_2301>(m));
  m = fmaxf(m, dpp_f<DPP_ROW_HALF_MIRROR>(m));
  m = fmaxf(m, dpp_f<DPP_ROW_MIRROR>(m));
  constexpr float DB_PER_LOG2 = 3.01029995663981195f;
  const float ref = mf.ref_is_max ? m : fabsf(mf.ref_value);
  const float reflog = __builtin_amdgcn_logf(fmaxf(mf.amin, ref));
  // log_spec.max() - top_db, with log_spec monotone in the power
  const float flo = (mf.top_db >= 0.f) ? DB_PER_LOG2 * (__builtin_amdgcn_logf(fmaxf(mf.amin, m)) - reflog) - mf.top_db
                                       : -3.4e38f;
  // out[k, t] = sum_m dct[k, m] * dB[m, t]
  const int ktiles = (mf.n_mfcc + 15) >> 4, ttiles = mf.tp >> 4;
  const int f = lane & 15, g = lane >> 4;
  for (int ot = w; ot < ktiles * ttiles; ot += stride) {
    const int kt = ot / ttiles, tq = ot - kt * ttiles;
    const int krow = kt * 16 + f, tcol = tq * 16 + f;
    v4f acc = {0.f, 0.f, 0.f, 0.f};
    // four steps per round: the operand reads of a round are in flight together (this wave is on the
    // critical path to barrier B)
    for (int m0 = 0; m0 < n_mels; m0 += 16) {
      float a[4], bv[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int mm = m0 + 4 * i + g;
        const bool ok = mm < n_mels;
        a[i] = (krow < mf.n_mfcc && ok) ? dctl[krow * n_mels + mm] : 0.f;
        bv[i] = ok ? clipmel[mm * mf.tp + tcol] : 1.f;
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float db = fmaxf(DB_PER_LOG2 * (__builtin_amdgcn_logf(fmaxf(mf.amin, bv[i])) - reflog), flo);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], db, acc, 0, 0, 0);     // (rows past n_mels: a = 0)
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int k = kt * 16 + 4 * g + r;
      if (k < mf.n_mfcc && tcol < T) {
        float val = acc[r];
        if (mf.lifter) val *= lifl[k];
        mf.out[(b * mf.rows_per_clip + k) * (int64_t)T + tcol] = val;
      }
    }
  }
}

// ----------------------------------------------------------------------------------
// MODE 0: mel only   MODE 1: mel + per-frame statistics / contrast   MODE 2: complex STFT output
// LOAD 0 / 1: frames read straight from global memory   LOAD 2: tiles staged in LDS by LDS-DMA (hop <= 512)
//
// LDS map (floats):  Pbuf [TILE_T][P_STRIDE] + 16   power rows; a wave's exchange scratch aliases ITS OWN row
//                                                   (the row is only written after the last scratch read, and
//                                                   nobody reads rows while an FFT phase is running)
//                    slab [WAVES][16][TILE_T]       per-wave partial mel tiles
//                    tw2l, tw1l                     twiddle tables        cpl: contrast plan
//                    winl [2048]                    analysis window
//                    stage [(WAVES-1)*512 + 2048]   the tile's contiguous sample run (LOAD 2)
// Per tile: FFT(v) -> rows | barrier A | MFMA -> slab ; fetch the next frame into v | barrier B |
//           start the DMA of the tile after next ; reduce + store [; statistics | barrier].
// The DMA therefore runs behind the reduce and the whole next FFT phase, and is drained at barrier A.
template <int WAVES, bool SLAB = true, int NPASS = 2>
struct Lds {
  static constexpr int TILE_T = WAVES;
  // (four-pass tables: 16 words in front of the first row, so that row 0's lead words are inside the allocation)
  static constexpr int O_P = (NPASS == 4) ? 16 : 0;
  static constexpr int P_FLOATS = O_P + TILE_T * P_STRIDE + 16;
  static constexpr int SLAB_FLOATS = SLAB ? WAVES * 16 * TILE_T : 0;    // (MODE 6 projects per wave: no partial tiles)
  // contrast plan + the mel plan's slot / group tables; MODE 6: the piece table of the segment-sum projection instead
  static constexpr int SEG_WORDS = NPASS * (SEGTAB_WORDS / 2);
  static constexpr int CPL_FLOATS = SLAB ? 3 * SYG_MAX_BANDS + MTAB_INTS : SEG_WORDS + 3 * SYG_MAX_BANDS;
  static constexpr int STAGE_FLOATS = (WAVES - 1) * 512 + NFFT;
  static constexpr int O_SLAB = P_FLOATS;
  static constexpr int O_TW2 = O_SLAB + SLAB_FLOATS;
  static constexpr int O_TW1 = O_TW2 + TW2_FLOATS;
  static constexpr int O_CPL = O_TW1 + TW1_FLOATS;
  static constexpr int O_WIN = O_CPL + CPL_FLOATS;
  static constexpr int O_STAGE = O_WIN + NFFT;
  static constexpr int TOTAL = O_STAGE + STAGE_FLOATS;
  static_assert(SC_COMPLEX * 2 <= P_STRIDE, "exchange scratch must fit inside a power row");
  static_assert(O_TW2 % 4 == 0 && O_WIN % 4 == 0 && O_STAGE % 4 == 0 && O_CPL % 4 == 0, "16-byte aligned LDS sections");
};

template <int WAVES, int LOAD, int MODE>
__global__ __launch_bounds__(WAVES * 64, 4) void stft2048_kernel(
    const float* __restrict__ y, int64_t L, int64_t ldy, int hop, int pad, int64_t T, int tiles_per_clip,
    int64_t total_tiles, int tiles_per_wg, const float2* __restrict__ win2, const float2* __restrict__ twid,
    const float* __restrict__ wpacked, MelPlan plan, int n_mels, float* __restrict__ mel_out, float binhz,
    float roll_percent, float bw_p, int smask, float* __restrict__ stats_out, ContrastPlan cplan,
    float* __restrict__ contrast_out, float2* __restrict__ cout, int dma_wide, MfccArgs mf) {
  // MODE 6: MODE 3 with the per-wave projection by segment sums (tri_project); MODE 7: MODE 6 + the per-frame row
  // functions of MODE 1 (statistics / contrast): config C4's four features from one launch, no mel matrix in HBM
  // MODE 8: the tile form of MODE 6 -- the per-wave projection with a FOUR-pass table (up to 256 pieces: the reference's
  // default 128 bands, 64 bands at 44.1 / 48 kHz), every frame's mel column written straight to HBM (a 128-band clip
  // matrix does not fit the LDS beside the rows; syg_logmel_dct_f32 is the second launch), tiles shared out evenly over
  // the workgroups (no whole-clip chunks: one long clip fills the chip); MODE 9: MODE 8 + the row functions of MODE 7
  constexpr bool TRIMEL = (MODE == 8 || MODE == 9);
  constexpr bool TRI = (MODE == 6 || MODE == 7 || TRIMEL);
  constexpr int NPASS = TRIMEL ? 4 : 2;
  typedef Lds<WAVES, !TRI, NPASS> LM;
  constexpr int SEG_WORDS = LM::SEG_WORDS;
  constexpr int NTHREADS = WAVES * 64;
  constexpr int TILE_T = WAVES;                                    // one frame per wave per tile
  constexpr bool COMPLEX_OUT = (MODE == 2);
  // MODE 5 = MODE 1 (statistics / contrast rows) + MODE 3 (clip-resident dB + DCT): config C4's four features from ONE
  // launch -- only samples in, MFCCs + statistics rows + contrast tail means out (the mel matrix never reaches HBM)
  constexpr bool ROWFN = (MODE == 1 || MODE == 5 || MODE == 7 || MODE == 9);   // per-frame row functions (MODE 1 / 5: behind barrier B)
  constexpr bool CLIPM = (MODE == 3 || MODE == 5 || MODE == 6 || MODE == 7);   // the clip's mel matrix lives in LDS; epilogue at clip end
  // MODE 0 / 3 (mel only): the power rows hold 4 |X|^2 (wave_rfft2048<.., X2>); the factor is taken back -- exactly, a
  // power of two -- where mel values leave the kernel (MODE 0: at the store; MODE 3: the dB conversion works on 4 x mel
  // with 4 x amin and 4 x ref, the optional mel copy is scaled at its store).  MODE 1's statistics need the true powers.
  constexpr bool X2 = X2_MEL && (MODE == 0 || MODE == 3 || MODE == 6 || MODE == 8);
  constexpr float MELSC = X2 ? 0.25f : 1.f;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* Pbuf = lds + LM::O_P;
  float* slab = lds + LM::O_SLAB;
  float2* tw2l = reinterpret_cast<float2*>(lds + LM::O_TW2);        // [4][18] complex
  float2* tw1l = reinterpret_cast<float2*>(lds + LM::O_TW1);        // [15][64] complex
  int* cpl = reinterpret_cast<int*>(lds + LM::O_CPL);
  int* cplc = TRI ? cpl + SEG_WORDS : cpl;           // contrast plan (MODE 6 ... 9: behind the piece table)
  float2* winl = reinterpret_cast<float2*>(lds + LM::O_WIN);
  float* stage = lds + LM::O_STAGE;
  float* clipmel = lds + LM::TOTAL;                 // MODE 3: [n_mels][mf.tp], red[WAVES], dct rows, lifter
  // MODE 6: two mel matrices (the DCT of a clip runs beside the first tile of the next), red[2][WAVES], dct rows, lifter
  float* tri_red = clipmel + 2 * n_mels * mf.tp;
  float* tri_dct = tri_red + 2 * WAVES;

  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform: keeps frame addressing on the scalar unit

  // persistent workgroup: a contiguous chunk of tiles (consecutive tiles of a clip share part of their
  // samples, so the re-reads of the frame overlap stay in this XCD's L2)
  const int64_t tile_begin = (int64_t)blockIdx.x * tiles_per_wg;
  const int64_t tile_end = (tile_begin + tiles_per_wg < total_tiles) ? tile_begin + tiles_per_wg : total_tiles;
  if (tile_begin >= tile_end) return;

  // tile -> (clip, first frame); total_tiles < 2^31 (checked on the host)
  auto clip_of = [&](int64_t tile) { return (uint32_t)tile / (uint32_t)tiles_per_clip; };
  auto t0_of = [&](int64_t tile, uint32_t cq) {
    return (int64_t)((uint32_t)tile - cq * (uint32_t)tiles_per_clip) * TILE_T;
  };
  const int span = (TILE_T - 1) * hop + NFFT;              // samples a tile touches
  auto dma = [&](int64_t tile) {
    const uint32_t cq = clip_of(tile);
    int ld = lane;
    asm volatile("" : "+v"(ld));      // (per-lane offsets are formed per refill: hoisted out of the tile loop they were spilled)
    stage_tile<WAVES>(y + (int64_t)cq * ldy, (int)(L * 4), stage, (int)(t0_of(tile, cq) * hop) - pad, span,
                      dma_wide != 0, w, ld);
  };
  if (LOAD == 2) dma(tile_begin);

  LaneConst lc;
  init_lane_const(lc, lane, twid);
  float* prow = Pbuf + w * P_STRIDE;
  float2* sc = reinterpret_cast<float2*>(prow);
  if (tid < 64) tw2l[(tid >> 4) * TW2_STRIDE + (tid & 15)] = twid[32 * (tid >> 4) * (tid & 15)];
  for (int i = tid; i < 15 * 64; i += NTHREADS) tw1l[i] = twid[2 * (i & 63) * ((i >> 6) + 1)];
  for (int i = tid; i < NFFT / 2; i += NTHREADS) winl[i] = win2[i];
  if (CLIPM) {
    if (TRI)
      for (int i = tid; i < 2 * n_mels * mf.tp; i += NTHREADS) clipmel[i] = 0.f;
    float* dctl = TRI ? tri_dct : clipmel + n_mels * mf.tp + WAVES;
    for (int i = tid; i < mf.n_mfcc * n_mels; i += NTHREADS) dctl[i] = mf.dct[i];
    if (mf.lifter != nullptr && tid < mf.n_mfcc) dctl[mf.n_mfcc * n_mels + tid] = mf.lifter[tid];
  }

  int* mtab = cpl + 3 * SYG_MAX_BANDS;              // [4][64]: slot first bin, slot group, group first slot, group slots
  if (!COMPLEX_OUT) {
    // pad words of the skewed rows, the row tails and the slack are read against zero weights: they must
    // hold finite values, so the whole buffer (and the slab behind it) is cleared once
    for (int i = tid; i < LM::P_FLOATS + LM::SLAB_FLOATS; i += NTHREADS) lds[i] = 0.f;
    if (TRI && wpacked != nullptr) {                    // (no table: statistics only, nothing is projected)
      // (word 1 of a lane's first 16 bytes: the band it stores -> that band's byte offset inside a mel matrix -- the
      // clip's LDS matrix [n_mels][tp], or (MODE 8 / 9) the clip's [n_mels][T] block of mel_out)
      const int band_bytes = (TRIMEL ? (int)T : mf.tp) * 4;
      for (int i = tid; i < SEG_WORDS; i += NTHREADS) {
        int v = reinterpret_cast<const int*>(wpacked)[i];
        if ((i & 3) == 1 && ((i >> 8) & 1) == 0 && v >= 0) v *= band_bytes;
        cpl[i] = v;
      }
    }
    if (!TRI || ROWFN) {
#pragma unroll
      for (int r = 0; r < SYG_MAX_BANDS; ++r)
        if (tid == r) { cplc[r] = cplan.lo[r]; cplc[SYG_MAX_BANDS + r] = cplan.hi[r]; cplc[2 * SYG_MAX_BANDS + r] = cplan.k[r]; }
    }
    if (!TRI)
      for (int i = tid; i < MTAB_INTS; i += NTHREADS) mtab[i] = reinterpret_cast<const int*>(wpacked)[plan.table_off + i];
  }
  if (LOAD == 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  float cmax = 0.f;       // MODE 3: running maximum of the clip's mel powers produced by this thread
  int64_t pend_b = -1;    // MODE 3: clip whose dB matrix waits for its DCT
  int cur = 0;            // MODE 6: which of the two mel matrices the current clip fills
  // MODE 6: output tiles of the clip epilogue; waves without a frame in a clip's last tile; whether those waves take the
  // epilogue of the clip before (at most three output tiles each: it must stay shorter than a transform)
  bool tri_scan8 = false;
  if (TRI && wpacked != nullptr) {
    unsigned lk = 0;
#pragma unroll
    for (int p = 0; p < NPASS; ++p) lk |= (unsigned)(cpl[4 * (128 * p + lane) + 2] | cpl[4 * (128 * p + lane) + 3]);
    tri_scan8 = __builtin_amdgcn_ballot_w64((lk >> 24) != 0) != 0;
  }
  const int tri_ndct = ((mf.n_mfcc + 15) >> 4) * (mf.tp >> 4);
  const int tri_idle = (TRI && !TRIMEL) ? mf.tp - (int)T : 0;
  const bool tri_defer = TRI && !TRIMEL && tri_idle > 0 && (tri_ndct + tri_idle - 1) / tri_idle <= 3;
  // staged mode: the frame of the NEXT tile is fetched (LDS -> registers) one phase ahead, so that the stage
  // buffer can be refilled behind the FFT phase; direct modes load at the top of the tile loop
  float2 v[16];
  bool have = false;
  auto fetch = [&](int64_t tile) {
    const uint32_t cq = clip_of(tile);
    const int64_t t = t0_of(tile, cq) + w;
    have = t < T;
    int lf = lane;                    // laundered like lv below: no hoisted per-lane addresses
    asm volatile("" : "+v"(lf));
    if (have) fetch_frame<LOAD>(v, y + (int64_t)cq * ldy, L, t * (int64_t)hop - pad, stage + w * hop, lf);
  };
  if (LOAD == 2) {
    fetch(tile_begin);
    __syncthreads();                                  // every wave holds its frame: the stage may be refilled
    if (tile_begin + 1 < tile_end) dma(tile_begin + 1);
  }

#if SYG_ABL == 9
  unsigned long long tacc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, tprev;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tprev)::"memory");
#endif
#pragma unroll 1
  for (int64_t tile = tile_begin; tile < tile_end; ++tile) {
    const uint32_t cq = clip_of(tile);
    const int64_t b = cq;
    const int64_t t0 = t0_of(tile, cq);
    const int64_t t = t0 + w;
    if (LOAD != 2) fetch(tile);
    // MODE 1: the row functions behind barrier B are chains of dependent wave-level steps that hide their latency only
    // behind each other; a wave that is through starts its next transform, whose dense vector work would take the
    // issue slots from the waves still in their row functions (oldest wave first at equal priority) and stretch the
    // tile.  The row functions therefore run at the top level and the transform one level lower (SYG_TAILPRIO = 1;
    // 0: all equal -- measured 428 vs 373-390 us per 1024 clips of C4; 2-4: other splits, no better).
#ifndef SYG_TAILPRIO
#define SYG_TAILPRIO 1
#endif
    // MODE 0 / 3: projection, clip epilogue and slab combine at the top level, the transform one level lower
    // (SYG_CPRIO = 3: 150.3 vs 151.0 us for the one-launch MFCC at C2; 0: the round-2 levels)
#ifndef SYG_CPRIO
#define SYG_CPRIO 3
#endif
    constexpr int PD = ROWFN ? (SYG_TAILPRIO == 2 || SYG_TAILPRIO == 3 ? 2 : SYG_TAILPRIO == 1 ? 1 : SYG_TAILPRIO == 4 ? 3 : 0)
                                   : (SYG_CPRIO == 1 || SYG_CPRIO == 3) ? 1 : 0;
#ifndef SYG_AGEPRIO
#define SYG_AGEPRIO 0
#endif
    // (wave-uniform; SYG_AGEPRIO = n > 0: the n oldest waves one level lower -- measured SLOWER: 148.0 -> 148.7 / 151.2 /
    // 152.4 us for n = 4 / 8 / 12; n < 0: the -n youngest waves one level lower)
    const bool older = (SYG_AGEPRIO > 0 && w < SYG_AGEPRIO) || (SYG_AGEPRIO < 0 && w >= WAVES + SYG_AGEPRIO);
    SETPRIO_AGE(3 - PD > 0 ? 3 - PD : 0, older);
    // The filterbank operands of this wave's slots are the same for every tile but cannot stay resident (the FFT needs
    // all 128 VGPRs): the first NPRE groups of four steps are re-fetched every tile, behind pass 3 of the transform.
    constexpr int NPRE = 7;            // unconditional: every wave's segment holds >= NPRE groups (zero padded)
#ifndef SYG_NEARLY
#define SYG_NEARLY 0
#endif
    constexpr int NEARLY = SYG_NEARLY; // ... of which this many may be requested behind pass 3 already (up to 5 fit the registers;
                                       // measured: no gain -- barrier A's wait covers the latency either way -- so 0)
    const int ng = plan.steps >> 2;
    const float4* wp4w = reinterpret_cast<const float4*>(wpacked) + (int64_t)w * ng * 64;
    float4 apre[NPRE];
    if (have) {
      // the lane id is laundered through an empty asm each iteration: the LDS / global addresses derived
      // from it are then recomputed per frame (a few integer ops) instead of being hoisted out of the tile
      // loop as ~100 loop-invariant registers that would spill
      int lv = lane;
      asm volatile("" : "+v"(lv));
#if SYG_ABL == 1
#pragma unroll
      for (int a = 0; a < 16; ++a) v[a] = make_float2((float)(lv + a) * 1e-3f, (float)(lv - a) * 1e-3f);
#endif
#if SYG_ABL == 5
      apply_window(v, win2, lv);
#else
      apply_window(v, winl, lv);
#endif
      TICK(0, v[0].x);
      float2 xs[2][4], xm[2][4], x512;
      wave_rfft2048<COMPLEX_OUT ? 0 : NEARLY, PD, X2>(v, lc, sc, tw1l, tw2l, lv, xs, xm, x512, wp4w, apre, older TPASS);
      if (COMPLEX_OUT) {
        float2* o = cout + (b * T + t) * NBIN;
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
          for (int d = 0; d < 4; ++d) {
            int k = lc.kb[u] + 256 * d;
            if (u == 0 && d >= 2 && lane == 0) k = (d == 2) ? 128 : 384;
            o[k] = xs[u][d];
            o[MC - k] = xm[u][d];
          }
        if (lane == 0) o[512] = x512;
      } else {
        int dA2 = lc.dA2, dA3 = lc.dA3;
        // (MODE 7: the four addresses formed from these are summed per frame -- kept across the tile loop they were spilled,
        // and a spill's reload waits for every outstanding memory operation)
        if (TRI && ROWFN) asm volatile("" : "+v"(dA2), "+v"(dA3));
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
          for (int d = 0; d < 4; ++d) {
            const int off = (u == 0 && d == 2) ? dA2 : (u == 0 && d == 3) ? dA3 : 272 * d;
            prow[lc.pkb[u] + off] = fmaf(xs[u][d].x, xs[u][d].x, xs[u][d].y * xs[u][d].y);
            prow[lc.pmb[u] - off] = fmaf(xm[u][d].x, xm[u][d].x, xm[u][d].y * xm[u][d].y);
          }
        if (lane == 0) prow[ppos(512)] = fmaf(x512.x, x512.x, x512.y * x512.y);
      }
    } else if (!COMPLEX_OUT && !TRI) {
      for (int k = lane; k < P_STRIDE; k += 64) prow[k] = 0.f;
#pragma unroll
      for (int q = 0; q < NEARLY; ++q) apre[q] = wp4w[q * 64 + lane];
    }
    SETPRIO(0);
    if (COMPLEX_OUT) continue;
    if (TRI) {
      // ---- MODE 6: this wave projects its own row (no barrier A, no slab, no combine); the two barriers below only
      // hand the stage buffer over: every DMA part of the next tile has landed | X1 | fetch the next frame | X2 | refill.
      // dB + DCT of a finished clip: when the clip's last tile leaves waves without a frame (tp - T of them), THEY form
      // the output tiles of the clip BEFORE while the others transform -- the epilogue then costs nothing; otherwise the
      // first waves form them right behind the clip's last tile, beside the other waves' next transform (which fills
      // the OTHER mel matrix).  Whatever is pending when the workgroup runs out of tiles is formed behind the loop.
#ifndef SYG_TRIPRIO
#define SYG_TRIPRIO 3
#endif
      SETPRIO(SYG_TRIPRIO);
      const bool mine = (t < T);
      const bool clip_done = (t0 + TILE_T >= T);
      float* cmc = clipmel + cur * (n_mels * mf.tp);
      // MODE 6, WHEN a wave projects (SYG_P6SPLIT=1; off: what pays for the row functions of MODE 7 below does not pay
      // here, the transform is 80 % of the interval): half of the waves of every SIMD (w & 4) behind the two barriers
      // instead of in front of them, with their next frame already in registers -- their column of the clip's matrix and their share of its
      // maximum arrive one barrier interval late (before X1 of the NEXT tile), which the deferred epilogue never sees (it
      // runs a whole clip later; hence: deferred epilogue only, at least two tiles per clip, a barrier in front of the
      // epilogue behind the loop).  Between two barriers every wave still transforms one frame and projects one row, but
      // one half projects (latency-bound LDS reads and scans) while the other half transforms.
      const bool proj_split = SYG_P6SPLIT && !ROWFN && LOAD == 2 && tri_defer && tiles_per_clip >= 2;
      const bool proj_late = proj_split && ((w >> 2) & 1);
      auto project = [&]() {
        int la = lane;
        asm volatile("" : "+v"(la)::"memory");
        wave_lds_sync();
        // (the table's band word was turned into the band's BYTE offset inside a mel matrix when the workgroup copied it)
        if (TRIMEL) {
          char* colg = reinterpret_cast<char*>(mel_out + (b * n_mels) * T + t);
          tri_project<4>(prow - TRI4_ROW_BASE, reinterpret_cast<const float4*>(cpl), la, tri_scan8,
                         [&](int boff, float v) { *reinterpret_cast<float*>(colg + boff) = MELSC * v; });
        } else {
          char* colb = reinterpret_cast<char*>(cmc + (int)t);
          tri_project<2>(prow, reinterpret_cast<const float4*>(cpl), la, tri_scan8, [&](int boff, float v) {
            *reinterpret_cast<float*>(colb + boff) = v;
            asm("v_max_f32_e32 %0, %0, %1" : "+v"(cmax) : "v"(v));
          });
        }
      };
      auto publish_max = [&]() {
        const float cm = wave_max(cmax);
        cmax = 0.f;
        if (lane == 0) tri_red[cur * WAVES + w] = cm;
      };
      const bool tri_proj = wpacked != nullptr;
      if (mine) {
        if (!proj_late && tri_proj) project();
      } else if (tri_defer && pend_b >= 0 && SYG_TRIX != 1) {
        float* cmp = clipmel + (cur ^ 1) * (n_mels * mf.tp);
        clip_dct<WAVES>((int)(uintptr_t)(lds_fptr)cmp, (int)(uintptr_t)(lds_fptr)(tri_red + (cur ^ 1) * WAVES),
                        (int)(uintptr_t)(lds_fptr)tri_dct, mf, n_mels, (int)T, (int)pend_b, w - (WAVES - tri_idle), lane, tri_idle);
      }
      // MODE 7: statistics / contrast of this wave's own row.  Out-of-line: the entry of such a function waits for every
      // outstanding memory operation, so the results wait in lanes of three registers and are stored -- and the stage
      // refill is issued -- behind the barriers.  Statistics first (a wide contrast band may park its lists in the row's
      // low words).  WHEN a wave runs them: half of the waves of every SIMD (w & 4) right behind their projection, the
      // other half behind X2 in front of their next transform -- between two barriers every wave does the same work, but
      // one half transforms while the other half runs its chains of dependent reductions, instead of all sixteen doing
      // the same thing at the same time (all behind X2: SYG_R7SPLIT=0; all in front of X1 was the first build, 832 us).
      float row_sres = 0.f;
      float2 row_pv = make_float2(0.f, 0.f);
      const bool row_early = SYG_R7SPLIT && ((w >> SYG_R7SHIFT) & 1);
      auto row_compute = [&]() {
        if (SYG_R7PRIO != SYG_TRIPRIO) SETPRIO(SYG_R7PRIO);
        // timing ablations (WRONG results): 1 = a trivial inline stand-in, 2 = a trivial out-of-line function
#if SYG_R7ABL == 1
        if (true) { row_sres = wave_sum(prow[17 * lane]); } else
#elif SYG_R7ABL == 2
        if (true) { row_sres = row_trivial((lds_row)prow, lane); } else
#endif
        if (SYG_ROWBOTH && stats_out != nullptr && contrast_out != nullptr) {
          const float3 f = row_features((lds_row)prow, lane, binhz, roll_percent, bw_p, smask, (lds_iptr)cplc, cplan.n_rows,
                                        cplan.ascending);
          row_sres = f.x; row_pv = make_float2(f.y, f.z);
        } else {
          if (stats_out != nullptr) row_sres = row_stats((lds_row)prow, lane, binhz, roll_percent, bw_p, smask);
          if (contrast_out != nullptr) row_pv = row_contrast_all((lds_row)prow, lane, (lds_iptr)cplc, cplan.n_rows, cplan.ascending);
        }
      };
      if (ROWFN && mine && row_early) {
        // (the three result registers wait in the wave's own row, dead until its next transform: live across the fetch of
        // the next frame they cost three spilled registers)
        row_compute();
        prow[lane] = row_sres; prow[64 + lane] = row_pv.x; prow[128 + lane] = row_pv.y;
      }
      if (CLIPM && clip_done && !proj_late) publish_max();
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();                                  // X1: the clip's columns of this tile are written too
      if (LOAD == 2) {
        have = false;
        if (tile + 1 < tile_end) fetch(tile + 1);
        __syncthreads();                                // X2: every wave holds its next frame
      }
      if (proj_late) {
        if (tile + 2 < tile_end) dma(tile + 2);
        if (mine && tri_proj) project();
        if (clip_done) publish_max();
      }
      if (CLIPM && clip_done) {
        // (clip_dct's entry waits for outstanding memory operations, so the refill is issued behind it)
        if (tri_defer) pend_b = mf.n_mfcc > 0 ? (int64_t)b : pend_b;
        else if (w < tri_ndct && SYG_TRIX != 1)
          clip_dct<WAVES>((int)(uintptr_t)(lds_fptr)cmc, (int)(uintptr_t)(lds_fptr)(tri_red + cur * WAVES),
                          (int)(uintptr_t)(lds_fptr)tri_dct, mf, n_mels, (int)T, (int)b, w, lane);
        cur ^= 1;
      }
      if (ROWFN && mine) {
        if (!row_early) row_compute();
        else { row_sres = prow[lane]; row_pv = make_float2(prow[64 + lane], prow[128 + lane]); }
        if (contrast_out != nullptr) {
          if (lane < cplan.n_rows) {
            contrast_out[((b * 2 + 0) * cplan.n_rows + lane) * T + t] = row_pv.x;
            contrast_out[((b * 2 + 1) * cplan.n_rows + lane) * T + t] = row_pv.y;
          }
        }
        if (stats_out != nullptr && lane < SYG_NSTAT && ((stats_row_mask(smask) >> lane) & 1))
          stats_out[(b * SYG_NSTAT + lane) * T + t] = row_sres;
      }
      if (LOAD == 2 && !proj_late && tile + 2 < tile_end) dma(tile + 2);
      SETPRIO(0);
      continue;
    }
#if SYG_ABL == 10 || SYG_ABL == 11
    // ablation (WRONG results): free-running waves -- no projection, no slab, no reduce, no workgroup barrier; the
    // stage hand-over is unsynchronised.  Lower bound for a design whose waves never meet (10), or meet once per
    // tile (11).
    if (LOAD == 2) {
      have = false;
      if (tile + 1 < tile_end) fetch(tile + 1);
#if SYG_ABL == 11
      __syncthreads();
#endif
      if (tile + 2 < tile_end) dma(tile + 2);
      asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    }
    if (lane == 0 && mel_out != nullptr && t < T) mel_out[(b * n_mels) * T + t] = prow[5];
    continue;
#endif
#if SYG_ABL == 9
    int tdep = lane;
    TICK(5, tdep);
#endif
    int la = lane;                     // laundered: per-lane addresses are recomputed here, not kept across the FFT
    asm volatile("" : "+v"(la)::"memory");
    const float4* wp4 = wp4w + la;
#pragma unroll
    for (int q = NEARLY; q < NPRE; ++q) apre[q] = wp4[q * 64];
    const int pslot = mtab[w * 4 + (la >> 4)];          // row POSITION (skewed, see ppos()) of this lane's slot
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (staged mode: the next tile's samples have landed too)
    __syncthreads();                                    // barrier A: rows complete
    TICK(6, tdep);
    if (!ROWFN && SYG_CPRIO == 3) SETPRIO(3);           // projection (+ clip epilogue) at the top level
    if (CLIPM && pend_b >= 0) {
#ifndef SYG_DCTSHIFT
#define SYG_DCTSHIFT 0
#endif
      {
        const int wd = (w + WAVES - SYG_DCTSHIFT) % WAVES;       // output tile wd is formed by wave (wd + SYG_DCTSHIFT) mod WAVES
        if (wd < ((mf.n_mfcc + 15) >> 4) * (mf.tp >> 4) && SYG_ABL != 6 && SYG_ABL != 7)
          clip_dct<WAVES>((int)(uintptr_t)(lds_fptr)clipmel, (int)(uintptr_t)(lds_fptr)(clipmel + n_mels * mf.tp),
                          (int)(uintptr_t)(lds_fptr)(clipmel + n_mels * mf.tp + WAVES), mf, n_mels, (int)T, (int)pend_b, wd, lane);
      }
      pend_b = -1;
      TICK(10, tdep);
    }

    // ---- phase 2: block-sparse mel projection on the matrix cores, v_mfma_f32_4x4x1_16b_f32: sixteen independent
    // 4 x 4 outer-product blocks per instruction.  Block b = lane >> 2 = (slot s = b >> 2, frame group h = b & 3):
    // A = four mel rows of the slot's group at the slot's current row position (lane & 3 = row), B = the power at that
    // position in the four frames 4 h + (lane & 3), D[row][frame] accumulates in four registers.  A slot walks
    // CONSECUTIVE WORDS of the skewed power rows -- the pad word after every 16 bins carries a zero weight -- so every
    // B read is one base register plus an immediate offset, all of them are in flight together, and the MFMAs then
    // issue back to back.  Only the non-zero ranges of the 4-row groups are multiplied (1221 bin-steps per tile at
    // 40 mels against 4544 for 16-row tiles).
    {
      const int fr = (4 * ((la >> 2) & 3) + (la & 3)) & (TILE_T - 1);       // frame of this lane's block column
      const float* pq = Pbuf + fr * P_STRIDE + pslot;     // 8-byte aligned: P_STRIDE and the slot positions are even
      float bq[4 * NPRE];
#pragma unroll
      for (int i = 0; i < 2 * NPRE; ++i) {                 // two positions per LDS read (half the LDS cycles of b32 reads)
        const float2 v2 = reinterpret_cast<const float2*>(pq)[i];
        bq[2 * i] = v2.x; bq[2 * i + 1] = v2.y;
      }
      // two accumulators (even / odd positions): consecutive matrix instructions do not wait for each other
      v4f acc = {0.f, 0.f, 0.f, 0.f}, acc2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int q = 0; q < NPRE; ++q) {
        acc = __builtin_amdgcn_mfma_f32_4x4x1f32(apre[q].x, bq[4 * q + 0], acc, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f32_4x4x1f32(apre[q].y, bq[4 * q + 1], acc2, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_4x4x1f32(apre[q].z, bq[4 * q + 2], acc, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f32_4x4x1f32(apre[q].w, bq[4 * q + 3], acc2, 0, 0, 0);
      }
      acc += acc2;
      for (int q = NPRE; q < ng; ++q) {       // (filterbanks whose chunks are longer than 28 row positions)
        const float4 a4 = wp4[q * 64];
        const float* pb = pq + 4 * q;
        acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a4.x, pb[0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a4.y, pb[1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a4.z, pb[2], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a4.w, pb[3], acc, 0, 0, 0);
      }
      // partial sums of (slot, mel row r, frame): slab[wave][slot][r][frame]
      if (4 * ((la >> 2) & 3) + (la & 3) < TILE_T) {
        float* sl = slab + (w * 4 + (la >> 4)) * (4 * TILE_T) + fr;
#pragma unroll
        for (int r = 0; r < 4; ++r) sl[r * TILE_T] = acc[r];
      }
      TICK(7, acc[0]);
    }
    // ---- staged mode: the next tile's frame (LDS -> registers); once every wave holds its frame (barrier B)
    // the stage is refilled with the tile after next.  The statistics of MODE 1 run behind barrier B with the fetched
    // frame live in callee-saved registers (the out-of-line row functions stay inside the caller-saved ones), so that
    // mode needs no barrier of its own.
    constexpr bool FETCH_EARLY = (LOAD == 2);
    if (FETCH_EARLY) {
      have = false;
      if (tile + 1 < tile_end) fetch(tile + 1);
    }
    __syncthreads();                // barrier B: slab complete (and every wave has read its staged frame)
    TICK(8, tdep);
    if (!ROWFN && (SYG_CPRIO == 1 || SYG_CPRIO == 2)) SETPRIO(3);         // experiment: the slab combine at the top level
    const bool clip_done = CLIPM && (t0 + TILE_T >= T);
    // MODE 1 runs out-of-line row functions below: a function entry waits for EVERY outstanding memory operation
    // (s_waitcnt vmcnt(0) -- the callee cannot know the caller's counters), so nothing may be in flight when they are
    // called: the refill of the stage buffer is issued behind them and the tile's stores behind the last call.
    constexpr bool TAIL = ROWFN;
    if (!TAIL && FETCH_EARLY && tile + 2 < tile_end) dma(tile + 2);

    // ---- combine the slots of each mel group in a fixed order (ascending bins): one wave per group of four mel rows,
    // lane = (row, frame); the slots of a group are consecutive.  (MODE 1: behind the row functions -- the slab stays
    // valid until the next tile's projection -- so that their entry does not wait for these stores.)
    auto combine = [&]() {
      constexpr int GL = 4 * TILE_T;               // outputs per group and tile (64 at 16 frames)
#ifndef SYG_REDSHIFT
#define SYG_REDSHIFT 6
#endif
      // group g is combined by wave (g + SYG_REDSHIFT) mod WAVES.  Ten groups at 40 mels: with shift 0 the ten OLDEST waves
      // carry the combine and the six youngest -- which the arbiter already serves last -- none; shift 6 gives it to
      // the waves 6 .. 15 (148.5 vs 150.8 us for the one-launch MFCC at C2; 3: 149.1, 10: slower than 6)
      for (int g = (w + WAVES - SYG_REDSHIFT) % WAVES; g < plan.n_groups; g += WAVES) {
        const int first = __builtin_amdgcn_readfirstlane(mtab[128 + g]);
        const int cnt = __builtin_amdgcn_readfirstlane(mtab[192 + g]);
        int lq = lane;                    // laundered (see lv above): no hoisted per-lane addresses that would spill
        asm volatile("" : "+v"(lq));
        if (GL == 64 || lq < GL) {
          const int m = lq / TILE_T, tt = lq & (TILE_T - 1);
          const float* sp = slab + first * GL + lq;
          float sum = 0.f;
          int q = 0;
          for (; q + 4 <= cnt; q += 4) {
            const float a0 = sp[0], a1 = sp[GL], a2 = sp[2 * GL], a3 = sp[3 * GL];
            sum += a0; sum += a1; sum += a2; sum += a3;
            sp += 4 * GL;
          }
          for (; q < cnt; ++q) { sum += sp[0]; sp += GL; }
          const int mel = g * 4 + m;
          if (CLIPM) {
            if (mel < n_mels) {
              clipmel[mel * mf.tp + (int)t0 + tt] = sum;     // frames >= T hold 0 (rows were cleared)
              cmax = fmaxf(cmax, sum);                        // power is non-negative
            }
            if (SYG_ABL != 9 && mel_out != nullptr && mel < n_mels && t0 + tt < T) mel_out[(b * n_mels + mel) * T + t0 + tt] = MELSC * sum;
          } else {
            if (mel < n_mels && t0 + tt < T) mel_out[(b * n_mels + mel) * T + t0 + tt] = MELSC * sum;
          }
        }
      }
    };
    auto publish = [&]() {
      // last tile of the clip: publish the per-wave maxima; the dB + DCT epilogue runs in the next tile's
      // projection phase (or behind the loop)
      const float cm = wave_max(cmax);
      cmax = 0.f;
      if (lane == 0) clipmel[n_mels * mf.tp + w] = cm;
      pend_b = b;
    };
    if (!TAIL) {
      combine();
      if (clip_done) publish();
    }

    TICK(9, tdep);
    // ---- phase 2b: per-frame statistics / contrast means from the same LDS rows
    if (ROWFN && (stats_out != nullptr || contrast_out != nullptr)) {
      if (SYG_TAILPRIO == 1 || SYG_TAILPRIO == 3 || SYG_TAILPRIO == 4) SETPRIO(3);
      if (SYG_TAILPRIO == 2) { if (w >= WAVES / 2) SETPRIO(3); else SETPRIO(2); }   // the younger half would otherwise run on leftovers
      if (t < T) {
        // statistics first (a wide contrast band parks its lists in the row's low words); every result waits in a lane
        // of a register and is stored behind the last call
        float sres = 0.f, pk = 0.f, vl = 0.f;
        // timing ablations (WRONG results): 20 = no row function is called, 21 = a trivial inline stand-in
#if SYG_ABL == 20
        if (false) {
#elif SYG_ABL == 21
        if (stats_out != nullptr) sres = wave_sum(prow[17 * lane]);
        if (false) {
#else
        const bool both = SYG_ROWBOTH && stats_out != nullptr && contrast_out != nullptr;
        if (both) {
          const float3 f = row_features((lds_row)prow, lane, binhz, roll_percent, bw_p, smask, (lds_iptr)cplc, cplan.n_rows,
                                        cplan.ascending);
          sres = f.x; pk = f.y; vl = f.z;
        } else if (stats_out != nullptr) sres = row_stats((lds_row)prow, lane, binhz, roll_percent, bw_p, smask);
        if (contrast_out != nullptr) {
#endif
          if (!both) {
            const float2 pv = row_contrast_all((lds_row)prow, lane, (lds_iptr)cplc, cplan.n_rows, cplan.ascending);
            pk = pv.x; vl = pv.y;
          }
          if (lane < cplan.n_rows) {
            contrast_out[((b * 2 + 0) * cplan.n_rows + lane) * T + t] = pk;
            contrast_out[((b * 2 + 1) * cplan.n_rows + lane) * T + t] = vl;
          }
        }
#if SYG_ABL == 9
        if (stats_out != nullptr && sres == 12345.678f)      // (timeline build: stats_out carries the phase counters)
#else
        if (stats_out != nullptr && lane < SYG_NSTAT && ((stats_row_mask(smask) >> lane) & 1))
#endif
          stats_out[(b * SYG_NSTAT + lane) * T + t] = sres;
      }
      // (a wave only reads and -- parking sorted lists -- overwrites ITS OWN row here; every projection that read the
      // row finished before barrier B, and the row is next written by this wave's own FFT: no barrier needed)
    }
    if (TAIL) {
      if (SYG_TAILPRIO != 0) SETPRIO(0);
      TICK(11, tdep);
      combine();
      if (clip_done) publish();
      if (FETCH_EARLY && tile + 2 < tile_end) dma(tile + 2);
    }
  }
  if (TRIMEL) {
    // (no clip epilogue: the mel columns are in HBM)
  } else if (TRI) {
    // (deferred epilogue: the last clip's matrix is complete behind X1 of its last tile -- or, with the projections of half
    // of the waves behind the barriers, behind one more)
    if (SYG_P6SPLIT && !ROWFN && LOAD == 2 && tri_defer && tiles_per_clip >= 2) __syncthreads();
    if (pend_b >= 0 && w < tri_ndct && SYG_TRIX != 1)
      clip_dct<WAVES>((int)(uintptr_t)(lds_fptr)(clipmel + (cur ^ 1) * (n_mels * mf.tp)),
                      (int)(uintptr_t)(lds_fptr)(tri_red + (cur ^ 1) * WAVES), (int)(uintptr_t)(lds_fptr)tri_dct, mf, n_mels,
                      (int)T, (int)pend_b, w, lane);
  } else if (CLIPM && pend_b >= 0) {
    __syncthreads();
    if (w < ((mf.n_mfcc + 15) >> 4) * (mf.tp >> 4) && SYG_ABL != 6)
      clip_dct<WAVES>((int)(uintptr_t)(lds_fptr)clipmel, (int)(uintptr_t)(lds_fptr)(clipmel + n_mels * mf.tp),
                      (int)(uintptr_t)(lds_fptr)(clipmel + n_mels * mf.tp + WAVES), mf, n_mels, (int)T, (int)pend_b, w, lane);
  }
#if SYG_ABL == 9
  if (MODE == 3 && lane < 12)
    mel_out[((int64_t)blockIdx.x * WAVES + w) * 16 + lane] = (float)tacc[lane] / (float)(tile_end - tile_begin);
  if (MODE == 1 && lane < 12)
    stats_out[((int64_t)blockIdx.x * WAVES + w) * 16 + lane] = (float)tacc[lane] / (float)(tile_end - tile_begin);
#endif
}

template <int WAVES, bool SLAB = true, int NPASS = 2>
constexpr size_t lds_bytes() {
  return (size_t)Lds<WAVES, SLAB, NPASS>::TOTAL * sizeof(float);
}

// Workgroups per CU: two of 8 waves or one of 16; each takes a contiguous chunk of tiles.
// syg_set_option(SYG_OPT_RESERVED_CUS, n) leaves n CUs out of the grid.  A workgroup of these kernels fills a
// CU (16 waves x 128 VGPRs), so a kernel of another stream that needs a few CUs at the same time -- RCCL's send /
// receive workgroups while the previous batch is gathered -- either waits for a whole launch or makes this launch wait
// for it (tools/queue_bench.py: 167 -> 256 us for every second launch).  With the CUs set aside both run side by side.
void persistent_grid(int64_t total_tiles, int waves, int& wgs, int& per) {
  // the CU count of the CURRENT device, asked at every call (an attribute query, no device properties round trip):
  // no process-wide cache that a second device or a second thread could read stale
  int n_cu = 0, dev = 0;
  if (hipGetDevice(&dev) != hipSuccess ||
      hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n_cu <= 0)
    n_cu = 256;
  int use_cu = n_cu;
  const int r = option(SYG_OPT_RESERVED_CUS);
  if (r > 0 && r < n_cu) use_cu = n_cu - r;
  const int64_t slots = (int64_t)use_cu * (waves == 8 ? 2 : 1);
  int64_t p = (total_tiles + slots - 1) / slots;
  if (p < 1) p = 1;
  per = (int)p;
  wgs = (int)((total_tiles + p - 1) / p);
}

// SYG_OPT_STFT_LOAD = 0 | 1 | 2 forces the frame load path (the tests compare the three); default: staged tiles
int load_mode() {
  const int v = option(SYG_OPT_STFT_LOAD);
  return (v >= 0 && v <= 2) ? v : 2;
}

int check_common(const float* y, int64_t B, int64_t L, int64_t ldy, int hop, int center, int64_t T,
                 const float* window, const float* twiddle, int waves) {
  SYG_REQUIRE(y && window && twiddle, "stft2048: null pointer argument");
  SYG_REQUIRE(B >= 1 && L >= 1 && ldy >= L, "stft2048: need B >= 1, L >= 1, ldy >= L (B=%lld L=%lld ldy=%lld)",
              (long long)B, (long long)L, (long long)ldy);
  SYG_REQUIRE(hop >= 1, "stft2048: hop must be >= 1 (got %d)", hop);
  const int64_t Texp = center ? 1 + L / hop : (L >= NFFT ? 1 + (L - NFFT) / hop : 0);
  SYG_REQUIRE(T >= 1 && T == Texp, "stft2048: T=%lld does not match the framing rule (%lld)", (long long)T,
              (long long)Texp);
  SYG_REQUIRE(B * ((T + waves - 1) / waves) < (int64_t)0x7fffffff, "stft2048: grid too large");
  return SYG_OK;
}

constexpr size_t LDS_LIMIT = 160 * 1024;

template <int WAVES, int MODE>
int launch(int load, const float* y, int64_t B, int64_t L, int64_t ldy, int hop, int center, int64_t T,
           const float* window, const float* twiddle, const float* wpacked, const MelPlan& plan, int n_mels,
           float* mel_out, float binhz, float roll_percent, float bw_p, int smask, float* stats_out,
           const ContrastPlan& cp,
           float* contrast_out, float* cout, hipStream_t st, MfccArgs mf = MfccArgs()) {
  const int pad = center ? NFFT / 2 : 0;
  const int tiles = (int)((T + WAVES - 1) / WAVES);
  const int64_t total_tiles = B * tiles;
  int wgs = 0, per = 0;
  persistent_grid(total_tiles, WAVES, wgs, per);
  constexpr bool TRIMEL = (MODE == 8 || MODE == 9);      // tile form of the segment-sum projection, four-pass table
  constexpr bool TRI = (MODE == 6 || MODE == 7);
  size_t lds = TRIMEL ? lds_bytes<WAVES, false, 4>() : lds_bytes<WAVES, !TRI>();
  if (TRIMEL) mf.tp = tiles * WAVES;
  if (MODE == 3 || MODE == 5 || TRI) {
    // whole clips per workgroup; the clip's mel matrix [n_mels][tiles * WAVES] sits behind the fixed LDS map
    int cw = 0, cper = 0;
    persistent_grid(B, WAVES, cw, cper);
    per = cper * tiles;
    wgs = cw;
    mf.tp = tiles * WAVES;
    if (X2_MEL && (MODE == 3 || MODE == 6)) { mf.amin *= 4.f; mf.ref_value *= 4.f; }     // the clip's mel matrix holds 4 x mel (exact scaling)
    lds += ((size_t)(TRI ? 2 : 1) * ((size_t)n_mels * mf.tp + WAVES) + (size_t)mf.n_mfcc * (n_mels + 1)) * sizeof(float);
    SYG_REQUIRE(lds <= LDS_LIMIT, "stft2048_mfcc: the clip's mel matrix (%d x %d) does not fit the LDS left over (%zu B > %zu B); "
                "use syg_stft2048_mel_f32 + syg_logmel_dct_f32", n_mels, mf.tp, lds, LDS_LIMIT);
  }
  // staged tiles (LDS-DMA) need the tile's sample run to fit the stage buffer and 32-bit byte offsets
  const bool can_stage = (MODE != 2) && hop <= 512 && L < ((int64_t)1 << 28);
  if (load == 2 && !can_stage) load = 1;
  const bool vec2 = (hop % 2 == 0) && (ldy % 2 == 0) && (((uintptr_t)y) % 8 == 0);
  if (load == 1 && !vec2) load = 0;
  const int dma_wide = (hop % 4 == 0) && (pad % 4 == 0) && (ldy % 4 == 0) && (L % 4 == 0) && (((uintptr_t)y) % 16 == 0);
  auto kern = load == 2 ? stft2048_kernel<WAVES, 2, MODE>
                        : load == 1 ? stft2048_kernel<WAVES, 1, MODE> : stft2048_kernel<WAVES, 0, MODE>;
  {
    // set at every launch: the attribute belongs to the (function, device) pair, and a per-process "already set"
    // flag would leave a second device without it
    const size_t cap = (MODE == 3 || MODE == 5 || TRI) ? LDS_LIMIT : TRIMEL ? lds_bytes<WAVES, false, 4>() : lds_bytes<WAVES>();
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)cap);
    if (e != hipSuccess) {
      set_error("stft2048: cannot reserve %zu B LDS: %s", cap, hipGetErrorString(e));
      return SYG_E_LAUNCH;
    }
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)wgs), dim3(WAVES * 64), lds, st, y, L, ldy, hop, pad, T,
                     tiles, total_tiles, per, (const float2*)window, (const float2*)twiddle, wpacked, plan, n_mels,
                     mel_out, binhz, roll_percent, bw_p, smask, stats_out, cp, contrast_out, (float2*)cout, dma_wide, mf);
  SYG_CHECK_LAUNCH("stft2048");
  return SYG_OK;
}

}  // namespace
}  // namespace syg

using namespace syg;

namespace syg {
namespace {
int parse_contrast_plan(const float* contrast_out, const int32_t* cplan_host, ContrastPlan& cp) {
  memset(&cp, 0, sizeof(cp));
  if (!contrast_out) return SYG_OK;
  SYG_REQUIRE(cplan_host, "stft2048: contrast_out given without cplan_host");
  cp.n_rows = cplan_host[0];
  SYG_REQUIRE(cp.n_rows >= 1 && cp.n_rows <= SYG_MAX_BANDS, "stft2048: contrast rows must be in [1, %d]", SYG_MAX_BANDS);
  for (int r = 0; r < cp.n_rows; ++r) {
    cp.lo[r] = cplan_host[1 + r];
    cp.hi[r] = cplan_host[1 + SYG_MAX_BANDS + r];
    cp.k[r] = cplan_host[1 + 2 * SYG_MAX_BANDS + r];
    SYG_REQUIRE(cp.lo[r] >= 0 && cp.hi[r] <= NBIN && cp.lo[r] < cp.hi[r] && cp.k[r] >= 1 && cp.k[r] <= cp.hi[r] - cp.lo[r],
                "stft2048: contrast band %d invalid (lo=%d hi=%d k=%d)", r, cp.lo[r], cp.hi[r], cp.k[r]);
  }
  cp.ascending = 1;
  for (int r = 1; r < cp.n_rows; ++r)
    if (cp.lo[r] < cp.hi[r - 1] - 1 || cp.hi[r] < cp.hi[r - 1]) cp.ascending = 0;   // (a band may include the bin below it)
  return SYG_OK;
}

int parse_mel_plan(const char* who, const int32_t* plan_host, int n_mels, MelPlan& plan) {
  // plan_host: {2 (layout version), waves, steps, n_groups, table_off}
  SYG_REQUIRE(plan_host[0] == 2, "%s: mel plan layout %d, this library needs layout 2 (sygnals_amd._tables.pack_mel_plan)",
              who, plan_host[0]);
  const int waves = plan_host[1];
  SYG_REQUIRE(waves == 8 || waves == 16, "%s: plan must be built for 8 or 16 waves (got %d)", who, waves);
  plan.steps = plan_host[2];
  plan.n_groups = plan_host[3];
  plan.table_off = plan_host[4];
  SYG_REQUIRE(n_mels >= 1 && plan.n_groups == (n_mels + 3) / 4 && plan.n_groups <= 64,
              "%s: plan has %d groups of four mel rows, n_mels=%d needs %d (at most 64)", who, plan.n_groups, n_mels,
              (n_mels + 3) / 4);
  SYG_REQUIRE(plan.steps >= 28 && plan.steps % 4 == 0 && plan.steps <= P_STRIDE, "%s: bad step count %d", who, plan.steps);
  SYG_REQUIRE(plan.table_off >= waves * plan.steps * 64 && plan.table_off % 4 == 0, "%s: bad table offset", who);
  static_assert(P_STRIDE % 2 == 0, "slot reads are 8-byte words");
  return SYG_OK;
}
}  // namespace
}  // namespace syg

extern "C" int syg_stft2048_mel_f32(const float* y, int64_t B, int64_t L, int64_t ldy, int hop, int center,
                                    int64_t T, const float* window, const float* twiddle, const float* wpacked,
                                    const int32_t* plan_host, int n_mels, float* mel_out, float sr,
                                    float roll_percent, float bw_p, int stats_mask, float* stats_out,
                                    const int32_t* cplan_host,
                                    float* contrast_out, void* stream) {
  SYG_REQUIRE(wpacked && plan_host && mel_out, "stft2048_mel: null pointer argument");
  MelPlan plan;
  int rc = parse_mel_plan("stft2048_mel", plan_host, n_mels, plan);
  if (rc) return rc;
  const int waves = plan_host[1];
  rc = check_common(y, B, L, ldy, hop, center, T, window, twiddle, waves);
  if (rc) return rc;
  ContrastPlan cp;
  rc = parse_contrast_plan(contrast_out, cplan_host, cp);
  if (rc) return rc;
  if (stats_out) SYG_REQUIRE(sr > 0.f && roll_percent >= 0.f && roll_percent <= 1.f && bw_p > 0.f &&
                                 (stats_mask & 31) != 0 && stats_mask > 0 && stats_mask < 64 && T < ((int64_t)1 << 27),
                             "stft2048_mel: invalid statistics parameters");
  const bool extra = (stats_out != nullptr) || (contrast_out != nullptr);
  const int load = load_mode();
  const float binhz = sr / (float)NFFT;
  hipStream_t st = (hipStream_t)stream;
#define SYG_LAUNCH(W, M)                                                                                        \
  launch<W, M>(load, y, B, L, ldy, hop, center, T, window, twiddle, wpacked, plan, n_mels, mel_out, binhz,      \
               roll_percent, bw_p, stats_mask, stats_out, cp, contrast_out, nullptr, st)
  if (waves == 8) return extra ? SYG_LAUNCH(8, 1) : SYG_LAUNCH(8, 0);
  return extra ? SYG_LAUNCH(16, 1) : SYG_LAUNCH(16, 0);
#undef SYG_LAUNCH
}

// 1 when the clip-resident form has room for the clip's mel matrix + DCT rows + lifter behind the fixed LDS map
extern "C" int syg_stft2048_mfcc_fits(int n_mels, int64_t T, int n_mfcc) {
  if (n_mels < 1 || n_mels > 16 * MAXW || T < 1 || n_mfcc < 1 || n_mfcc > n_mels) return 0;
  const int64_t tp = ((T + MAXW - 1) / MAXW) * MAXW;
  const int64_t bytes = (int64_t)lds_bytes<16>() + ((int64_t)n_mels * tp + 16 + (int64_t)n_mfcc * (n_mels + 1)) * 4;
  return bytes <= (int64_t)LDS_LIMIT ? 1 : 0;
}

extern "C" int syg_stft2048_mfcc_f32(const float* y, int64_t B, int64_t L, int64_t ldy, int hop, int center,
                                     int64_t T, const float* window, const float* twiddle, const float* wpacked,
                                     const int32_t* plan_host, int n_mels, const float* dct, int n_mfcc,
                                     const float* lifter, float amin, float top_db, int ref_is_max, float ref_value,
                                     float* mel_out, float* mfcc_out, void* stream) {
  SYG_REQUIRE(wpacked && plan_host && dct && mfcc_out, "stft2048_mfcc: null pointer argument");
  MelPlan plan;
  int rc = parse_mel_plan("stft2048_mfcc", plan_host, n_mels, plan);
  if (rc) return rc;
  SYG_REQUIRE(plan_host[1] == 16, "stft2048_mfcc: needs a 16-wave plan (got %d)", plan_host[1]);
  rc = check_common(y, B, L, ldy, hop, center, T, window, twiddle, 16);
  if (rc) return rc;
  SYG_REQUIRE(n_mfcc >= 1 && n_mfcc <= n_mels, "stft2048_mfcc: need 1 <= n_mfcc <= n_mels (n_mfcc=%d n_mels=%d)",
              n_mfcc, n_mels);
  SYG_REQUIRE(amin >= 1.17549435e-38f, "stft2048_mfcc: amin must be strictly positive (a normal float)");
  SYG_REQUIRE(ref_is_max == 0 || ref_is_max == 1, "stft2048_mfcc: ref_is_max must be 0 or 1");
  SYG_REQUIRE(T < ((int64_t)1 << 24), "stft2048_mfcc: clip too long");
  ContrastPlan cp;
  memset(&cp, 0, sizeof(cp));
  MfccArgs mf;
  mf.dct = dct; mf.lifter = lifter; mf.out = mfcc_out; mf.n_mfcc = n_mfcc; mf.ref_is_max = ref_is_max;
  mf.ref_value = ref_value; mf.amin = amin; mf.top_db = top_db; mf.tp = 0; mf.rows_per_clip = n_mfcc;
  return launch<16, 3>(load_mode(), y, B, L, ldy, hop, center, T, window, twiddle, wpacked, plan, n_mels, mel_out, 0.f,
                       0.f, 0.f, 0, nullptr, cp, nullptr, nullptr, (hipStream_t)stream, mf);
}

// MODE 6: syg_stft2048_mfcc_f32 for TRIANGULAR filterbanks, the mel projection by segment sums inside each wave
// (tri_project; no weight matrix, no workgroup barrier in the projection).  segtab: the piece table of
// sygnals_amd._tables.pack_mel_segments ([2][2][64][4] words on the device, n_segtab = 1024).  Same results as
// the matrix form to rounding (both sum in float32; the affine pieces reproduce the float32 weights to 1e-7 of the
// largest -- checked on the host when the table is built).
extern "C" int syg_stft2048_mfcc_tri_fits(int n_mels, int64_t T, int n_mfcc) {
  if (n_mels < 1 || n_mels > 127 || T < 1 || n_mfcc < 1 || n_mfcc > n_mels) return 0;
  const int64_t tp = ((T + MAXW - 1) / MAXW) * MAXW;
  const int64_t bytes = (int64_t)lds_bytes<16, false>() + (2 * ((int64_t)n_mels * tp + 16) + (int64_t)n_mfcc * (n_mels + 1)) * 4;
  return bytes <= (int64_t)LDS_LIMIT ? 1 : 0;
}

extern "C" int syg_stft2048_mfcc_tri_f32(const float* y, int64_t B, int64_t L, int64_t ldy, int hop, int center,
                                         int64_t T, const float* window, const float* twiddle, const float* segtab,
                                         int n_segtab, int n_mels, const float* dct, int n_mfcc, const float* lifter,
                                         float amin, float top_db, int ref_is_max, float ref_value, float* mfcc_out,
                                         void* stream) {
  SYG_REQUIRE(segtab && dct && mfcc_out, "stft2048_mfcc_tri: null pointer argument");
  SYG_REQUIRE(n_segtab == SEGTAB_WORDS, "stft2048_mfcc_tri: the piece table has %d words, this library reads %d "
              "(sygnals_amd._tables.pack_mel_segments)", n_segtab, SEGTAB_WORDS);
  SYG_REQUIRE(((uintptr_t)segtab) % 16 == 0, "stft2048_mfcc_tri: the piece table must be 16-byte aligned");
  int rc = check_common(y, B, L, ldy, hop, center, T, window, twiddle, 16);
  if (rc) return rc;
  SYG_REQUIRE(n_mels >= 1 && n_mels <= 127 && n_mfcc >= 1 && n_mfcc <= n_mels,
              "stft2048_mfcc_tri: need 1 <= n_mfcc <= n_mels <= 127 (n_mfcc=%d n_mels=%d)", n_mfcc, n_mels);
  SYG_REQUIRE(amin >= 1.17549435e-38f, "stft2048_mfcc_tri: amin must be strictly positive (a normal float)");
  SYG_REQUIRE(ref_is_max == 0 || ref_is_max == 1, "stft2048_mfcc_tri: ref_is_max must be 0 or 1");
  SYG_REQUIRE(T < ((int64_t)1 << 24), "stft2048_mfcc_tri: clip too long");
  ContrastPlan cp;
  memset(&cp, 0, sizeof(cp));
  MelPlan plan;
  memset(&plan, 0, sizeof(plan));
  MfccArgs mf;
  mf.dct = dct; mf.lifter = lifter; mf.out = mfcc_out; mf.n_mfcc = n_mfcc; mf.ref_is_max = ref_is_max;
  mf.ref_value = ref_value; mf.amin = amin; mf.top_db = top_db; mf.tp = 0; mf.rows_per_clip = n_mfcc;
  return launch<16, 6>(load_mode(), y, B, L, ldy, hop, center, T, window, twiddle, segtab, plan, n_mels, nullptr, 0.f,
                       0.f, 0.f, 0, nullptr, cp, nullptr, nullptr, (hipStream_t)stream, mf);
}

// MODE 7: syg_stft2048_features_f32 (MODE 5: MFCC rows + statistics rows + contrast tail means from ONE launch) with the
// segment-sum projection of syg_stft2048_mfcc_tri_f32 -- BASELINE config C4 without a mel matrix in HBM and without
// the projection's barriers; the clip epilogue runs on the waves that have no frame (see MODE 6).
extern "C" int syg_stft2048_features_tri_f32(const float* y, int64_t B, int64_t L, int64_t ldy, int hop, int center, int64_t T,
                                             const float* window, const float* twiddle, const float* segtab, int n_segtab,
                                             int n_mels, const float* dct, int n_mfcc, const float* lifter, float amin,
                                             float top_db, int ref_is_max, float ref_value, float sr, float roll_percent,
                                             float bw_p, int stats_mask, float* stats_out, const int32_t* cplan_host,
                                             float* contrast_out, float* mfcc_out, int mfcc_rows_per_clip, void* stream) {
  SYG_REQUIRE(segtab && dct && mfcc_out, "stft2048_features_tri: null pointer argument");
  SYG_REQUIRE(stats_out || contrast_out, "stft2048_features_tri: no statistics requested (use syg_stft2048_mfcc_tri_f32)");
  SYG_REQUIRE(n_segtab == SEGTAB_WORDS, "stft2048_features_tri: the piece table has %d words, this library reads %d "
              "(sygnals_amd._tables.pack_mel_segments)", n_segtab, SEGTAB_WORDS);
  SYG_REQUIRE(((uintptr_t)segtab) % 16 == 0, "stft2048_features_tri: the piece table must be 16-byte aligned");
  int rc = check_common(y, B, L, ldy, hop, center, T, window, twiddle, 16);
  if (rc) return rc;
  SYG_REQUIRE(n_mels >= 1 && n_mels <= 127 && n_mfcc >= 1 && n_mfcc <= n_mels && mfcc_rows_per_clip >= n_mfcc,
              "stft2048_features_tri: need 1 <= n_mfcc <= n_mels <= 127 and mfcc_rows_per_clip >= n_mfcc");
  SYG_REQUIRE(amin >= 1.17549435e-38f, "stft2048_features_tri: amin must be strictly positive (a normal float)");
  SYG_REQUIRE(ref_is_max == 0 || ref_is_max == 1, "stft2048_features_tri: ref_is_max must be 0 or 1");
  SYG_REQUIRE(T < ((int64_t)1 << 24), "stft2048_features_tri: clip too long");
  ContrastPlan cp;
  rc = parse_contrast_plan(contrast_out, cplan_host, cp);
  if (rc) return rc;
  if (stats_out) SYG_REQUIRE(sr > 0.f && roll_percent >= 0.f && roll_percent <= 1.f && bw_p > 0.f && (stats_mask & 31) != 0 &&
                                 stats_mask > 0 && stats_mask < 64, "stft2048_features_tri: invalid statistics parameters");
  MelPlan plan;
  memset(&plan, 0, sizeof(plan));
  MfccArgs mf;
  mf.dct = dct; mf.lifter = lifter; mf.out = mfcc_out; mf.n_mfcc = n_mfcc; mf.ref_is_max = ref_is_max;
  mf.ref_value = ref_value; mf.amin = amin; mf.top_db = top_db; mf.tp = 0; mf.rows_per_clip = mfcc_rows_per_clip;
  return launch<16, 7>(load_mode(), y, B, L, ldy, hop, center, T, window, twiddle, segtab, plan, n_mels, nullptr,
                       sr / (float)NFFT, roll_percent, bw_p, stats_mask, stats_out, cp, contrast_out, nullptr,
                       (hipStream_t)stream, mf);
}

// MODE 8 / 9: the TILE form of the segment-sum projection with a FOUR-pass piece table (up to 256 pieces: the reference's
// default filterbank of 128 bands, manager.py:214, and 64 ... 128 bands at the usual sample rates, which have no two-pass
// table) -- samples in, mel POWER out [B, n_mels, T] (no weight matrix, no projection barriers, every frame's column written
// by the wave that transformed it); syg_logmel_dct_f32 is the second launch of an MFCC.  Optional statistics / contrast
// rows from the same launch (MODE 9: the row functions of syg_stft2048_features_tri_f32).  Tiles are shared out evenly
// over the workgroups, so one long clip (BASELINE config C1) fills the chip.
//   segtab   pack_mel_segments(..., n_pass=4, row_base=4): [4][2][64][4] words, n_segtab = 2048
extern "C" int syg_stft2048_mel_tri_f32(const float* y, int64_t B, int64_t L, int64_t ldy, int hop, int center, int64_t T,
                                        const float* window, const float* twiddle, const float* segtab, int n_segtab,
                                        int n_mels, float* mel_out, float sr, float roll_percent, float bw_p, int stats_mask,
                                        float* stats_out, const int32_t* cplan_host, float* contrast_out, void* stream) {
  SYG_REQUIRE(segtab && mel_out, "stft2048_mel_tri: null pointer argument");
  SYG_REQUIRE(n_segtab == SEGTAB4_WORDS, "stft2048_mel_tri: the piece table has %d words, this library reads %d "
              "(sygnals_amd._tables.pack_mel_segments(..., n_pass=4, row_base=4))", n_segtab, SEGTAB4_WORDS);
  SYG_REQUIRE(((uintptr_t)segtab) % 16 == 0, "stft2048_mel_tri: the piece table must be 16-byte aligned");
  int rc = check_common(y, B, L, ldy, hop, center, T, window, twiddle, 16);
  if (rc) return rc;
  SYG_REQUIRE(n_mels >= 1 && n_mels <= 255, "stft2048_mel_tri: need 1 <= n_mels <= 255 (got %d)", n_mels);
  SYG_REQUIRE(T * (int64_t)n_mels < ((int64_t)1 << 29), "stft2048_mel_tri: clip too long (32-bit byte offsets inside a clip's mel block)");
  ContrastPlan cp;
  rc = parse_contrast_plan(contrast_out, cplan_host, cp);
  if (rc) return rc;
  if (stats_out) SYG_REQUIRE(sr > 0.f && roll_percent >= 0.f && roll_percent <= 1.f && bw_p > 0.f && (stats_mask & 31) != 0 &&
                                 stats_mask > 0 && stats_mask < 64, "stft2048_mel_tri: invalid statistics parameters");
  MelPlan plan;
  memset(&plan, 0, sizeof(plan));
  MfccArgs mf;
  memset(&mf, 0, sizeof(mf));
  mf.amin = 1e-10f; mf.top_db = -1.f;
  const bool extra = stats_out != nullptr || contrast_out != nullptr;
  if (extra)
    return launch<16, 9>(load_mode(), y, B, L, ldy, hop, center, T, window, twiddle, segtab, plan, n_mels, mel_out,
                         sr / (float)NFFT, roll_percent, bw_p, stats_mask, stats_out, cp, contrast_out, nullptr,
                         (hipStream_t)stream, mf);
  return launch<16, 8>(load_mode(), y, B, L, ldy, hop, center, T, window, twiddle, segtab, plan, n_mels, mel_out, 0.f, 0.f,
                       0.f, 0, nullptr, cp, nullptr, nullptr, (hipStream_t)stream, mf);
}

// MODE 7 without a filterbank: the per-frame statistics / contrast tail means alone (spectral_centroid / bandwidth /
// flatness / rolloff / contrast of manager.py:289-343 need no mel spectrogram) -- transform + row functions, nothing is
// projected, no clip epilogue; the waves only meet at the two stage hand-over barriers and run their row functions in two
// staggered halves (see MODE 7).  hop <= 512 (staged tiles); other hops: syg_stft2048_mel_f32.
extern "C" int syg_stft2048_stats_f32(const float* y, int64_t B, int64_t L, int64_t ldy, int hop, int center, int64_t T,
                                      const float* window, const float* twiddle, float sr, float roll_percent, float bw_p,
                                      int stats_mask, float* stats_out, const int32_t* cplan_host, float* contrast_out,
                                      void* stream) {
  SYG_REQUIRE(stats_out || contrast_out, "stft2048_stats: no statistics requested");
  int rc = check_common(y, B, L, ldy, hop, center, T, window, twiddle, 16);
  if (rc) return rc;
  SYG_REQUIRE(hop <= 512 && L < ((int64_t)1 << 28), "stft2048_stats: needs hop <= 512 (staged tiles); use syg_stft2048_mel_f32");
  SYG_REQUIRE(T < ((int64_t)1 << 24), "stft2048_stats: clip too long");
  ContrastPlan cp;
  rc = parse_contrast_plan(contrast_out, cplan_host, cp);
  if (rc) return rc;
  if (stats_out) SYG_REQUIRE(sr > 0.f && roll_percent >= 0.f && roll_percent <= 1.f && bw_p > 0.f && (stats_mask & 31) != 0 &&
                                 stats_mask > 0 && stats_mask < 64, "stft2048_stats: invalid statistics parameters");
  MelPlan plan;
  memset(&plan, 0, sizeof(plan));
  MfccArgs mf;
  memset(&mf, 0, sizeof(mf));
  mf.amin = 1e-10f; mf.top_db = -1.f;
  return launch<16, 7>(2, y, B, L, ldy, hop, center, T, window, twiddle, nullptr, plan, 0, nullptr, sr / (float)NFFT,
                       roll_percent, bw_p, stats_mask, stats_out, cp, contrast_out, nullptr, (hipStream_t)stream, mf);
}

// MODE 5: the statistics / contrast rows of syg_stft2048_mel_f32 AND the clip-resident MFCC of syg_stft2048_mfcc_f32
// from one launch (BASELINE config C4: extract_features(["mfcc", "spectral_centroid", "spectral_rolloff",
// "spectral_contrast"]), manager.py:289-371).  mfcc_out rows of clip b start at (b * mfcc_rows_per_clip) * T, so the
// MFCCs can be written straight into the head of a wider per-clip block.
extern "C" int syg_stft2048_features_f32(const float* y, int64_t B, int64_t L, int64_t ldy, int hop, int center, int64_t T,
                                         const float* window, const float* twiddle, const float* wpacked,
                                         const int32_t* plan_host, int n_mels, const float* dct, int n_mfcc,
                                         const float* lifter, float amin, float top_db, int ref_is_max, float ref_value,
                                         float sr, float roll_percent, float bw_p, int stats_mask, float* stats_out,
                                         const int32_t* cplan_host, float* contrast_out, float* mel_out, float* mfcc_out,
                                         int mfcc_rows_per_clip, void* stream) {
  SYG_REQUIRE(wpacked && plan_host && dct && mfcc_out, "stft2048_features: null pointer argument");
  SYG_REQUIRE(stats_out || contrast_out, "stft2048_features: no statistics requested (use syg_stft2048_mfcc_f32)");
  MelPlan plan;
  int rc = parse_mel_plan("stft2048_features", plan_host, n_mels, plan);
  if (rc) return rc;
  SYG_REQUIRE(plan_host[1] == 16, "stft2048_features: needs a 16-wave plan (got %d)", plan_host[1]);
  rc = check_common(y, B, L, ldy, hop, center, T, window, twiddle, 16);
  if (rc) return rc;
  SYG_REQUIRE(n_mfcc >= 1 && n_mfcc <= n_mels && mfcc_rows_per_clip >= n_mfcc, "stft2048_features: need 1 <= n_mfcc <= n_mels "
              "and mfcc_rows_per_clip >= n_mfcc");
  SYG_REQUIRE(amin >= 1.17549435e-38f, "stft2048_features: amin must be strictly positive (a normal float)");
  SYG_REQUIRE(ref_is_max == 0 || ref_is_max == 1, "stft2048_features: ref_is_max must be 0 or 1");
  SYG_REQUIRE(T < ((int64_t)1 << 24), "stft2048_features: clip too long");
  ContrastPlan cp;
  rc = parse_contrast_plan(contrast_out, cplan_host, cp);
  if (rc) return rc;
  if (stats_out) SYG_REQUIRE(sr > 0.f && roll_percent >= 0.f && roll_percent <= 1.f && bw_p > 0.f && (stats_mask & 31) != 0 &&
                                 stats_mask > 0 && stats_mask < 64, "stft2048_features: invalid statistics parameters");
  MfccArgs mf;
  mf.dct = dct; mf.lifter = lifter; mf.out = mfcc_out; mf.n_mfcc = n_mfcc; mf.ref_is_max = ref_is_max;
  mf.ref_value = ref_value; mf.amin = amin; mf.top_db = top_db; mf.tp = 0; mf.rows_per_clip = mfcc_rows_per_clip;
  return launch<16, 5>(load_mode(), y, B, L, ldy, hop, center, T, window, twiddle, wpacked, plan, n_mels, mel_out,
                       sr / (float)NFFT, roll_percent, bw_p, stats_mask, stats_out, cp, contrast_out, nullptr,
                       (hipStream_t)stream, mf);
}

extern "C" int syg_stft2048_c2c_f32(const float* y, int64_t B, int64_t L, int64_t ldy, int hop, int center,
                                    int64_t T, const float* window, const float* twiddle, float* out,
                                    void* stream) {
  int rc = check_common(y, B, L, ldy, hop, center, T, window, twiddle, 8);
  if (rc) return rc;
  SYG_REQUIRE(out, "stft2048_c2c: null output");
  MelPlan plan;
  ContrastPlan cp;
  memset(&plan, 0, sizeof(plan));
  memset(&cp, 0, sizeof(cp));
  return launch<8, 2>(1, y, B, L, ldy, hop, center, T, window, twiddle, nullptr, plan, 0, nullptr, 0.f, 0.f, 0.f, 0,
                      nullptr, cp, nullptr, out, (hipStream_t)stream);
}
