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

// Issue priority falls as a wave advances through its frame: the SIMD's arbiter otherwise favours the oldest
// wave, which then idles at barrier A while the youngest finishes alone with nothing to hide its LDS latency.
#define SETPRIO(n) __builtin_amdgcn_s_setprio(n)
// -DSYG_DEV=1: per-phase cycle stamps (tools/timeline.py; a development build -- syg_build_variant() != 0 -- whose
// statistics output carries the counters)
#include "stft_dev.h"

namespace syg {
namespace {

constexpr bool X2_MEL = true;    // mel-only modes keep 4 |X|^2 in the power rows (see stft2048_kernel)
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

#include "row_features.h"

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
// TWB: also fetch twb from the twiddle table (false: the caller keeps twb and re-makes the rest -- lane arithmetic only)
template <bool TWB = true>
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
    if (TWB) lc.twb[j] = twid[kb];
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
// PD: priority drop (MODE 1 keeps the top level for its row functions, the transform then runs one level lower)
// X2: return 2 X instead of X (the two halvings of the real split are left out; the caller's powers are then 4 |X|^2 --
// an exact scaling that the mel-only modes take back where the mel values leave the kernel: 32 instructions per frame)
template <int PD = 0, bool X2 = false>
__device__ __forceinline__ void wave_rfft2048(float2 (&v)[16], const LaneConst& lc, float2* __restrict__ sc,
                                             const float2* __restrict__ tw1l, const float2* __restrict__ tw2l,
                                             int lane, float2 (&xs)[2][4], float2 (&xm)[2][4], float2& x512 TARGS) {
  const int cl = lane >> 2, bp = lane & 3;
  // ---- pass 1: radix-16 over a (stride 64), twiddle W_1024^(b*c)
  dft16(v);
#pragma unroll
  for (int c = 1; c < 16; ++c) v[c] = cmul(v[c], tw1l[(c - 1) * 64 + lane]);
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
  }
  TICK(2, t[0].x);
  // ---- pass 2: lane = (c = lane>>2, b' = lane&3); radix-16 over a'
  SETPRIO(2 - PD > 0 ? 2 - PD : 0);
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
  }
  TICK(4, G[0][0].x);
  SETPRIO(1 - PD > 0 ? 1 - PD : 0);
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
__device__ __forceinline__ MfccArgs uni(const MfccArgs& a) {
  MfccArgs u;
  u.dct = uni(a.dct); u.lifter = uni(a.lifter); u.out = uni(a.out); u.n_mfcc = uni(a.n_mfcc);
  u.ref_is_max = uni(a.ref_is_max); u.ref_value = uni(a.ref_value); u.amin = uni(a.amin); u.top_db = uni(a.top_db);
  u.tp = uni(a.tp); u.rows_per_clip = uni(a.rows_per_clip);
  return u;
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
  // MODE 10 / 11: MODE 8 / 9 with a TWO-pass table (filterbanks of up to 128 pieces, e.g. 40 bands) -- the tile form of
  // MODE 6 / 7 without the clip-resident epilogue; with 8 waves two workgroups share a CU and drift out of phase
  constexpr bool TRIMEL = (MODE >= 8 && MODE <= 11);
  constexpr bool TRI = (MODE == 6 || MODE == 7 || TRIMEL);
  constexpr int NPASS = (MODE == 8 || MODE == 9) ? 4 : 2;
  constexpr int TRI_ROW_BASE = (NPASS == 4) ? TRI4_ROW_BASE : 0;
  typedef Lds<WAVES, !TRI, NPASS> LM;
  constexpr int SEG_WORDS = LM::SEG_WORDS;
  constexpr int NTHREADS = WAVES * 64;
  constexpr int TILE_T = WAVES;                                    // one frame per wave per tile
  constexpr bool COMPLEX_OUT = (MODE == 2);
  constexpr bool ROWFN = (MODE == 1 || MODE == 7 || MODE == 9 || MODE == 11);   // per-frame row functions (MODE 1: behind barrier B)
  constexpr bool CLIPM = (MODE == 3 || MODE == 6 || MODE == 7);      // the clip's mel matrix lives in LDS; epilogue at clip end
  // MODE 0 / 3 (mel only): the power rows hold 4 |X|^2 (wave_rfft2048<.., X2>); the factor is taken back -- exactly, a
  // power of two -- where mel values leave the kernel (MODE 0: at the store; MODE 3: the dB conversion works on 4 x mel
  // with 4 x amin and 4 x ref, the optional mel copy is scaled at its store).  MODE 1's statistics need the true powers.
  constexpr bool X2 = X2_MEL && (MODE == 0 || MODE == 3 || MODE == 6 || MODE == 8 || MODE == 10);
#ifdef SYG_DEV_NO_RELC
  constexpr bool RELC = false;                                       // (timeline builds: round 3's form, for the before / after table)
#else
  constexpr bool RELC = (MODE == 7);                                 // lane constants re-made per frame (see the tile loop)
#endif
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
  // MODE 3: [n_mels][mf.tp], red[WAVES], dct rows, lifter -- behind the fixed map; a launch that does not stage its tiles
  // (LOAD != 2) has no stage buffer, the clip's matrix starts there (128-band matrices only fit that way)
  float* clipmel = lds + (LOAD == 2 ? LM::TOTAL : LM::O_STAGE);
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
  const bool tri_defer = TRI && !TRIMEL && tri_idle > 0 && (tri_ndct + tri_idle - 1) / (tri_idle > 0 ? tri_idle : 1) <= 3;
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

#if SYG_DEV
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
    // tile.  The row functions therefore run at the top level and the transform one level lower (all equal: 428 against
    // 373-390 us per 1024 clips of C4; other splits no better).  MODE 0 / 3: projection, clip epilogue and slab combine at
    // the top level, the transform one level lower (150.3 against 151.0 us for the one-launch MFCC at C2).
    constexpr int PD = 1;
    SETPRIO(3 - PD);
    // The filterbank operands of this wave's slots are the same for every tile but cannot stay resident (the FFT needs
    // all 128 VGPRs): the first NPRE groups of four steps are re-fetched every tile, behind pass 3 of the transform.
    constexpr int NPRE = 7;            // unconditional: every wave's segment holds >= NPRE groups (zero padded); requested
                                       // in front of barrier A (behind pass 3 already: measured, no gain -- the barrier's
                                       // wait covers the latency either way)
    const int ng = plan.steps >> 2;
    const float4* wp4w = reinterpret_cast<const float4*>(wpacked) + (int64_t)w * ng * 64;
    float4 apre[NPRE];
    if (have) {
      // the lane id is laundered through an empty asm each iteration: the LDS / global addresses derived
      // from it are then recomputed per frame (a few integer ops) instead of being hoisted out of the tile
      // loop as ~100 loop-invariant registers that would spill
      int lv = lane;
      asm volatile("" : "+v"(lv));
      // MODE 7 (row functions AND the clip epilogue: the tightest register budget): the 16 integer / constant members of
      // the lane constants are re-made per frame from the laundered lane id (about 40 integer instructions) -- kept live
      // across the out-of-line row functions they pushed a register into scratch, and a scratch reload is a memory round
      // trip that nothing hides (it sat directly behind both call sites: the fixed cost of "any real row function")
      if (RELC) init_lane_const<false>(lc, lv, twid);
      apply_window(v, winl, lv);
      TICK(0, v[0].x);
      float2 xs[2][4], xm[2][4], x512;
      wave_rfft2048<PD, X2>(v, lc, sc, tw1l, tw2l, lv, xs, xm, x512 TPASS);
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
      SETPRIO(3);             // projection, clip epilogue and row functions at the top level
      const bool mine = (t < T);
      const bool clip_done = (t0 + TILE_T >= T);
      float* cmc = clipmel + cur * (n_mels * mf.tp);
      auto project = [&]() {
        int la = lane;
        asm volatile("" : "+v"(la)::"memory");
        wave_lds_sync();
        // (the table's band word was turned into the band's BYTE offset inside a mel matrix when the workgroup copied it)
        if (TRIMEL) {
          char* colg = reinterpret_cast<char*>(mel_out + (b * n_mels) * T + t);
          tri_project<NPASS>(prow - TRI_ROW_BASE, reinterpret_cast<const float4*>(cpl), la, tri_scan8,
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
#if SYG_DEV
      int tdep = lane;
      TICK(5, tdep);                   // (split + row store)
#endif
      if (mine) {
        if (tri_proj) project();
      } else if (tri_defer && pend_b >= 0) {
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
      // the same thing at the same time (all behind X2, all in front of X1 -- the first build, 832 us -- measured slower).
      // Waves w, w + 4, w + 8, w + 12 share a SIMD: two of them early, two late.
      float row_sres = 0.f;
      float2 row_pv = make_float2(0.f, 0.f);
      const bool row_early = (w >> 2) & 1;               // (8 waves: w, w + 4 share a SIMD -- one early, one late)
      auto row_compute = [&]() {
        if (stats_out != nullptr && contrast_out != nullptr) {
          const float3 f = row_features<NBIN, 0>((lds_row)prow, lane, binhz, roll_percent, bw_p, smask, (lds_iptr)cplc,
                                                     cplan.n_rows, cplan.ascending);
          row_sres = f.x; row_pv = make_float2(f.y, f.z);
        } else {
          if (stats_out != nullptr) row_sres = row_stats<NBIN, 0>((lds_row)prow, lane, binhz, roll_percent, bw_p, smask);
          if (contrast_out != nullptr)
            row_pv = row_contrast_all<0>((lds_row)prow, lane, (lds_iptr)cplc, cplan.n_rows, cplan.ascending);
        }
      };
      TICK(6, tdep);                   // projection (or the deferred clip epilogue)
      if (ROWFN && mine && row_early) {
        // (the three result registers wait in the wave's own row, dead until its next transform: live across the fetch of
        // the next frame they cost three spilled registers)
        row_compute();
        prow[lane] = row_sres; prow[64 + lane] = row_pv.x; prow[128 + lane] = row_pv.y;
      }
      if (CLIPM && clip_done) publish_max();
      TICK(7, tdep);                   // row functions of the early half
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();                                  // X1: the clip's columns of this tile are written too
      TICK(8, tdep);                   // wait at X1
      if (LOAD == 2) {
        have = false;
        if (tile + 1 < tile_end) fetch(tile + 1);
        __syncthreads();                                // X2: every wave holds its next frame
      }
      TICK(9, tdep);                   // fetch of the next frame + wait at X2
      if (CLIPM && clip_done) {
        // (clip_dct's entry waits for outstanding memory operations, so the refill is issued behind it)
        if (tri_defer) pend_b = mf.n_mfcc > 0 ? (int64_t)b : pend_b;
        else if (w < tri_ndct)
          clip_dct<WAVES>((int)(uintptr_t)(lds_fptr)cmc, (int)(uintptr_t)(lds_fptr)(tri_red + cur * WAVES),
                          (int)(uintptr_t)(lds_fptr)tri_dct, mf, n_mels, (int)T, (int)b, w, lane);
        cur ^= 1;
      }
      if (ROWFN && mine) {
        if (!row_early) row_compute();
        else { row_sres = prow[lane]; row_pv = make_float2(prow[64 + lane], prow[128 + lane]); }
        TICK(10, tdep);                // row functions of the late half (+ the clip epilogue where it is not deferred)
        if (contrast_out != nullptr) {
          if (lane < cplan.n_rows) {
            contrast_out[((b * 2 + 0) * cplan.n_rows + lane) * T + t] = row_pv.x;
            contrast_out[((b * 2 + 1) * cplan.n_rows + lane) * T + t] = row_pv.y;
          }
        }
#if SYG_DEV
        if (stats_out != nullptr && row_sres == 12345.678f)      // (timeline build: stats_out carries the phase counters)
#else
        if (stats_out != nullptr && lane < SYG_NSTAT && ((stats_row_mask(smask) >> lane) & 1))
#endif
          stats_out[(b * SYG_NSTAT + lane) * T + t] = row_sres;
      }
      if (LOAD == 2 && tile + 2 < tile_end) dma(tile + 2);
      TICK(11, tdep);                  // the tile's stores + the stage refill
      SETPRIO(0);
      continue;
    }
#if SYG_DEV
    int tdep = lane;
    TICK(5, tdep);
#endif
    int la = lane;                     // laundered: per-lane addresses are recomputed here, not kept across the FFT
    asm volatile("" : "+v"(la)::"memory");
    const float4* wp4 = wp4w + la;
#pragma unroll
    for (int q = 0; q < NPRE; ++q) apre[q] = wp4[q * 64];
    const int pslot = mtab[w * 4 + (la >> 4)];          // row POSITION (skewed, see ppos()) of this lane's slot
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (staged mode: the next tile's samples have landed too)
    __syncthreads();                                    // barrier A: rows complete
    TICK(6, tdep);
    if (!ROWFN) SETPRIO(3);                             // projection (+ clip epilogue) at the top level
    if (CLIPM && pend_b >= 0) {
      if (w < ((mf.n_mfcc + 15) >> 4) * (mf.tp >> 4))    // output tile w is formed by wave w
        clip_dct<WAVES>((int)(uintptr_t)(lds_fptr)clipmel, (int)(uintptr_t)(lds_fptr)(clipmel + n_mels * mf.tp),
                        (int)(uintptr_t)(lds_fptr)(clipmel + n_mels * mf.tp + WAVES), mf, n_mels, (int)T, (int)pend_b, w, lane);
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
      // group g is combined by wave (g + 6) mod WAVES.  Ten groups at 40 mels: with shift 0 the ten OLDEST waves carry
      // the combine and the six youngest -- which the arbiter already serves last -- none; shift 6 gives it to the
      // waves 6 .. 15 (148.5 vs 150.8 us for the one-launch MFCC at C2; 3: 149.1, 10: slower than 6)
      constexpr int REDSHIFT = 6;
      for (int g = (w + WAVES - REDSHIFT) % WAVES; g < plan.n_groups; g += WAVES) {
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
            if (!SYG_DEV && mel_out != nullptr && mel < n_mels && t0 + tt < T) mel_out[(b * n_mels + mel) * T + t0 + tt] = MELSC * sum;
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
      SETPRIO(3);
      if (t < T) {
        // statistics first (a wide contrast band parks its lists in the row's low words); every result waits in a lane
        // of a register and is stored behind the last call
        float sres = 0.f, pk = 0.f, vl = 0.f;
        const bool both = stats_out != nullptr && contrast_out != nullptr;
        if (both) {
          const float3 f = row_features<NBIN, 0>((lds_row)prow, lane, binhz, roll_percent, bw_p, smask, (lds_iptr)cplc,
                                                     cplan.n_rows, cplan.ascending);
          sres = f.x; pk = f.y; vl = f.z;
        } else if (stats_out != nullptr) sres = row_stats<NBIN, 0>((lds_row)prow, lane, binhz, roll_percent, bw_p, smask);
        if (contrast_out != nullptr) {
          if (!both) {
            const float2 pv = row_contrast_all<0>((lds_row)prow, lane, (lds_iptr)cplc, cplan.n_rows, cplan.ascending);
            pk = pv.x; vl = pv.y;
          }
          if (lane < cplan.n_rows) {
            contrast_out[((b * 2 + 0) * cplan.n_rows + lane) * T + t] = pk;
            contrast_out[((b * 2 + 1) * cplan.n_rows + lane) * T + t] = vl;
          }
        }
#if SYG_DEV
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
      SETPRIO(0);
      TICK(11, tdep);
      combine();
      if (clip_done) publish();
      if (FETCH_EARLY && tile + 2 < tile_end) dma(tile + 2);
    }
  }
  if (TRIMEL) {
    // (no clip epilogue: the mel columns are in HBM)
  } else if (TRI) {
    // (deferred epilogue: the last clip's matrix is complete behind X1 of its last tile)
    if (pend_b >= 0 && w < tri_ndct)
      clip_dct<WAVES>((int)(uintptr_t)(lds_fptr)(clipmel + (cur ^ 1) * (n_mels * mf.tp)),
                      (int)(uintptr_t)(lds_fptr)(tri_red + (cur ^ 1) * WAVES), (int)(uintptr_t)(lds_fptr)tri_dct, mf, n_mels,
                      (int)T, (int)pend_b, w, lane);
  } else if (CLIPM && pend_b >= 0) {
    __syncthreads();
    if (w < ((mf.n_mfcc + 15) >> 4) * (mf.tp >> 4))
      clip_dct<WAVES>((int)(uintptr_t)(lds_fptr)clipmel, (int)(uintptr_t)(lds_fptr)(clipmel + n_mels * mf.tp),
                      (int)(uintptr_t)(lds_fptr)(clipmel + n_mels * mf.tp + WAVES), mf, n_mels, (int)T, (int)pend_b, w, lane);
  }
#if SYG_DEV
  if (MODE == 3 && lane < 12)
    mel_out[((int64_t)blockIdx.x * WAVES + w) * 16 + lane] = (float)tacc[lane] / (float)(tile_end - tile_begin);
  if (ROWFN && stats_out != nullptr && lane < 12)
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
  constexpr bool TRIMEL = (MODE >= 8 && MODE <= 11);     // tile form of the segment-sum projection
  constexpr int TRIMEL_NPASS = (MODE == 8 || MODE == 9) ? 4 : 2;
  constexpr bool TRI = (MODE == 6 || MODE == 7);
  size_t lds = TRIMEL ? lds_bytes<WAVES, false, TRIMEL_NPASS>() : lds_bytes<WAVES, !TRI>();
  size_t clip_extra = 0;
  if (TRIMEL) mf.tp = tiles * WAVES;
  if (MODE == 3 || TRI) {
    // whole clips per workgroup; the clip's mel matrix [n_mels][tiles * WAVES] sits behind the fixed LDS map
    int cw = 0, cper = 0;
    persistent_grid(B, WAVES, cw, cper);
    per = cper * tiles;
    wgs = cw;
    mf.tp = tiles * WAVES;
    if (X2_MEL && (MODE == 3 || MODE == 6)) { mf.amin *= 4.f; mf.ref_value *= 4.f; }     // the clip's mel matrix holds 4 x mel (exact scaling)
    clip_extra = ((size_t)(TRI ? 2 : 1) * ((size_t)n_mels * mf.tp + WAVES) + (size_t)mf.n_mfcc * (n_mels + 1)) * sizeof(float);
  }
  // staged tiles (LDS-DMA) need the tile's sample run to fit the stage buffer and 32-bit byte offsets
  bool can_stage = (MODE != 2) && hop <= 512 && L < ((int64_t)1 << 28);
  // a clip matrix that does not fit behind the stage buffer takes the buffer's place: frames straight from global memory
  constexpr size_t STAGE_BYTES = (size_t)Lds<WAVES, !TRI>::STAGE_FLOATS * sizeof(float);
  if ((MODE == 3 || TRI) && !TRIMEL && lds + clip_extra > LDS_LIMIT) can_stage = false;
  if (load == 2 && !can_stage) load = 1;
  const bool vec2 = (hop % 2 == 0) && (ldy % 2 == 0) && (((uintptr_t)y) % 8 == 0);
  if (load == 1 && !vec2) load = 0;
  if ((MODE == 3 || TRI) && !TRIMEL) {
    lds = lds - (load != 2 ? STAGE_BYTES : 0) + clip_extra;
    SYG_REQUIRE(lds <= LDS_LIMIT, "stft2048_mfcc: the clip's mel matrix (%d x %d) does not fit the LDS left over (%zu B > %zu B); "
                "use syg_stft2048_mel_f32 + syg_logmel_dct_f32", n_mels, mf.tp, lds, LDS_LIMIT);
  }
  const int dma_wide = (hop % 4 == 0) && (pad % 4 == 0) && (ldy % 4 == 0) && (L % 4 == 0) && (((uintptr_t)y) % 16 == 0);
  auto kern = load == 2 ? stft2048_kernel<WAVES, 2, MODE>
                        : load == 1 ? stft2048_kernel<WAVES, 1, MODE> : stft2048_kernel<WAVES, 0, MODE>;
  {
    // set at every launch: the attribute belongs to the (function, device) pair, and a per-process "already set"
    // flag would leave a second device without it
    const size_t cap = (MODE == 3 || TRI) ? LDS_LIMIT : TRIMEL ? lds_bytes<WAVES, false, TRIMEL_NPASS>() : lds_bytes<WAVES>();
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
  // 2: beside the stage buffer; 1: in the stage buffer's place (the launch then loads its frames straight from global
  // memory: 128 bands x 94 frames; measured 186 against 196 us for the two-launch form there, 176 against 166 at 64 bands)
  const int64_t extra = ((int64_t)n_mels * tp + 16 + (int64_t)n_mfcc * (n_mels + 1)) * 4;
  if ((int64_t)lds_bytes<16>() + extra <= (int64_t)LDS_LIMIT) return 2;
  return (int64_t)lds_bytes<16>() - (int64_t)Lds<16>::STAGE_FLOATS * 4 + extra <= (int64_t)LDS_LIMIT ? 1 : 0;
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

// MODE 7: MFCC rows + statistics rows + contrast tail means from ONE launch, with the
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

// MODE 8 ... 11: the TILE form of the segment-sum projection -- samples in, mel POWER out [B, n_mels, T] (no weight matrix,
// no projection barriers, every frame's column written by the wave that transformed it); syg_logmel_dct_f32 /
// syg_feature_block_f32 is the second launch of an MFCC.  Optional statistics / contrast rows from the same launch (the row
// functions of syg_stft2048_features_tri_f32).  Tiles are shared out evenly over the workgroups, so one long clip
// (BASELINE config C1) fills the chip.
//   segtab   n_segtab = 2048: pack_mel_segments(..., n_pass=4, row_base=4), [4][2][64][4] words -- up to 256 pieces: the
//            reference's default filterbank of 128 bands (manager.py:214), 64 ... 200 bands at the usual sample rates;
//            n_segtab = 1024: the two-pass table of syg_stft2048_mfcc_tri_f32 (up to 128 pieces, e.g. 40 bands)
//   waves    16 (one workgroup per CU) or 8 (two per CU, each with its own tile barriers: they drift out of phase, so that
//            one's transforms run beside the other's LDS exchanges)
extern "C" int syg_stft2048_mel_tri_f32(const float* y, int64_t B, int64_t L, int64_t ldy, int hop, int center, int64_t T,
                                        const float* window, const float* twiddle, const float* segtab, int n_segtab,
                                        int n_mels, float* mel_out, float sr, float roll_percent, float bw_p, int stats_mask,
                                        float* stats_out, const int32_t* cplan_host, float* contrast_out, int waves,
                                        void* stream) {
  SYG_REQUIRE(segtab && mel_out, "stft2048_mel_tri: null pointer argument");
  SYG_REQUIRE(n_segtab == SEGTAB4_WORDS || n_segtab == SEGTAB_WORDS, "stft2048_mel_tri: the piece table has %d words, this "
              "library reads %d (four passes, row_base 4) or %d (two passes) -- sygnals_amd._tables.pack_mel_segments", n_segtab,
              SEGTAB4_WORDS, SEGTAB_WORDS);
  SYG_REQUIRE(((uintptr_t)segtab) % 16 == 0, "stft2048_mel_tri: the piece table must be 16-byte aligned");
  SYG_REQUIRE(waves == 8 || waves == 16, "stft2048_mel_tri: waves must be 8 or 16 (got %d)", waves);
  int rc = check_common(y, B, L, ldy, hop, center, T, window, twiddle, waves);
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
  const bool four = n_segtab == SEGTAB4_WORDS;
  const float binhz = extra ? sr / (float)NFFT : 0.f;
  if (!extra) { roll_percent = 0.f; bw_p = 0.f; stats_mask = 0; }
#define SYG_LAUNCH(W, M)                                                                                               \
  launch<W, M>(load_mode(), y, B, L, ldy, hop, center, T, window, twiddle, segtab, plan, n_mels, mel_out, binhz, roll_percent, \
               bw_p, stats_mask, stats_out, cp, contrast_out, nullptr, (hipStream_t)stream, mf)
  if (waves == 16) {
    if (four) return extra ? SYG_LAUNCH(16, 9) : SYG_LAUNCH(16, 8);
    return extra ? SYG_LAUNCH(16, 11) : SYG_LAUNCH(16, 10);
  }
  if (four) return extra ? SYG_LAUNCH(8, 9) : SYG_LAUNCH(8, 8);
  return extra ? SYG_LAUNCH(8, 11) : SYG_LAUNCH(8, 10);
#undef SYG_LAUNCH
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
