// Zero-phase SOS filtering (scipy.signal.sosfiltfilt semantics) for a batch of clips.
//
// Reference: apply_sos_filter, sygnals/core/filters.py:85-115 -> scipy.signal.sosfiltfilt(sos, x)
// with padtype='odd', padlen = 3*ntaps: odd-extend both ends, run the direct-form-II-transposed
// cascade forward from the steady-state initial condition zi*ext[0], reverse, run it again from
// zi*y_rev[0], reverse, trim.
//
// The recurrence is serial in time, so time is cut into chunks of CS samples and the cascade is
// treated as one linear system of dimension D = 2*n_sections:
//   pass A  every chunk runs from a zero state          -> its zero-state end state
//   pass B  a short serial scan per clip, s <- A^CS s + zs (A^n formed on the host in float64)
//           gives every chunk its true initial state
//   pass C  every chunk re-runs from the true state and writes its output (reversed, so the
//           backward sweep reads forward again)
// Recurrences run in float64 (the pole radius of the headline band-pass is 0.987; float32 state
// alone costs half the 1e-5 parity budget); signals are stored float32.
//
// Data movement (it, not the fp64 arithmetic, bounds these passes).  Each sweep works on an ALIGNED GRID: grid
// position g = sequence index + skip, with skip chosen per sweep so that every 4-sample group of the grid is
// a 16-byte aligned float4 -- and every 8-lane, 32-sample segment a whole 128-byte line -- both where it is read and
// where it is written:
//   forward sweep   reads the clip itself (odd extension computed on the fly): skip_f = -pad mod 32 aligns x;
//                   writes its output reversed into the work buffer G at grid K - g (K fixed by skip_b);
//   backward sweep  reads G in place order: skip_b = -(lext + skip_f) mod 256 makes the forward stores aligned and
//                   lets forward chunk c and backward chunk n - 1 - c cover the same samples.
// The skip leading positions of chunk 0 are not samples: its lane leaves the state untouched there.  Chunk 0 is
// therefore short, so its pass A starts from the true initial state zi * first sample (known up front) and what
// it reports is already the true state at the start of chunk 1 -- the scan takes it as is.  A wave owns 64 consecutive chunks of one clip and moves 64 x 32-sample
// tiles through LDS (lane-major rows of 36 floats: 128-bit, conflict-free LDS accesses on both sides) so that
// global accesses stay coalesced while each lane walks its own chunk; the next tile's loads are issued before the
// recurrence over the current one.  One wave per workgroup: ordering inside the wave replaces barriers.
#include "common.h"
#include "sosfilt_clip.h"
#include <string.h>
#include <stdlib.h>

namespace syg {
namespace {

constexpr int CS = 256;      // samples per chunk
constexpr int TS = 32;       // samples per LDS tile row group
constexpr int MAXS = 8;      // sections
constexpr int MAXD = 2 * MAXS;
constexpr int LSTR = 36;     // LDS tile row stride (floats): one row = the 32 samples of one chunk (+4 pad)

struct SosParams {
  double b0[MAXS], b1[MAXS], b2[MAXS], a1[MAXS], a2[MAXS];
  double zi[MAXD];
  double apow[MAXD * MAXD];  // A^CS, row-major D x D (padded to MAXD)
};

// Sample i of the odd-extended, zero-tailed signal (scipy's padtype='odd'): read straight from the clip, so the
// forward sweep needs no extended copy in HBM.
__device__ __forceinline__ float ext_at(const float* __restrict__ xb, int64_t L, int pad, int64_t lext, int64_t i) {
  if (i < 0) return 0.f;                                  // (grid positions in front of the sequence)
  if (i < pad) return 2.f * xb[0] - xb[pad - i];
  if (i < pad + L) return xb[i - pad];
  if (i < lext) return 2.f * xb[L - 1] - xb[L - 2 - (i - pad - L)];
  return 0.f;
}
typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));   // 16-byte load at 4-byte alignment

// ---- pass A / C
// MODE 0: zero-state pass, writes end states.  MODE 1: true pass of the forward sweep, output reversed into the
// work buffer (grid K - g).  MODE 2: true pass of the backward sweep, output reversed and trimmed into y[b, n].
// SRCX: the input is the clip itself, odd-extended on the fly (forward sweep; `in` = x, row stride `ldin`);
// otherwise the [B, lpad] work buffer (backward sweep).  Grid position g <-> sequence index i = g - skip.
template <int S, int MODE, bool SRCX>
__global__ __launch_bounds__(64) void chunk_kernel(const float* __restrict__ in, int64_t ldin, int nch, int64_t lext,
                                                   int skip, SosParams P, const double* __restrict__ init,
                                                   double* __restrict__ zs, float* __restrict__ dst, int64_t lddst,
                                                   int64_t K, int pad, int64_t L) {
  __shared__ __attribute__((aligned(16))) float tile[64 * LSTR];
  const int lane = threadIdx.x;
  const int64_t b = blockIdx.y;
  const int c0 = blockIdx.x * 64;
  const int c = c0 + lane;
  const float* inb = in + b * ldin;
  double z0[S], z1[S];
#pragma unroll
  for (int s = 0; s < S; ++s) { z0[s] = 0.0; z1[s] = 0.0; }
  if (MODE != 0 && c < nch) {
    const double* ip = init + ((int64_t)b * nch + c) * (2 * S);
#pragma unroll
    for (int s = 0; s < S; ++s) { z0[s] = ip[2 * s]; z1[s] = ip[2 * s + 1]; }
  }
  if (MODE == 0 && c == 0) {
    // chunk 0 runs from the true initial state zi * (sequence sample 0): see the header
    const double first = (double)(SRCX ? ext_at(inb, L, pad, lext, 0) : inb[skip]);
#pragma unroll
    for (int s = 0; s < S; ++s) { z0[s] = P.zi[2 * s] * first; z1[s] = P.zi[2 * s + 1] * first; }
  }
  // cooperative load of one 64-chunk x 32-sample tile: 8 lanes x float4 cover one chunk's 32 samples
  auto load_tile = [&](int j0, float4 (&v)[8]) {
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const int ch = r * 8 + (lane >> 3), part = lane & 7;
      v[r] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (c0 + ch < nch) {
        const int64_t g0 = (int64_t)(c0 + ch) * CS + j0 + 4 * part;
        if (!SRCX) {
          v[r] = *reinterpret_cast<const float4*>(inb + g0);                 // work buffer: aligned by construction
        } else {
          const int64_t i0 = g0 - skip;                                      // sequence index; x index = i0 - pad
          if (i0 >= pad && i0 + 3 < pad + L) {
            v[r] = *reinterpret_cast<const float4*>(inb + (i0 - pad));       // 16-byte aligned: skip = -pad mod 4
          } else {
            v[r] = make_float4(ext_at(inb, L, pad, lext, i0), ext_at(inb, L, pad, lext, i0 + 1),
                               ext_at(inb, L, pad, lext, i0 + 2), ext_at(inb, L, pad, lext, i0 + 3));
          }
        }
      }
    }
  };
  auto step = [&](double u) {
#pragma unroll
    for (int s = 0; s < S; ++s) {
      const double yv = fma(P.b0[s], u, z0[s]);
      z0[s] = fma(P.b1[s], u, fma(-P.a1[s], yv, z1[s]));
      z1[s] = fma(P.b2[s], u, -P.a2[s] * yv);
      u = yv;
    }
    return u;
  };
  float4 nxt[8], nx2[8];                 // two tiles in flight: the passes are bound by memory latency, not arithmetic
  load_tile(0, nxt);
  load_tile(TS, nx2);
  for (int j0 = 0; j0 < CS; j0 += TS) {
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const int ch = r * 8 + (lane >> 3), part = lane & 7;
      *reinterpret_cast<float4*>(&tile[ch * LSTR + 4 * part]) = nxt[r];
    }
#pragma unroll
    for (int r = 0; r < 8; ++r) nxt[r] = nx2[r];
    if (j0 + 2 * TS < CS) load_tile(j0 + 2 * TS, nx2);
    wave_lds_sync();
    float* row = &tile[lane * LSTR];
    int jb = 0;
    // the wave that holds chunk 0: its first `skip` (< CS) grid positions are not samples.  A tile that lies wholly
    // in front of the first sample runs as usual and the chunk-0 lane takes its state back afterwards; the one tile
    // the first sample falls into is walked sample by sample with that lane sitting the leading positions out
    const bool dead_tile = (c0 == 0) && (j0 + TS <= skip);
    double s0[S], s1[S];
    if (dead_tile) {
#pragma unroll
      for (int s = 0; s < S; ++s) { s0[s] = z0[s]; s1[s] = z1[s]; }
    }
    if (c0 == 0 && j0 < skip && !dead_tile) {
      for (int j = 0; j < TS; ++j) {
        float o = 0.f;
        if (c != 0 || j0 + j >= skip) o = (float)step((double)row[j]);
        if (MODE != 0) row[j] = o;
      }
      jb = TS;
    }
#pragma unroll 2
    for (int j = jb; j < TS; j += 4) {
      const float4 q = *reinterpret_cast<const float4*>(row + j);
      const float o0 = (float)step((double)q.x), o1 = (float)step((double)q.y), o2 = (float)step((double)q.z),
                  o3 = (float)step((double)q.w);
      if (MODE != 0) *reinterpret_cast<float4*>(row + j) = make_float4(o0, o1, o2, o3);
    }
    if (dead_tile && c == 0) {
#pragma unroll
      for (int s = 0; s < S; ++s) { z0[s] = s0[s]; z1[s] = s1[s]; }
    }
    if (MODE != 0) {
      wave_lds_sync();
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        const int ch = r * 8 + (lane >> 3), part = lane & 7;
        if (c0 + ch < nch) {
          const float4 v = *reinterpret_cast<const float4*>(&tile[ch * LSTR + 4 * part]);
          const int64_t g0 = (int64_t)(c0 + ch) * CS + j0 + 4 * part;
          const int64_t i0 = g0 - skip;                       // sequence index of v.x
          if (MODE == 1) {
            // reversed into the work buffer: sample g -> grid K - g; (K - g0 - 3) is a multiple of 4
            float* d = dst + b * lddst + (K - g0 - 3);
            if (i0 >= 0 && i0 + 3 < lext) {
              *reinterpret_cast<float4*>(d) = make_float4(v.w, v.z, v.y, v.x);
            } else {
              if (i0 + 3 >= 0 && i0 + 3 < lext) d[0] = v.w;
              if (i0 + 2 >= 0 && i0 + 2 < lext) d[1] = v.z;
              if (i0 + 1 >= 0 && i0 + 1 < lext) d[2] = v.y;
              if (i0 >= 0 && i0 < lext) d[3] = v.x;
            }
          } else {
            // y[n], n = lext - 1 - pad - i: alignment of the result is the caller's, so dword stores
            const int64_t n0 = lext - 1 - pad - i0;
            float* d = dst + b * lddst;
            if (i0 >= 0 && n0 - 3 >= 0 && n0 < L) {          // interior: one 16-byte store at 4-byte alignment
              f4u o; o.x = v.w; o.y = v.z; o.z = v.y; o.w = v.x;
              *reinterpret_cast<f4u*>(d + (n0 - 3)) = o;
              continue;
            }
            if (i0 >= 0 && n0 >= 0 && n0 < L) d[n0] = v.x;
            if (i0 + 1 >= 0 && n0 - 1 >= 0 && n0 - 1 < L) d[n0 - 1] = v.y;
            if (i0 + 2 >= 0 && n0 - 2 >= 0 && n0 - 2 < L) d[n0 - 2] = v.z;
            if (i0 + 3 >= 0 && n0 - 3 >= 0 && n0 - 3 < L) d[n0 - 3] = v.w;
          }
        }
      }
    }
    wave_lds_sync();
  }
  if (MODE == 0 && c < nch) {
    double* zp = zs + ((int64_t)b * nch + c) * (2 * S);
#pragma unroll
    for (int s = 0; s < S; ++s) { zp[2 * s] = z0[s]; zp[2 * s + 1] = z1[s]; }
  }
}

// ---- pass B: 16 lanes per clip (one DPP row); lane r carries state component r.  The state vector is broadcast
// inside the row with DPP row_newbcast (two 32-bit moves per double) instead of ds_bpermute, and the zero-state
// end states of the next chunks are loaded four steps ahead of the serial chain.
template <int J>
__device__ __forceinline__ double row_bcast(double v) {
  const uint64_t u = (uint64_t)__double_as_longlong(v);
  const int lo = __builtin_amdgcn_update_dpp(0, (int)(uint32_t)u, 0x150 + J, 0xF, 0xF, false);
  const int hi = __builtin_amdgcn_update_dpp(0, (int)(uint32_t)(u >> 32), 0x150 + J, 0xF, 0xF, false);
  return __longlong_as_double((long long)(((uint64_t)(uint32_t)hi << 32) | (uint32_t)lo));
}
template <int D, int J>
__device__ __forceinline__ void bcast_all(double (&b)[D], double s) {
  if constexpr (J < D) {
    b[J] = row_bcast<J>(s);
    bcast_all<D, J + 1>(b, s);
  }
}

// row . state with four independent partial sums (the serial scan is latency-bound: one accumulator would chain
// D dependent fp64 FMAs per step); fixed summation order, so results are reproducible
template <int D>
__device__ __forceinline__ double row_dot(const double (&arow)[D], double s, double z) {
  double b[D];
  bcast_all<D, 0>(b, s);
  double p0 = z, p1 = 0.0, p2 = 0.0, p3 = 0.0;
#pragma unroll
  for (int j = 0; j < D; j += 4) {
    p0 = fma(arow[j], b[j], p0);
    if (j + 1 < D) p1 = fma(arow[j + 1], b[j + 1], p1);
    if (j + 2 < D) p2 = fma(arow[j + 2], b[j + 2], p2);
    if (j + 3 < D) p3 = fma(arow[j + 3], b[j + 3], p3);
  }
  return (p0 + p1) + (p2 + p3);
}

template <int S, bool SRCX>
__global__ __launch_bounds__(256) void scan_kernel(const float* __restrict__ in, int64_t ldin, int nch, int64_t B,
                                                   int skip, SosParams P, const double* __restrict__ zs,
                                                   double* __restrict__ init, int pad, int64_t L, int64_t lext) {
  constexpr int D = 2 * S;
  const int r = threadIdx.x & 15;
  const int64_t b = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4);
  const bool live = (b < B) && (r < D);
  const int64_t bb = (b < B) ? b : (B - 1);
  double arow[D];
#pragma unroll
  for (int j = 0; j < D; ++j) arow[j] = 0.0;
#pragma unroll
  for (int rr = 0; rr < D; ++rr)
    if (r == rr) {
#pragma unroll
      for (int j = 0; j < D; ++j) arow[j] = P.apow[rr * MAXD + j];
    }
  double zi = 0.0;
#pragma unroll
  for (int rr = 0; rr < D; ++rr)
    if (r == rr) zi = P.zi[rr];
  const float first = SRCX ? ext_at(in + bb * ldin, L, pad, lext, 0) : in[bb * ldin + skip];   // sequence sample 0
  double s = zi * (double)first;
  const int rc = (r < D) ? r : 0;
  const double* zp = zs + (int64_t)bb * nch * D + rc;
  double* ip = init + (int64_t)bb * nch * D + rc;
  constexpr int AHEAD = 32;       // chunk states in flight: a step is ~150 cycles, memory latency an order more
  double zq[AHEAD];
#pragma unroll
  for (int k = 0; k < AHEAD; ++k) zq[k] = (live && k < nch) ? zp[(int64_t)k * D] : 0.0;
  for (int c0 = 0; c0 < nch; c0 += AHEAD) {
    double zn[AHEAD];
#pragma unroll
    for (int k = 0; k < AHEAD; ++k) zn[k] = (live && c0 + AHEAD + k < nch) ? zp[(int64_t)(c0 + AHEAD + k) * D] : 0.0;
#pragma unroll
    for (int k = 0; k < AHEAD; ++k) {
      if (c0 + k < nch) {
        if (live) ip[(int64_t)(c0 + k) * D] = s;
        // chunk 0 (short by `skip`) ran pass A from the true state: its report is the state at chunk 1
        s = (c0 + k == 0) ? zq[k] : row_dot<D>(arow, s, zq[k]);
      }
    }
#pragma unroll
    for (int k = 0; k < AHEAD; ++k) zq[k] = zn[k];
  }
}

void host_step(const double* sos5, int S, double* z, double x) {
  double u = x;
  for (int s = 0; s < S; ++s) {
    const double* c = sos5 + 5 * s;  // b0 b1 b2 a1 a2
    const double y = c[0] * u + z[2 * s];
    z[2 * s] = c[1] * u - c[3] * y + z[2 * s + 1];
    z[2 * s + 1] = c[2] * u - c[4] * y;
    u = y;
  }
}

// A^n (row-major D x D in a MAXD-strided array) by repeated multiplication; n <= CS
void mat_pow(const double* A1, int D, int n, double* out) {     // square and multiply (n <= 256: at most 16 products)
  double R[MAXD * MAXD] = {0}, Bm[MAXD * MAXD] = {0}, T[MAXD * MAXD];
  auto mul = [&](const double* X, const double* Y, double* Z) {  // Z = X Y (Z may alias X or Y)
    for (int i = 0; i < D; ++i)
      for (int j = 0; j < D; ++j) {
        double acc = 0.0;
        for (int k = 0; k < D; ++k) acc += X[i * MAXD + k] * Y[k * MAXD + j];
        T[i * MAXD + j] = acc;
      }
    for (int i = 0; i < D; ++i)
      for (int j = 0; j < D; ++j) Z[i * MAXD + j] = T[i * MAXD + j];
  };
  for (int i = 0; i < D; ++i) R[i * MAXD + i] = 1.0;
  for (int i = 0; i < D; ++i)
    for (int j = 0; j < D; ++j) Bm[i * MAXD + j] = A1[i * MAXD + j];
  for (int e = n; e > 0; e >>= 1) {
    if (e & 1) mul(R, Bm, R);
    if (e > 1) mul(Bm, Bm, Bm);
  }
  for (int i = 0; i < MAXD * MAXD; ++i) out[i] = R[i];
}

int64_t grid_chunks(int64_t lext, int skip) { return (lext + skip + CS - 1) / CS; }

template <int S>
int launch_all(const float* x, int64_t B, int64_t L, int64_t ldx, const SosParams& P, int pad, float* y,
               int64_t ldy, void* work, hipStream_t st) {
  const int64_t lext = L + 2 * (int64_t)pad;
  // 32-sample (128-byte) alignment of every 8-lane segment: x index of grid g = g - skip_f - pad = 0 (mod 32);
  // forward stores K - g - 31 .. K - g land on aligned 128-byte segments of G
  const int skip_f = (TS - pad % TS) % TS;
  // skip_b makes K + 1 a multiple of the chunk length: forward chunk c and backward chunk nch_f - 1 - c then cover
  // the same samples (and every 128-byte segment stays aligned, CS being a multiple of TS)
  const int skip_b = (int)((CS - (lext + skip_f) % CS) % CS);
  const int64_t K = lext - 1 + skip_f + skip_b;                  // forward grid g  <->  backward grid K - g
  const int nch_f = (int)grid_chunks(lext, skip_f), nch_b = (int)grid_chunks(lext, skip_b);
  const int nch_w = (int)grid_chunks(lext, CS - 1);              // what syg_sosfiltfilt_work_bytes sized the buffers for
  const int64_t lpad = (int64_t)nch_w * CS;
  float* G = (float*)work;                                       // forward output, reversed: [B, lpad]
  double* zs = (double*)(G + B * lpad);
  double* init = zs + B * (int64_t)nch_w * (2 * S);
  dim3 gsc((unsigned)((B + 15) / 16));
  // forward sweep: reads the clip (odd extension on the fly), writes G
  dim3 gf((unsigned)((nch_f + 63) / 64), (unsigned)B);
  hipLaunchKernelGGL((chunk_kernel<S, 0, true>), gf, dim3(64), 0, st, x, ldx, nch_f, lext, skip_f, P,
                     (const double*)nullptr, zs, (float*)nullptr, (int64_t)0, K, pad, L);
  hipLaunchKernelGGL((scan_kernel<S, true>), gsc, dim3(256), 0, st, x, ldx, nch_f, B, skip_f, P, (const double*)zs,
                     init, pad, L, lext);
  hipLaunchKernelGGL((chunk_kernel<S, 1, true>), gf, dim3(64), 0, st, x, ldx, nch_f, lext, skip_f, P,
                     (const double*)init, (double*)nullptr, G, lpad, K, pad, L);
  SYG_CHECK_LAUNCH("sosfiltfilt forward");
  // backward sweep: grid positions of G outside [skip_b, skip_b + lext) are never written: the leading ones are
  // skipped by chunk 0, whatever lies behind the sequence cannot reach an output
  dim3 gb((unsigned)((nch_b + 63) / 64), (unsigned)B);
  hipLaunchKernelGGL((chunk_kernel<S, 0, false>), gb, dim3(64), 0, st, (const float*)G, lpad, nch_b, lext, skip_b, P,
                     (const double*)nullptr, zs, (float*)nullptr, (int64_t)0, K, pad, L);
  hipLaunchKernelGGL((scan_kernel<S, false>), gsc, dim3(256), 0, st, (const float*)G, lpad, nch_b, B, skip_b, P,
                     (const double*)zs, init, pad, L, lext);
  hipLaunchKernelGGL((chunk_kernel<S, 2, false>), gb, dim3(64), 0, st, (const float*)G, lpad, nch_b, lext, skip_b, P,
                     (const double*)init, (double*)nullptr, y, ldy, K, pad, L);
  SYG_CHECK_LAUNCH("sosfiltfilt backward");
  return SYG_OK;
}

}  // namespace
}  // namespace syg

using namespace syg;

// true when the clip-resident form (both sweeps in one launch, no workspace) takes this shape
static bool clip_resident(int64_t lext, int n_sections) {
  // (SYG_OPT_SOS_CLIP = 0 keeps the chunked path: the tests hold the two forms to the same results)
  return sos_clip_chunk(lext) > 0 && sos_clip_supported(n_sections) && option(SYG_OPT_SOS_CLIP) != 0;
}

extern "C" int64_t syg_sosfiltfilt_work_bytes(int64_t B, int64_t L, int padlen, int n_sections) {
  if (B < 1 || L < 1 || padlen < 0 || n_sections < 1 || n_sections > MAXS) return -1;
  const int64_t lext = L + 2 * (int64_t)padlen;
  if (clip_resident(lext, n_sections)) return 0;        // the clip stays in registers: `work` may be NULL
  const int64_t nch = (lext + (CS - 1) + CS - 1) / CS;   // grid = sequence + up to CS - 1 alignment positions
  return B * nch * CS * (int64_t)sizeof(float) + 2 * B * nch * 2 * n_sections * (int64_t)sizeof(double);
}

extern "C" int syg_sosfiltfilt_f32(const float* x, int64_t B, int64_t L, int64_t ldx, const double* sos_host,
                                   const double* zi_host, int n_sections, int padlen, float* y, int64_t ldy,
                                   void* work, void* stream) {
  SYG_REQUIRE(x && y && sos_host && zi_host, "sosfiltfilt: null pointer argument");
  SYG_REQUIRE(n_sections >= 1 && n_sections <= MAXS, "sosfiltfilt: n_sections must be in [1, %d] (got %d)", MAXS,
              n_sections);
  SYG_REQUIRE(B >= 1 && B <= 65535 && L >= 2 && ldx >= L && ldy >= L, "sosfiltfilt: bad B/L/ld");
  SYG_REQUIRE(padlen >= 0 && padlen < L, "The length of the input vector x must be greater than padlen, which is %d.",
              padlen);
  const int S = n_sections, D = 2 * S;
  SosParams P;
  memset(&P, 0, sizeof(P));
  double sos5[5 * MAXS];
  for (int s = 0; s < S; ++s) {
    const double* c = sos_host + 6 * s;
    SYG_REQUIRE(c[3] != 0.0, "sosfiltfilt: a0 of section %d is zero", s);
    P.b0[s] = sos5[5 * s + 0] = c[0] / c[3];
    P.b1[s] = sos5[5 * s + 1] = c[1] / c[3];
    P.b2[s] = sos5[5 * s + 2] = c[2] / c[3];
    P.a1[s] = sos5[5 * s + 3] = c[4] / c[3];
    P.a2[s] = sos5[5 * s + 4] = c[5] / c[3];
    P.zi[2 * s] = zi_host[2 * s];
    P.zi[2 * s + 1] = zi_host[2 * s + 1];
  }
  // A: homogeneous one-step map, column j = step(e_j, x = 0); A^CS (and A^(CS - skip) per sweep) on the host
  double A1[MAXD * MAXD] = {0};
  for (int j = 0; j < D; ++j) {
    double z[MAXD] = {0};
    z[j] = 1.0;
    host_step(sos5, S, z, 0.0);
    for (int i = 0; i < D; ++i) A1[i * MAXD + j] = z[i];
  }
  hipStream_t st = (hipStream_t)stream;
  // clips that fit a workgroup's registers: both sweeps in one launch (sosfilt_clip.hip); SYGNALS_AMD_SOS_CLIP=0 keeps
  // the chunked path below (development / tests)
  {
    const int64_t lext = L + 2 * (int64_t)padlen;
    const int cs = sos_clip_chunk(lext);
    if (clip_resident(lext, S)) {
      SosClipParams C;
      memset(&C, 0, sizeof(C));
      for (int s = 0; s < S; ++s) {
        C.b0[s] = P.b0[s]; C.b1[s] = P.b1[s]; C.b2[s] = P.b2[s]; C.a1[s] = P.a1[s]; C.a2[s] = P.a2[s];
        C.zi[2 * s] = P.zi[2 * s]; C.zi[2 * s + 1] = P.zi[2 * s + 1];
      }
      double Ap[MAXD * MAXD];
      mat_pow(A1, D, cs, Ap);
      for (int i = 0; i < D; ++i)
        for (int j = 0; j < D; ++j) C.apow[i * SOSC_MAXD + j] = Ap[i * MAXD + j];
      sos_clip_launch(x, B, (int)L, ldx, C, S, cs, padlen, y, ldy, st);
      SYG_CHECK_LAUNCH("sosfiltfilt (clip-resident)");
      return SYG_OK;
    }
  }
  SYG_REQUIRE(work, "sosfiltfilt: this shape takes the chunked path and needs the work buffer of syg_sosfiltfilt_work_bytes");
  mat_pow(A1, D, CS, P.apow);
  switch (S) {
    case 1: return launch_all<1>(x, B, L, ldx, P, padlen, y, ldy, work, st);
    case 2: return launch_all<2>(x, B, L, ldx, P, padlen, y, ldy, work, st);
    case 3: return launch_all<3>(x, B, L, ldx, P, padlen, y, ldy, work, st);
    case 4: return launch_all<4>(x, B, L, ldx, P, padlen, y, ldy, work, st);
    case 5: return launch_all<5>(x, B, L, ldx, P, padlen, y, ldy, work, st);
    case 6: return launch_all<6>(x, B, L, ldx, P, padlen, y, ldy, work, st);
    case 7: return launch_all<7>(x, B, L, ldx, P, padlen, y, ldy, work, st);
    default: return launch_all<8>(x, B, L, ldx, P, padlen, y, ldy, work, st);
  }
}
