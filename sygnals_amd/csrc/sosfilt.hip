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
//   pass B  a short serial scan per clip, s <- A^CS s + zs (A^CS formed on the host in float64)
//           gives every chunk its true initial state
//   pass C  every chunk re-runs from the true state and writes its output (reversed, so the
//           backward sweep reads forward again)
// Recurrences run in float64 (the pole radius of the headline band-pass is 0.987; float32 state
// alone costs half the 1e-5 parity budget); signals are stored float32.  A wave owns 64 consecutive
// chunks of one clip and moves 64x32-sample tiles through LDS so that global accesses stay
// coalesced while each lane walks its own chunk.
#include "common.h"
#include <string.h>

namespace syg {
namespace {

constexpr int CS = 256;      // samples per chunk
constexpr int TS = 32;       // samples per LDS tile row group
constexpr int MAXS = 8;      // sections
constexpr int MAXD = 2 * MAXS;
constexpr int TSTRIDE = 65;  // LDS tile row stride (floats)

struct SosParams {
  double b0[MAXS], b1[MAXS], b2[MAXS], a1[MAXS], a2[MAXS];
  double zi[MAXD];
  double apow[MAXD * MAXD];  // A^CS, row-major D x D (padded to MAXD)
};

// Sample i of the odd-extended, zero-tailed signal (scipy's padtype='odd'): read straight from the clip, so the
// forward sweep needs no extended copy in HBM.
__device__ __forceinline__ float ext_at(const float* __restrict__ xb, int64_t L, int pad, int64_t lext, int64_t i) {
  if (i < pad) return 2.f * xb[0] - xb[pad - i];
  if (i < pad + L) return xb[i - pad];
  if (i < lext) return 2.f * xb[L - 1] - xb[L - 2 - (i - pad - L)];
  return 0.f;
}
typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));   // 16-byte load at 4-byte alignment

// ---- pass A / C
// MODE 0: zero-state pass, writes end states.  MODE 1: true pass, output reversed into `dst`
// (dst[lext-1-i]).  MODE 2: true pass, output reversed and trimmed into y[b, n], n = lext-1-pad-i.
// SRCX: the input is the clip itself, odd-extended on the fly (forward sweep; `in` = x, row stride `ldin`);
// otherwise a [B, lpad] work buffer (backward sweep).
template <int S, int MODE, bool SRCX>
__global__ __launch_bounds__(64) void chunk_kernel(const float* __restrict__ in, int64_t ldin, int64_t lpad, int nch,
                                                   int64_t lext, SosParams P, const double* __restrict__ init,
                                                   double* __restrict__ zs, float* __restrict__ dst, int64_t lddst,
                                                   int pad, int64_t L) {
  __shared__ float tile[TS * TSTRIDE];
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
  // cooperative coalesced load of one 64-chunk x 32-sample tile: 8 lanes x float4 cover one chunk's 32 samples.
  // The loads of tile k+1 are issued before the recurrence over tile k starts, so their latency hides behind it.
  auto load_tile = [&](int j0, float4 (&v)[8]) {
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const int ch = r * 8 + (lane >> 3), part = lane & 7;
      v[r] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (c0 + ch < nch) {
        const int64_t i0 = (int64_t)(c0 + ch) * CS + j0 + 4 * part;
        if (!SRCX) {
          v[r] = *reinterpret_cast<const float4*>(inb + i0);
        } else if (i0 >= pad && i0 + 3 < pad + L) {
          const f4u u4 = *reinterpret_cast<const f4u*>(inb + (i0 - pad));
          v[r] = make_float4(u4.x, u4.y, u4.z, u4.w);
        } else {
          v[r] = make_float4(ext_at(inb, L, pad, lext, i0), ext_at(inb, L, pad, lext, i0 + 1),
                             ext_at(inb, L, pad, lext, i0 + 2), ext_at(inb, L, pad, lext, i0 + 3));
        }
      }
    }
  };
  float4 nxt[8];
  load_tile(0, nxt);
  for (int j0 = 0; j0 < CS; j0 += TS) {
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const int ch = r * 8 + (lane >> 3), part = lane & 7;
      tile[(4 * part + 0) * TSTRIDE + ch] = nxt[r].x;
      tile[(4 * part + 1) * TSTRIDE + ch] = nxt[r].y;
      tile[(4 * part + 2) * TSTRIDE + ch] = nxt[r].z;
      tile[(4 * part + 3) * TSTRIDE + ch] = nxt[r].w;
    }
    if (j0 + TS < CS) load_tile(j0 + TS, nxt);
    __syncthreads();
#pragma unroll 4
    for (int j = 0; j < TS; ++j) {
      double u = (double)tile[j * TSTRIDE + lane];
#pragma unroll
      for (int s = 0; s < S; ++s) {
        const double yv = fma(P.b0[s], u, z0[s]);
        z0[s] = fma(P.b1[s], u, fma(-P.a1[s], yv, z1[s]));
        z1[s] = fma(P.b2[s], u, -P.a2[s] * yv);
        u = yv;
      }
      if (MODE != 0) tile[j * TSTRIDE + lane] = (float)u;
    }
    if (MODE != 0) {
      __syncthreads();
      // coalesced (reversed) store: 32 lanes cover one chunk's 32 samples
#pragma unroll 4
      for (int r = 0; r < 32; ++r) {
        const int ch = r * 2 + (lane >> 5), j = lane & 31;
        if (c0 + ch < nch) {
          const int64_t i = (int64_t)(c0 + ch) * CS + j0 + j;
          if (MODE == 1) {
            if (i < lext) dst[b * lddst + (lext - 1 - i)] = tile[j * TSTRIDE + ch];
          } else {
            const int64_t n = lext - 1 - pad - i;
            if (n >= 0 && n < L) dst[b * lddst + n] = tile[j * TSTRIDE + ch];
          }
        }
      }
    }
    __syncthreads();
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
struct RowDot {
  static __device__ __forceinline__ double run(const double (&arow)[D], double s, double acc) {
    acc = fma(arow[J], row_bcast<J>(s), acc);
    return RowDot<D, J + 1>::run(arow, s, acc);
  }
};
template <int D>
struct RowDot<D, D> {
  static __device__ __forceinline__ double run(const double (&)[D], double, double acc) { return acc; }
};

template <int S, bool SRCX>
__global__ __launch_bounds__(256) void scan_kernel(const float* __restrict__ in, int64_t ldin, int nch, int64_t B,
                                                   SosParams P, const double* __restrict__ zs,
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
  const float first = SRCX ? ext_at(in + bb * ldin, L, pad, lext, 0) : in[bb * ldin];
  double s = zi * (double)first;
  const int rc = (r < D) ? r : 0;
  const double* zp = zs + (int64_t)bb * nch * D + rc;
  double* ip = init + (int64_t)bb * nch * D + rc;
  constexpr int AHEAD = 4;
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
        s = RowDot<D, 0>::run(arow, s, zq[k]);
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

template <int S>
int launch_all(const float* x, int64_t B, int64_t L, int64_t ldx, const SosParams& P, int pad, float* y, int64_t ldy,
               void* work, hipStream_t st) {
  const int64_t lext = L + 2 * (int64_t)pad;
  const int nch = (int)((lext + CS - 1) / CS);
  const int64_t lpad = (int64_t)nch * CS;
  float* G = (float*)work;                      // forward output, reversed: [B, lpad]
  double* zs = (double*)(G + B * lpad);
  double* init = zs + B * (int64_t)nch * (2 * S);
  dim3 gch((unsigned)((nch + 63) / 64), (unsigned)B), gsc((unsigned)((B + 15) / 16));
  // forward sweep: reads the clip (odd extension on the fly), writes G
  hipLaunchKernelGGL((chunk_kernel<S, 0, true>), gch, dim3(64), 0, st, x, ldx, lpad, nch, lext, P,
                     (const double*)nullptr, zs, (float*)nullptr, (int64_t)0, pad, L);
  hipLaunchKernelGGL((scan_kernel<S, true>), gsc, dim3(256), 0, st, x, ldx, nch, B, P, (const double*)zs, init, pad, L,
                     lext);
  hipLaunchKernelGGL((chunk_kernel<S, 1, true>), gch, dim3(64), 0, st, x, ldx, lpad, nch, lext, P,
                     (const double*)init, (double*)nullptr, G, lpad, pad, L);
  SYG_CHECK_LAUNCH("sosfiltfilt forward");
  // backward sweep: G[.., lext..lpad) is never written by the forward pass and must read as zero
  hipLaunchKernelGGL((chunk_kernel<S, 0, false>), gch, dim3(64), 0, st, (const float*)G, lpad, lpad, nch, lext, P,
                     (const double*)nullptr, zs, (float*)nullptr, (int64_t)0, pad, L);
  hipLaunchKernelGGL((scan_kernel<S, false>), gsc, dim3(256), 0, st, (const float*)G, lpad, nch, B, P,
                     (const double*)zs, init, pad, L, lext);
  hipLaunchKernelGGL((chunk_kernel<S, 2, false>), gch, dim3(64), 0, st, (const float*)G, lpad, lpad, nch, lext, P,
                     (const double*)init, (double*)nullptr, y, ldy, pad, L);
  SYG_CHECK_LAUNCH("sosfiltfilt backward");
  return SYG_OK;
}

}  // namespace
}  // namespace syg

using namespace syg;

extern "C" int64_t syg_sosfiltfilt_work_bytes(int64_t B, int64_t L, int padlen, int n_sections) {
  if (B < 1 || L < 1 || padlen < 0 || n_sections < 1 || n_sections > MAXS) return -1;
  const int64_t lext = L + 2 * (int64_t)padlen;
  const int64_t nch = (lext + CS - 1) / CS;
  return B * nch * CS * (int64_t)sizeof(float) + 2 * B * nch * 2 * n_sections * (int64_t)sizeof(double);
}

extern "C" int syg_sosfiltfilt_f32(const float* x, int64_t B, int64_t L, int64_t ldx, const double* sos_host,
                                   const double* zi_host, int n_sections, int padlen, float* y, int64_t ldy,
                                   void* work, void* stream) {
  SYG_REQUIRE(x && y && sos_host && zi_host && work, "sosfiltfilt: null pointer argument");
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
  // A: homogeneous one-step map, column j = step(e_j, x = 0); then A^CS by repeated squaring
  double A[MAXD * MAXD] = {0}, T[MAXD * MAXD];
  for (int j = 0; j < D; ++j) {
    double z[MAXD] = {0};
    z[j] = 1.0;
    host_step(sos5, S, z, 0.0);
    for (int i = 0; i < D; ++i) A[i * MAXD + j] = z[i];
  }
  for (int sq = 0; (1 << sq) < CS; ++sq) {
    for (int i = 0; i < D; ++i)
      for (int j = 0; j < D; ++j) {
        double acc = 0.0;
        for (int k = 0; k < D; ++k) acc += A[i * MAXD + k] * A[k * MAXD + j];
        T[i * MAXD + j] = acc;
      }
    for (int i = 0; i < D; ++i)
      for (int j = 0; j < D; ++j) A[i * MAXD + j] = T[i * MAXD + j];
  }
  for (int i = 0; i < MAXD * MAXD; ++i) P.apow[i] = A[i];
  hipStream_t st = (hipStream_t)stream;
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
