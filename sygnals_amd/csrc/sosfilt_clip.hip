// Zero-phase SOS filtering of clips that fit a workgroup's registers: BOTH sweeps of scipy.signal.sosfiltfilt in one
// launch, the clip read once and the result written once (apply_sos_filter, sygnals/core/filters.py:85-115).
//
// sosfilt.hip cuts time into chunks and needs, per sweep, a pass for the chunks' zero-state end states, a scan and the
// true pass -- the signal crosses HBM six times.  Here a workgroup of 256 lanes owns ONE clip: lane l holds the CS
// consecutive samples [l CS, (l + 1) CS) of the odd-extended signal in CS registers (CS = 64 ... 256: clips of up to
// 65536 - 2 padlen samples), and the same three steps run on the registers:
//   1  every lane runs the cascade over its samples from a zero state (lane 0 from the true start zi * ext[0])
//   2  inclusive prefix over the lanes, s_l <- s_l + M^(2^k) s_(l - 2^k), M = A^CS (host, float64), its squares formed
//      in LDS: lane l then holds the true state at the end of its chunk, lane l - 1 the state lane l starts from
//   3  every lane re-runs its samples from the true state; the outputs replace the samples (rounded to float32, as the
//      forward output is when sosfilt.hip stores it)
// and again backwards (samples walked from the top register down, lanes scanned from 255 to 0).  Positions past the end
// of the sequence (the tail of the last chunk, lanes with no samples) are filled with the last forward output y_last:
// the backward sweep starts from the steady state zi * y_last, which constant input leaves unchanged, so the idle
// positions in front of the true start need no special case (the forward sweep sees the zero tail behind its end, which
// nothing reads).  Recurrences in float64 exactly as in sosfilt.hip.
// The sample loops are fully unrolled (registers cannot be indexed at run time): four passes of CS steps.
#include "common.h"
#include "sosfilt_clip.h"

// development ablations (timing only, wrong results): -DSYG_SOSC_ABL=1 drops the zero-state passes, =2 all four passes
#if defined(SYG_SOSC_ABL) && SYG_SOSC_ABL == 2
#define SOSC_TRUE_PASS 0
#else
#define SOSC_TRUE_PASS 1
#endif

namespace syg {
namespace {

#ifdef SYG_SOSC_STAMP
__device__ unsigned long long sosc_stamp[8 * 1024];      // development build: per-clip wall-clock stamps (100 MHz)
#define SOSC_STAMP(k) do { __builtin_amdgcn_sched_barrier(0); __syncthreads(); if (threadIdx.x == 0 && blockIdx.x < 1024) sosc_stamp[blockIdx.x * 8 + (k)] = wall_clock64(); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define SOSC_STAMP(k) do { } while (0)
#endif

constexpr int NL = 256;      // lanes (chunks) per clip
constexpr int LSTR = 36;     // LDS tile row stride (floats): 32 samples + 4 pad (16-byte, conflict-free both ways)

typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));   // 16-byte access at 4-byte alignment

// sample i of the odd-extended signal (scipy padtype='odd'), zero behind its end
__device__ __forceinline__ float ext_at(const float* __restrict__ xb, int L, int pad, int lext, int i) {
  if (i < pad) return 2.f * xb[0] - xb[pad - i];
  if (i < pad + L) return xb[i - pad];
  if (i < lext) return 2.f * xb[L - 1] - xb[L - 2 - (i - pad - L)];
  return 0.f;
}

// four consecutive samples at a signal end (out of line: inlined 8 x CS / 32 times it is most of the kernel's code)
__device__ __noinline__ float4 ext4(const float* __restrict__ xb, int L, int pad, int lext, int i0) {
  return make_float4(ext_at(xb, L, pad, lext, i0), ext_at(xb, L, pad, lext, i0 + 1), ext_at(xb, L, pad, lext, i0 + 2),
                     ext_at(xb, L, pad, lext, i0 + 3));
}
__device__ __noinline__ void store4(float* __restrict__ yb, int L, int n0, float4 v) {
  if (n0 >= 0 && n0 < L) yb[n0] = v.x;
  if (n0 + 1 >= 0 && n0 + 1 < L) yb[n0 + 1] = v.y;
  if (n0 + 2 >= 0 && n0 + 2 < L) yb[n0 + 2] = v.z;
  if (n0 + 3 >= 0 && n0 + 3 < L) yb[n0 + 3] = v.w;
}

// UNIT: every section behind the first has b0 = b2 = 1 exactly -- what scipy.signal.butter(..., output='sos') returns for
// low-, high- and band-pass designs (numerators [1, +-2, 1]; the first section carries the gain) -- so that their steps
// are  y = u + z0;  z0 = b1 u + (z1 - a1 y);  z1 = u - a2 y : four float64 instructions instead of five (19 instead of 22
// per sample of an order-4 band-pass: the passes are float64-issue bound).  The host checks the coefficients.
template <int S, int CS, bool UNIT>
__global__ __launch_bounds__(NL, CS <= 192 ? 2 : 1) void sos_clip_kernel(const float* __restrict__ x, int64_t ldx, int L,
                                                                       int pad, SosClipParams P, float* __restrict__ y,
                                                                       int64_t ldy) {
  constexpr int D = 2 * S;
  constexpr int NT = CS / 32;                    // 32-sample tiles per chunk
  static_assert(CS % 32 == 0, "chunk length must be a multiple of the tile");
  __shared__ __attribute__((aligned(16))) float tile[4][64 * LSTR];
  __shared__ double ex[2 * NL * D];
  __shared__ double PW[8 * D * D];
  __shared__ float ylast;
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int64_t b = blockIdx.x;
  const float* xb = x + b * ldx;
  const int lext = L + 2 * pad;
  float xs[CS];
  SOSC_STAMP(0);

  // ---- load: 8 lanes x 16 bytes cover the 32 samples of one lane's tile; through LDS so that global accesses stay
  // along time while every lane ends up with its own chunk
  // every load of the clip is issued before the first transposition (one trip to HBM per clip, not one per tile): the
  // 16-byte pieces land in the registers that will hold the samples, tile t's eight pieces in xs[32 t .. 32 t + 31]
  float* tl = tile[w];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const int ch = r * 8 + (lane >> 3), part = lane & 7;
      const int i0 = (w * 64 + ch) * CS + 32 * t + 4 * part;          // sequence index of the piece's first sample
      float4 q;
      if (i0 >= pad && i0 + 3 < pad + L) {
        const f4u u = *reinterpret_cast<const f4u*>(xb + (i0 - pad));
        q = make_float4(u.x, u.y, u.z, u.w);
      } else {
        q = ext4(xb, L, pad, lext, i0);
      }
      xs[32 * t + 4 * r] = q.x; xs[32 * t + 4 * r + 1] = q.y; xs[32 * t + 4 * r + 2] = q.z; xs[32 * t + 4 * r + 3] = q.w;
    }
  }
  // tile t of the registers from piece order (8 lanes x 16 bytes per lane's row) to lane order (a lane's own 32 samples)
  auto to_lane_order = [&](int t) {
    wave_lds_sync();                             // the previous tile has been read
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const int ch = r * 8 + (lane >> 3), part = lane & 7;
      *reinterpret_cast<float4*>(&tl[ch * LSTR + 4 * part]) =
          make_float4(xs[32 * t + 4 * r], xs[32 * t + 4 * r + 1], xs[32 * t + 4 * r + 2], xs[32 * t + 4 * r + 3]);
    }
    wave_lds_sync();
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const float4 q = *reinterpret_cast<const float4*>(&tl[lane * LSTR + 4 * k]);
      xs[32 * t + 4 * k] = q.x; xs[32 * t + 4 * k + 1] = q.y; xs[32 * t + 4 * k + 2] = q.z; xs[32 * t + 4 * k + 3] = q.w;
    }
  };
  SOSC_STAMP(1);
  double z0[S], z1[S];
  auto step = [&](double u) {
#pragma unroll
    for (int s = 0; s < S; ++s) {
      if (UNIT && s > 0) {
        const double yv = u + z0[s];
        z0[s] = fma(P.b1[s], u, fma(-P.a1[s], yv, z1[s]));
        z1[s] = fma(-P.a2[s], yv, u);
        u = yv;
      } else {
        const double yv = fma(P.b0[s], u, z0[s]);
        z0[s] = fma(P.b1[s], u, fma(-P.a1[s], yv, z1[s]));
        z1[s] = fma(P.b2[s], u, -P.a2[s] * yv);
        u = yv;
      }
    }
    return u;
  };
  // M^(2^k), k = 0 .. 7, M = A^CS: squared once per workgroup, used by both sweeps' prefixes
  if (tid < D * D) PW[tid] = P.apow[(tid / D) * SOSC_MAXD + (tid % D)];
  __syncthreads();
#pragma unroll 1
  for (int k = 1; k < 8; ++k) {
    if (tid < D * D) {
      const double* Mp = PW + (k - 1) * D * D;
      const int r = tid / D, c = tid % D;
      double acc = 0.0;
#pragma unroll
      for (int q = 0; q < D; ++q) acc = fma(Mp[r * D + q], Mp[q * D + c], acc);
      PW[k * D * D + tid] = acc;
    }
    __syncthreads();
  }
  // inclusive prefix of the chunk end states over the scan order `ord` (0 = first chunk of the sweep); on exit
  // (z0, z1) = the true state this lane's chunk starts from (`start` for ord 0).  One barrier per round: the states
  // alternate between two LDS arrays (component-major: a lane's neighbours sit in the next banks).
  auto scan = [&](int ord, const double (&start)[D]) {
    double a[D];
#pragma unroll
    for (int s = 0; s < S; ++s) { a[2 * s] = z0[s]; a[2 * s + 1] = z1[s]; }
    int kk = 0;
#pragma unroll 1
    for (int off = 1; off < NL; off <<= 1, ++kk) {
      double* exk = ex + (kk & 1) * NL * D;
#pragma unroll
      for (int d = 0; d < D; ++d) exk[d * NL + ord] = a[d];
      __syncthreads();
      if (ord >= off) {
        const double* Mk = PW + kk * D * D;
        double v[D];
#pragma unroll
        for (int k = 0; k < D; ++k) v[k] = exk[k * NL + ord - off];
#pragma unroll
        for (int r = 0; r < D; ++r) {            // (fully unrolled: a[] stays in registers; Mk reads are LDS broadcasts)
          double p0 = 0.0, p1 = 0.0;
#pragma unroll
          for (int k = 0; k < D; k += 2) {
            p0 = fma(Mk[r * D + k], v[k], p0);
            p1 = fma(Mk[r * D + k + 1], v[k + 1], p1);
          }
          a[r] += p0 + p1;
        }
      }
    }
    double* exk = ex + (kk & 1) * NL * D;       // (eight rounds: the array the last round did not read)
#pragma unroll
    for (int d = 0; d < D; ++d) exk[d * NL + ord] = a[d];
    __syncthreads();
#pragma unroll
    for (int s = 0; s < S; ++s) {
      z0[s] = ord > 0 ? exk[(2 * s) * NL + ord - 1] : start[2 * s];
      z1[s] = ord > 0 ? exk[(2 * s + 1) * NL + ord - 1] : start[2 * s + 1];
    }
  };

  // ---- forward sweep
  double start[D];
  {
    const double first = (double)ext_at(xb, L, pad, lext, 0);
#pragma unroll
    for (int d = 0; d < D; ++d) start[d] = P.zi[d] * first;
  }
#pragma unroll
  for (int s = 0; s < S; ++s) { z0[s] = tid == 0 ? start[2 * s] : 0.0; z1[s] = tid == 0 ? start[2 * s + 1] : 0.0; }
  // (the fences keep the scheduler from converting dozens of samples ahead of their use: a register budget, not an order)
  // a tile is put in lane order when the zero-state pass reaches it: the later tiles' loads are still in flight
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    to_lane_order(t);
    __builtin_amdgcn_sched_barrier(0);
#if !defined(SYG_SOSC_ABL)
#pragma unroll
    for (int j = 32 * t; j < 32 * t + 32; ++j) { step((double)xs[j]); if ((j & 3) == 3) __builtin_amdgcn_sched_barrier(0); }
#endif
  }
  // (without this the compiler keeps the float64 conversions of pass 1 alive for pass 3: 2 CS registers more)
#pragma unroll
  for (int j = 0; j < CS; ++j) asm volatile("" : "+v"(xs[j]));
  SOSC_STAMP(2);
  scan(tid, start);
  SOSC_STAMP(3);
#pragma unroll
  for (int j = 0; j < (SOSC_TRUE_PASS ? CS : 0); ++j) {
    float o = (float)step((double)xs[j]);
    asm volatile("" : "+v"(o));                  // (the rounding happens here, not where the value is next used)
    xs[j] = o;
    if ((j & 3) == 3) __builtin_amdgcn_sched_barrier(0);
  }

  SOSC_STAMP(4);
  // ---- the last forward output, and the fill of the positions behind it
  {
    const int il = lext - 1;
    if (tid == il / CS) {
      float v = 0.f;
#pragma unroll
      for (int j = 0; j < CS; ++j) v = (j == il % CS) ? xs[j] : v;
      ylast = v;
    }
    __syncthreads();
    const float yl = ylast;
    const int nvalid = lext - tid * CS;          // positions j < nvalid of this lane are samples
#pragma unroll
    for (int j = 0; j < CS; ++j) xs[j] = (j < nvalid) ? xs[j] : yl;
#pragma unroll
    for (int d = 0; d < D; ++d) start[d] = P.zi[d] * (double)yl;
  }

  // ---- backward sweep: samples from the top register down, lanes from 255 to 0
  const int ord = NL - 1 - tid;
#pragma unroll
  for (int s = 0; s < S; ++s) { z0[s] = ord == 0 ? start[2 * s] : 0.0; z1[s] = ord == 0 ? start[2 * s + 1] : 0.0; }
#if !defined(SYG_SOSC_ABL)
#pragma unroll
  for (int j = CS - 1; j >= 0; --j) { step((double)xs[j]); if ((j & 3) == 0) __builtin_amdgcn_sched_barrier(0); }
#endif
#pragma unroll
  for (int j = 0; j < CS; ++j) asm volatile("" : "+v"(xs[j]));
  SOSC_STAMP(5);
  scan(ord, start);
  SOSC_STAMP(6);
  // the true backward pass, a tile at a time from the top; a finished tile goes out (y[n] = result at sequence index
  // n + pad) while the next one is computed
  float* yb = y + b * ldy;
#pragma unroll
  for (int t = NT - 1; t >= 0; --t) {
#pragma unroll
    for (int j = 32 * t + 31; j >= (SOSC_TRUE_PASS ? 32 * t : 32 * t + 32); --j) {
      float o = (float)step((double)xs[j]);
      asm volatile("" : "+v"(o));
      xs[j] = o;
      if ((j & 3) == 0) __builtin_amdgcn_sched_barrier(0);
    }
    wave_lds_sync();
#pragma unroll
    for (int k = 0; k < 8; ++k)
      *reinterpret_cast<float4*>(&tl[lane * LSTR + 4 * k]) =
          make_float4(xs[32 * t + 4 * k], xs[32 * t + 4 * k + 1], xs[32 * t + 4 * k + 2], xs[32 * t + 4 * k + 3]);
    wave_lds_sync();
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const int ch = r * 8 + (lane >> 3), part = lane & 7;
      const float4 v = *reinterpret_cast<const float4*>(&tl[ch * LSTR + 4 * part]);
      const int n0 = (w * 64 + ch) * CS + 32 * t + 4 * part - pad;
      if (n0 >= 0 && n0 + 3 < L) {
        f4u o; o.x = v.x; o.y = v.y; o.z = v.z; o.w = v.w;
        *reinterpret_cast<f4u*>(yb + n0) = o;
      } else {
        store4(yb, L, n0, v);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  SOSC_STAMP(7);
}

template <int S, bool UNIT>
int launch_su(int cs, dim3 grid, hipStream_t st, const float* x, int64_t ldx, int L, int pad, const SosClipParams& P, float* y,
              int64_t ldy) {
  switch (cs) {
    case 64: hipLaunchKernelGGL((sos_clip_kernel<S, 64, UNIT>), grid, dim3(NL), 0, st, x, ldx, L, pad, P, y, ldy); break;
    case 128: hipLaunchKernelGGL((sos_clip_kernel<S, 128, UNIT>), grid, dim3(NL), 0, st, x, ldx, L, pad, P, y, ldy); break;
    case 192: hipLaunchKernelGGL((sos_clip_kernel<S, 192, UNIT>), grid, dim3(NL), 0, st, x, ldx, L, pad, P, y, ldy); break;
    default: hipLaunchKernelGGL((sos_clip_kernel<S, 256, UNIT>), grid, dim3(NL), 0, st, x, ldx, L, pad, P, y, ldy); break;
  }
  return 0;
}
template <int S>
int launch_s(int cs, dim3 grid, hipStream_t st, const float* x, int64_t ldx, int L, int pad, const SosClipParams& P, float* y,
             int64_t ldy) {
  // (one section has no "sections behind the first": the general form)
  bool unit = S > 1;
  for (int s = 1; s < S; ++s) unit = unit && P.b0[s] == 1.0 && P.b2[s] == 1.0;
  if (S > 1 && unit) return launch_su<S, true>(cs, grid, st, x, ldx, L, pad, P, y, ldy);
  return launch_su<S, false>(cs, grid, st, x, ldx, L, pad, P, y, ldy);
}

}  // namespace

// chunk length for a clip of `lext` extended samples, 0 when the clip does not fit 256 lanes x 256 registers
int sos_clip_chunk(int64_t lext) {
  if (lext <= 256 * 64) return 64;
  if (lext <= 256 * 128) return 128;
  if (lext <= 256 * 192) return 192;
  if (lext <= 256 * 256) return 256;
  return 0;
}

bool sos_clip_supported(int n_sections) { return n_sections >= 1 && n_sections <= SOSC_MAXS; }

void sos_clip_launch(const float* x, int64_t B, int L, int64_t ldx, const SosClipParams& P, int n_sections, int cs, int pad,
                     float* y, int64_t ldy, hipStream_t st) {
  const dim3 grid((unsigned)B);
  switch (n_sections) {
    case 1: launch_s<1>(cs, grid, st, x, ldx, L, pad, P, y, ldy); break;
    case 2: launch_s<2>(cs, grid, st, x, ldx, L, pad, P, y, ldy); break;
    case 3: launch_s<3>(cs, grid, st, x, ldx, L, pad, P, y, ldy); break;
    default: launch_s<4>(cs, grid, st, x, ldx, L, pad, P, y, ldy); break;
  }
}

}  // namespace syg

#ifdef SYG_SOSC_STAMP
extern "C" int syg_debug_sosc_stamps(unsigned long long* host, int n) {
  return hipMemcpyFromSymbol(host, HIP_SYMBOL(syg::sosc_stamp), (size_t)n * sizeof(unsigned long long)) == hipSuccess ? 0 : -1;
}
#endif
