// Shared helpers for the sygnals_hip kernels (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

#include "../../include/sygnals_hip.h"

namespace syg {

// thread-local last-error string, defined in capi.hip
void set_error(const char* fmt, ...);
// process-wide option (syg_set_option / SYG_OPT_*), defined in capi.hip
int option(int key);

#define SYG_REQUIRE(cond, ...)                 \
  do {                                         \
    if (!(cond)) {                             \
      syg::set_error(__VA_ARGS__);             \
      return SYG_E_INVALID;                    \
    }                                          \
  } while (0)

#define SYG_CHECK_LAUNCH(what)                                               \
  do {                                                                       \
    hipError_t e_ = hipGetLastError();                                       \
    if (e_ != hipSuccess) {                                                  \
      syg::set_error("%s: HIP launch failed: %s", what, hipGetErrorString(e_)); \
      return SYG_E_LAUNCH;                                                   \
    }                                                                        \
  } while (0)

__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ float2 cmul(float2 a, float2 b) {
  return make_float2(fmaf(-a.y, b.y, a.x * b.x), fmaf(a.y, b.x, a.x * b.y));
}
// a * conj(b)
__device__ __forceinline__ float2 cmulc(float2 a, float2 b) {
  return make_float2(fmaf(a.y, b.y, a.x * b.x), fmaf(a.y, b.x, -a.x * b.y));
}
__device__ __forceinline__ float2 mul_mi(float2 a) { return make_float2(a.y, -a.x); }  // a * (-i)
__device__ __forceinline__ float2 mul_pi(float2 a) { return make_float2(-a.y, a.x); }  // a * (+i)

// Forward radix-4 butterfly: o[q] = sum_p a[p] * (-i)^(p*q)
__device__ __forceinline__ void bfly4(float2 a0, float2 a1, float2 a2, float2 a3, float2& o0, float2& o1,
                                      float2& o2, float2& o3) {
  float2 t0 = cadd(a0, a2), t1 = csub(a0, a2), t2 = cadd(a1, a3), t3 = csub(a1, a3);
  o0 = cadd(t0, t2);
  o2 = csub(t0, t2);
  o1 = make_float2(t1.x + t3.y, t1.y - t3.x);
  o3 = make_float2(t1.x - t3.y, t1.y + t3.x);
}

// Fused ends of the strided transforms (syg_fft_*_strided_ex_f32): a REAL input array (imaginary part 0: no packing pass),
// the analytic-signal weights of scipy.signal.hilbert applied to the loaded element (its index inside the row is its
// frequency bin: 1 at 0 and n / 2, 2 below n / 2, 0 above: no masking pass), magnitudes as the output (no |.| pass).
constexpr int SYG_FFT_REAL_IN = 1, SYG_FFT_ABS_OUT = 2, SYG_FFT_PAIR_IN = 4;
// element at position pos of the row that starts at element offset ibase.  PAIR_IN: `in` is a real array whose row (in_valid
// samples, zero beyond) is read as the complex sequence (x[2 p], x[2 p + 1]) -- the packed form of a real-input transform of
// twice the length (rfft_conv) without its packing pass; ibase is then the row's offset in FLOATS.
__device__ __forceinline__ float2 fft_load(const float2* __restrict__ in, int64_t ibase, int64_t pos, int flags, int64_t mask_n,
                                           int64_t in_valid) {
  float2 v;
  if (flags & SYG_FFT_PAIR_IN) {
    const float* r = reinterpret_cast<const float*>(in) + ibase;
    v = make_float2(2 * pos < in_valid ? r[2 * pos] : 0.f, 2 * pos + 1 < in_valid ? r[2 * pos + 1] : 0.f);
  } else if (flags & SYG_FFT_REAL_IN) {
    v = make_float2(reinterpret_cast<const float*>(in)[ibase + pos], 0.f);
  } else {
    v = in[ibase + pos];
  }
  if (mask_n > 0) {
    const float h = (pos == 0 || 2 * pos == mask_n) ? 1.f : (2 * pos < mask_n ? 2.f : 0.f);
    v.x *= h; v.y *= h;
  }
  return v;
}
__device__ __forceinline__ void fft_store(float2* __restrict__ out, int64_t idx, float2 v, int flags) {
  if (flags & SYG_FFT_ABS_OUT) reinterpret_cast<float*>(out)[idx] = sqrtf(fmaf(v.x, v.x, v.y * v.y));
  else out[idx] = v;
}

// In-register forward 16-point DFT, natural order in and out (radix-4 x radix-4).
__device__ __forceinline__ void dft16(float2 (&v)[16]) {
  constexpr float C1 = 0.92387953251128673848f, S1 = 0.38268343236508977173f, R = 0.70710678118654752440f;
  float2 s[16];  // s[i + 4q]
#pragma unroll
  for (int i = 0; i < 4; ++i) bfly4(v[i], v[i + 4], v[i + 8], v[i + 12], s[i], s[i + 4], s[i + 8], s[i + 12]);
  // twiddle s[i + 4q] *= W16^(i*q)
  s[1 + 4] = cmul(s[1 + 4], make_float2(C1, -S1));   // e=1
  s[2 + 4] = cmul(s[2 + 4], make_float2(R, -R));     // e=2
  s[3 + 4] = cmul(s[3 + 4], make_float2(S1, -C1));   // e=3
  s[1 + 8] = cmul(s[1 + 8], make_float2(R, -R));     // e=2
  s[2 + 8] = mul_mi(s[2 + 8]);                       // e=4
  s[3 + 8] = cmul(s[3 + 8], make_float2(-R, -R));    // e=6
  s[1 + 12] = cmul(s[1 + 12], make_float2(S1, -C1)); // e=3
  s[2 + 12] = cmul(s[2 + 12], make_float2(-R, -R));  // e=6
  s[3 + 12] = cmul(s[3 + 12], make_float2(-C1, S1)); // e=9
#pragma unroll
  for (int q = 0; q < 4; ++q)
    bfly4(s[4 * q], s[4 * q + 1], s[4 * q + 2], s[4 * q + 3], v[q], v[q + 4], v[q + 8], v[q + 12]);
}

// In-register forward 8-point DFT, natural order in and out (radix-2 x radix-4).
__device__ __forceinline__ void dft8(float2 (&a)[8]) {
  constexpr float R = 0.70710678118654752440f;
  float2 e0, e1, e2, e3, o0, o1, o2, o3;
  bfly4(a[0], a[2], a[4], a[6], e0, e1, e2, e3);
  bfly4(a[1], a[3], a[5], a[7], o0, o1, o2, o3);
  const float2 t1 = make_float2(R * (o1.x + o1.y), R * (o1.y - o1.x));      // W_8^1 o1
  const float2 t2 = mul_mi(o2);                                             // W_8^2 o2
  const float2 t3 = make_float2(R * (o3.y - o3.x), -R * (o3.x + o3.y));     // W_8^3 o3
  a[0] = cadd(e0, o0); a[4] = csub(e0, o0);
  a[1] = cadd(e1, t1); a[5] = csub(e1, t1);
  a[2] = cadd(e2, t2); a[6] = csub(e2, t2);
  a[3] = cadd(e3, t3); a[7] = csub(e3, t3);
}

// power_to_db on the hardware logarithm: 10 log10(x / ref) = 10 log10(2) (log2 x - log2 ref).  v_log_f32 is good to one
// ulp of the log2 value (<= 2.4e-6 at |log2| = 40, i.e. 7e-6 dB) against ~40 instructions for the library log10f -- the
// conversion was most of logmel_dct's time; the difference of logs is exactly 0 at x == ref.  Inputs must be normal
// numbers (callers clamp with amin >= FLT_MIN).
constexpr float SYG_DB_PER_LOG2 = 3.01029995663981195f;
__device__ __forceinline__ float syg_log2(float x) { return __builtin_amdgcn_logf(x); }

// Lanes of one wave exchange data through LDS without a workgroup barrier (the LDS executes a wave's
// DS instructions in order).  To the COMPILER that is a data race: it may assume a lane that did not store
// re-reads unchanged memory.  This wavefront-scope release/acquire pair emits no instruction but makes every
// later LDS read observe the stores of the other lanes.
__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// ---------------------------------------------------------------------------------------------
// Stockham autosort FFT of N = 2^m complex points in LDS (ping-pong buffers x, y; N >= 2), shared by the generic
// FFT / STFT / Welch kernels (a workgroup of `nt` threads) and the CQT (one wave, nt = 64 -- its callers keep
// all waves of a workgroup on the same trip counts, so the barriers line up).  tw[k] = W_N^k.
//   * radix-8 passes (8 points per thread in registers) with a radix-4 / radix-2 tail: 4 passes for N = 2048
//     instead of 6 radix-4/2 ones;
//   * index arithmetic by shifts (the stride is a power of two);
//   * the intermediate buffers are XOR-swizzled (i ^ ((i >> 4) & 15)): the stride-8 stores of the early passes
//     would otherwise hit two LDS banks with every lane; input and result stay in natural order.
// Returns the buffer that holds the result.
__device__ __forceinline__ int fft_swz(int i, bool on) { return on ? (i ^ ((i >> 4) & 15)) : i; }

// WAVE: the transform belongs to ONE wave (nt = 64, buffers of its own): the passes are ordered by the wave itself
// (wave_lds_sync) instead of a workgroup barrier that would tie independent transforms of other waves together.
template <bool WAVE = false>
__device__ __forceinline__ float2* block_fft(float2* x, float2* y, int N, const float2* __restrict__ tw, int tid,
                                             int nt) {
  auto pass_sync = [] { if (WAVE) wave_lds_sync(); else __syncthreads(); };
  int lN = 0;
  while ((1 << lN) < N) ++lN;
  const int n8 = lN / 3, tail = lN - 3 * n8;          // radix-8 passes, then a radix-4 (tail 2) or radix-2 (tail 1)
  const int npass = n8 + (tail ? 1 : 0);
  int ls = 0, pass = 0;                                // stride s = 1 << ls; sub-length n = N >> ls
  for (int k = 0; k < n8; ++k, ++pass) {
    const bool sin = pass > 0, sout = pass < npass - 1;
    const int s = 1 << ls, q8 = N >> (ls + 3);
    for (int i = tid; i < (N >> 3); i += nt) {
      const int p = i >> ls, q = i & (s - 1);
      float2 a[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) a[j] = x[fft_swz(q + ((p + j * q8) << ls), sin)];
      dft8(a);
      const int e = p << ls;                           // W_n^p = W_N^(p s)
      const int ob = q + ((8 * p) << ls);
      y[fft_swz(ob, sout)] = a[0];
      // W^(m e), m = 1..7, from three table reads: w3 = w1 w2, w5 = w1 w4, w6 = w2 w4, w7 = w3 w4 (one extra
      // rounding, ~6e-8, against four fewer dependent global loads per butterfly)
      const float2 w1 = tw[e], w2 = tw[2 * e], w4 = tw[4 * e];
      const float2 w3 = cmul(w1, w2), w5 = cmul(w1, w4), w6 = cmul(w2, w4), w7 = cmul(w3, w4);
      y[fft_swz(ob + (1 << ls), sout)] = cmul(a[1], w1);
      y[fft_swz(ob + (2 << ls), sout)] = cmul(a[2], w2);
      y[fft_swz(ob + (3 << ls), sout)] = cmul(a[3], w3);
      y[fft_swz(ob + (4 << ls), sout)] = cmul(a[4], w4);
      y[fft_swz(ob + (5 << ls), sout)] = cmul(a[5], w5);
      y[fft_swz(ob + (6 << ls), sout)] = cmul(a[6], w6);
      y[fft_swz(ob + (7 << ls), sout)] = cmul(a[7], w7);
    }
    pass_sync();
    float2* t = x; x = y; y = t;
    ls += 3;
  }
  if (tail == 2) {
    const bool sin = pass > 0;                         // last pass: natural output
    const int s = 1 << ls;                             // n = 4: p = 0, all twiddles are 1
    for (int i = tid; i < (N >> 2); i += nt) {
      float2 o0, o1, o2, o3;
      bfly4(x[fft_swz(i, sin)], x[fft_swz(i + s, sin)], x[fft_swz(i + 2 * s, sin)], x[fft_swz(i + 3 * s, sin)], o0, o1,
            o2, o3);
      y[i] = o0; y[i + s] = o1; y[i + 2 * s] = o2; y[i + 3 * s] = o3;
    }
    pass_sync();
    float2* t = x; x = y; y = t;
  } else if (tail == 1) {
    const bool sin = pass > 0;
    const int s = 1 << ls;                             // n = 2
    for (int i = tid; i < (N >> 1); i += nt) {
      const float2 a = x[fft_swz(i, sin)], b = x[fft_swz(i + s, sin)];
      y[i] = cadd(a, b);
      y[i + s] = csub(a, b);
    }
    pass_sync();
    float2* t = x; x = y; y = t;
  }
  return x;
}

// ---------------------------------------------------------------------------------------------
// Wave-wide (64-lane) reductions and scans on the DPP path.  hipcc lowers __shfl_xor to ds_bpermute_b32
// (the LDS crossbar, >100 cycles per hop, six dependent hops per reduction); the DPP row operations below
// cost a few cycles each: butterfly inside each row of 16 lanes (quad_perm, row_half_mirror, row_mirror),
// then the four row results are combined through v_readlane.
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
template <int CTRL>
__device__ __forceinline__ int dpp_i(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, true); }
constexpr int DPP_QP_1032 = 0xB1, DPP_QP_2301 = 0x4E, DPP_ROW_HALF_MIRROR = 0x141, DPP_ROW_MIRROR = 0x140;
constexpr int DPP_WAVE_SHR1 = 0x138;     // whole-wave shift right by one lane (gfx9 DPP)
constexpr int DPP_ROW_SHR1 = 0x111, DPP_ROW_SHR2 = 0x112, DPP_ROW_SHR4 = 0x114, DPP_ROW_SHR8 = 0x118;

__device__ __forceinline__ float rl_f(float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); }

// after the four in-row steps every lane of a row holds its row's result; two row broadcasts (lane 15 of the row
// below into rows 1 and 3, then lane 31 into rows 2 and 3) fold the rows into row 3, read back from lane 63
constexpr int DPP_ROW_BCAST15 = 0x142, DPP_ROW_BCAST31 = 0x143;
template <int CTRL, int ROWMASK>
__device__ __forceinline__ float dpp_rows_f(float v) {      // lanes of rows outside ROWMASK keep v
  return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), CTRL, ROWMASK, 0xF, false));
}
template <int CTRL, int ROWMASK>
__device__ __forceinline__ int dpp_rows_i(int v) { return __builtin_amdgcn_update_dpp(v, v, CTRL, ROWMASK, 0xF, false); }

__device__ __forceinline__ float wave_sum(float v) {
  v += dpp_f<DPP_QP_1032>(v);
  v += dpp_f<DPP_QP_2301>(v);
  v += dpp_f<DPP_ROW_HALF_MIRROR>(v);
  v += dpp_f<DPP_ROW_MIRROR>(v);
  return (rl_f(v, 0) + rl_f(v, 16)) + (rl_f(v, 32) + rl_f(v, 48));
}
__device__ __forceinline__ float wave_max(float v) {
  v = fmaxf(v, dpp_f<DPP_QP_1032>(v));
  v = fmaxf(v, dpp_f<DPP_QP_2301>(v));
  v = fmaxf(v, dpp_f<DPP_ROW_HALF_MIRROR>(v));
  v = fmaxf(v, dpp_f<DPP_ROW_MIRROR>(v));
  v = fmaxf(v, dpp_rows_f<DPP_ROW_BCAST15, 0xA>(v));
  v = fmaxf(v, dpp_rows_f<DPP_ROW_BCAST31, 0xC>(v));
  return rl_f(v, 63);
}
__device__ __forceinline__ float wave_min(float v) {
  v = fminf(v, dpp_f<DPP_QP_1032>(v));
  v = fminf(v, dpp_f<DPP_QP_2301>(v));
  v = fminf(v, dpp_f<DPP_ROW_HALF_MIRROR>(v));
  v = fminf(v, dpp_f<DPP_ROW_MIRROR>(v));
  v = fminf(v, dpp_rows_f<DPP_ROW_BCAST15, 0xA>(v));
  v = fminf(v, dpp_rows_f<DPP_ROW_BCAST31, 0xC>(v));
  return rl_f(v, 63);
}
// max of `a` and min of `b` over the wave in one interleaved chain of fused DPP operations (v_max_f32_dpp reads its
// first source through the DPP network, so a step is ONE instruction; the compiler's own lowering of the builtins is a
// v_mov_dpp plus a canonicalising v_max plus the v_max itself).  A DPP operand written by the previous VALU
// instruction needs two wait states: the other chain's step fills one, s_nop 0 the second.
__device__ __forceinline__ void wave_maxmin(float& a, float& b) {
  asm("s_nop 1\n\t"
      "v_max_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "v_min_f32_dpp %1, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 0\n\t"
      "v_max_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
      "v_min_f32_dpp %1, %1, %1 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 0\n\t"
      "v_max_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
      "v_min_f32_dpp %1, %1, %1 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 0\n\t"
      "v_max_f32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\t"
      "v_min_f32_dpp %1, %1, %1 row_mirror row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 0\n\t"
      "v_max_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
      "v_min_f32_dpp %1, %1, %1 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
      "s_nop 0\n\t"
      "v_max_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
      "v_min_f32_dpp %1, %1, %1 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
      "s_nop 0"
      : "+v"(a), "+v"(b));
  a = rl_f(a, 63);
  b = rl_f(b, 63);
}
__device__ __forceinline__ int wave_min_i(int v) {
  v = min(v, dpp_i<DPP_QP_1032>(v));
  v = min(v, dpp_i<DPP_QP_2301>(v));
  v = min(v, dpp_i<DPP_ROW_HALF_MIRROR>(v));
  v = min(v, dpp_i<DPP_ROW_MIRROR>(v));
  v = min(v, dpp_rows_i<DPP_ROW_BCAST15, 0xA>(v));
  v = min(v, dpp_rows_i<DPP_ROW_BCAST31, 0xC>(v));
  return __builtin_amdgcn_readlane(v, 63);
}
__device__ __forceinline__ int wave_sum_i(int v) {
  v += dpp_i<DPP_QP_1032>(v);
  v += dpp_i<DPP_QP_2301>(v);
  v += dpp_i<DPP_ROW_HALF_MIRROR>(v);
  v += dpp_i<DPP_ROW_MIRROR>(v);
  return (__builtin_amdgcn_readlane(v, 0) + __builtin_amdgcn_readlane(v, 16)) +
         (__builtin_amdgcn_readlane(v, 32) + __builtin_amdgcn_readlane(v, 48));
}
// exclusive prefix sum over the 64 lanes: Hillis-Steele inside each row of 16 (row_shr with zero fill),
// then the sums of the lower rows are added
__device__ __forceinline__ float wave_excl_scan(float v, int lane) {
  float s = v;
  s += dpp_f<DPP_ROW_SHR1>(s);
  s += dpp_f<DPP_ROW_SHR2>(s);
  s += dpp_f<DPP_ROW_SHR4>(s);
  s += dpp_f<DPP_ROW_SHR8>(s);
  const float r0 = rl_f(s, 15), r1 = rl_f(s, 31), r2 = rl_f(s, 47);
  const int row = lane >> 4;
  const float off = (row >= 1 ? r0 : 0.f) + (row >= 2 ? r1 : 0.f) + (row >= 3 ? r2 : 0.f);
  return s + off - v;
}

}  // namespace syg
