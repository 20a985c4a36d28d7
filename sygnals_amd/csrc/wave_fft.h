// One-wave 1024-point complex FFT held 16 points per lane (radix 16 x 16 x 4), for kernels whose unit of work is a
// single wavefront: two lane exchanges through a 4.1 KiB per-wave LDS scratch, everything else in registers.
// The same transform as the FFT phase of stft_mel.hip (index maps validated by tools/wave_fft_model_v4.py); kept in a
// header of its own so that kernels with a relaxed register budget (welch_wave.hip) can use it without touching the
// register-starved headline kernel.
//
// Layout on exit: unit u = lane + 64 j (j = 0, 1) owns the bins k = kb_j + 256 d (d = 0..3), kb = (u >> 3) + 16 (u & 7),
// and their mirrors 1024 - k; lane 0 / unit 0 owns k = {0, 256, 128, 384} instead (mirrors {0, 768, 896, 640}) and
// bin 512.
#pragma once
#include "common.h"

namespace syg {
namespace wfft {

constexpr int PL2 = 132;                 // exchange-2 plane stride (complex): 128 group slots + 4 skew
constexpr int SC_COMPLEX = 4 * PL2;      // per-wave exchange scratch: 528 complex = 4224 B
constexpr int TW2_STRIDE = 18;           // complex entries per lane class (16 + 2 pad: distinct banks)
constexpr int TW2_COMPLEX = 4 * TW2_STRIDE;
constexpr int TW1_COMPLEX = 15 * 64;

typedef float v4f __attribute__((ext_vector_type(4)));

// exchange 2 (half buffer by c' & 7, planar in b'): slot of group (c, c') inside a plane
__device__ __forceinline__ int x2g(int c, int cp) { return (cp & 7) * 16 + ((c + 4 * ((cp & 7) >> 1)) & 15); }

struct Lane {
  int kb[2];          // kb of each unit
  int g0[2], g1[2];   // exchange-2 slot of the primary (c' < 8) / mirror (c' >= 8) group
};

__device__ __forceinline__ void init_lane(Lane& lc, int lane) {
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int u = lane + 64 * j, c = u >> 3, cp = u & 7;
    int cm = 16 - c, cmp = 15 - cp;
    if (c == 0) { cm = 0; cmp = (cp == 0) ? 8 : 16 - cp; }
    lc.g0[j] = x2g(c, cp);
    lc.g1[j] = x2g(cm, cmp);
    lc.kb[j] = c + 16 * cp;
  }
}

// bin owned by (lane, unit j, pair d)
__device__ __forceinline__ int bin_of(int lane, int j, int d) {
  if (lane == 0 && j == 0) return d == 0 ? 0 : d == 1 ? 256 : d == 2 ? 128 : 384;
  const int u = lane + 64 * j;
  return (u >> 3) + 16 * (u & 7) + 256 * d;
}

// LDS tables shared by the waves of a workgroup, built from tw[m] = exp(-2 pi i m / n_tw), n_tw a multiple of 2048:
//   tw2l[b' * TW2_STRIDE + c'] = W_64^(b' c'),   tw1l[(c - 1) * 64 + b] = W_1024^(b c)
__device__ __forceinline__ void init_tables(float2* tw2l, float2* tw1l, const float2* __restrict__ tw, int n_tw, int tid,
                                            int nthreads) {
  const int s = n_tw / 2048;
  if (tid < 64) tw2l[(tid >> 4) * TW2_STRIDE + (tid & 15)] = tw[s * 32 * (tid >> 4) * (tid & 15)];
  for (int i = tid; i < TW1_COMPLEX; i += nthreads) tw1l[i] = tw[s * 2 * (i & 63) * ((i >> 6) + 1)];
}

// Forward transform of z[64 a + lane] = v[a].  On exit zk[j][d] = Z[k], zm[j][d] = Z[1024 - k] for the bins of the header
// comment (Z[1024] = Z[0]); z512 = Z[512] (meaningful in lane 0).
__device__ __forceinline__ void cfft1024(float2 (&v)[16], const Lane& lc, float2* __restrict__ sc,
                                         const float2* __restrict__ tw1l, const float2* __restrict__ tw2l, int lane,
                                         float2 (&zk)[2][4], float2 (&zm)[2][4], float2& z512) {
  const int cl = lane >> 2, bp = lane & 3;
  // ---- pass 1: radix-16 over a (stride 64), twiddle W_1024^(b c)
  dft16(v);
#pragma unroll
  for (int c = 1; c < 16; ++c) v[c] = cmul(v[c], tw1l[(c - 1) * 64 + lane]);
  // ---- exchange 1 in two half-rounds through a 512-complex buffer (rows c = 8h .. 8h + 7; the lane pair (L, L + 32)
  // shares the reading of a row and v_permlane32_swap hands each lane the half it is missing)
  float2 t[16];
  {
    const int r7 = cl & 7;
    const int rbase = r7 * 64 + bp + 8 * (cl & 8) / 2;
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
  // ---- pass 2: lane = (c = lane >> 2, b' = lane & 3); radix-16 over a', twiddle W_64^(b' c')
  dft16(t);
  {
    typedef __attribute__((address_space(3))) const v4f* lds_v4;
    lds_v4 t4 = (lds_v4)(tw2l + bp * TW2_STRIDE);
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      const v4f tt = t4[m];
      if (m > 0) t[2 * m] = cmul(t[2 * m], make_float2(tt.x, tt.y));
      t[2 * m + 1] = cmul(t[2 * m + 1], make_float2(tt.z, tt.w));
    }
  }
  // ---- exchange 2 in two half-rounds (c' < 8: primaries, c' >= 8: mirrors); pass 3 = radix-4 over b'
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
  z512 = G[0][2];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int d = 0; d < 4; ++d) { zk[j][d] = G[j][d]; zm[j][d] = H[j][3 - d]; }
  // unit 0 (lane 0) pairs the self-mirrored groups (0,0) and (0,8) differently
  const bool sp = (lane == 0);
  zk[0][2] = sp ? H[0][0] : zk[0][2];
  zk[0][3] = sp ? H[0][1] : zk[0][3];
  zm[0][0] = sp ? G[0][0] : zm[0][0];
  zm[0][1] = sp ? G[0][3] : zm[0][1];
  zm[0][2] = sp ? H[0][3] : zm[0][2];
  zm[0][3] = sp ? H[0][2] : zm[0][3];
}

}  // namespace wfft
}  // namespace syg
