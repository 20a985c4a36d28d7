// Generic power-of-two FFT in LDS (Stockham autosort, radix 8 with a radix-4/2 tail: block_fft in common.h) and the
// kernels built on it: batched c2c FFT/IFFT, framed STFT for any power-of-two n_fft, Welch PSD.
//
// Reference behaviour reproduced:
//   scipy.fft.fft / ifft            (compute_fft / compute_ifft, sygnals/core/dsp.py:104, 151)
//   librosa.stft                    (compute_stft, sygnals/core/dsp.py:216-224)
//   scipy.signal.welch              (compute_psd_welch, sygnals/core/dsp.py:545-555)
// One workgroup per transform; the two ping-pong buffers live in LDS (<= 128 KiB for
// n = 8192), so every transform reads its input once and writes its output once.
#include "common.h"

namespace syg {
namespace {

constexpr int MAX_N = 8192;

// (block_fft: common.h)

// Real-input split: Z = FFT_M(z), z[m] = x[2m] + i x[2m+1]; returns X[k], k in [0, M].
// tw2[k] = W_{2M}^k.
__device__ __forceinline__ float2 rfft_bin(const float2* Z, int M, int k, const float2* __restrict__ tw2) {
  const float2 zk = Z[k & (M - 1)], zm = Z[(M - k) & (M - 1)];
  const float2 E = make_float2(zk.x + zm.x, zk.y - zm.y);
  const float2 O = make_float2(zk.y + zm.y, zm.x - zk.x);
  const float2 wO = cmul(tw2[k], O);
  return make_float2(0.5f * (E.x + wO.x), 0.5f * (E.y + wO.y));
}

// ---------------------------------------------------------------------------------
__global__ void fft_pow2_kernel(const float2* __restrict__ in, float2* __restrict__ out, int n, int inverse,
                                const float2* __restrict__ tw) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float2* x = reinterpret_cast<float2*>(lds);
  float2* y = x + n;
  const int tid = threadIdx.x, nt = blockDim.x;
  const int64_t base = (int64_t)blockIdx.x * n;
  // inverse via conj(FFT(conj(x))) / n
  for (int i = tid; i < n; i += nt) {
    float2 v = in[base + i];
    if (inverse) v.y = -v.y;
    x[i] = v;
  }
  __syncthreads();
  float2* r = block_fft(x, y, n, tw, tid, nt);
  const float sc = inverse ? 1.0f / (float)n : 1.0f;
  for (int i = tid; i < n; i += nt) {
    float2 v = r[i];
    out[base + i] = inverse ? make_float2(v.x * sc, -v.y * sc) : v;
  }
}

// Framed STFT, one workgroup per frame; tw = W_{n_fft}^k (n_fft entries).
__global__ void stft_pow2_kernel(const float* __restrict__ y, int64_t L, int64_t ldy, int n_fft, int hop, int pad,
                                 int64_t T, const float* __restrict__ win, const float2* __restrict__ tw,
                                 float2* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int M = n_fft >> 1;
  float2* x = reinterpret_cast<float2*>(lds);
  float2* z = x + M;
  const int tid = threadIdx.x, nt = blockDim.x;
  const int64_t b = blockIdx.x / T, t = blockIdx.x % T;
  const float* yb = y + b * ldy;
  const int64_t s0 = t * (int64_t)hop - pad;
  for (int m = tid; m < M; m += nt) {
    const int64_t s = s0 + 2 * m;
    const float a = (s >= 0 && s < L) ? yb[s] * win[2 * m] : 0.f;
    const float c = (s + 1 >= 0 && s + 1 < L) ? yb[s + 1] * win[2 * m + 1] : 0.f;
    x[m] = make_float2(a, c);
  }
  __syncthreads();
  // the M-point transform uses W_M^k = W_{n_fft}^{2k}: pass a stride-2 view via index doubling
  // (block_fft indexes tw[e], e < M) -> precomputed separate table region: tw + n_fft holds W_M^k
  float2* Z = block_fft(x, z, M, tw + n_fft, tid, nt);
  float2* o = out + (b * T + t) * (int64_t)(M + 1);
  for (int k = tid; k <= M; k += nt) o[k] = rfft_bin(Z, M, k, tw);
}

__global__ void cabs_pow_kernel(const float2* __restrict__ x, int64_t n, int power, float* __restrict__ out) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float2 v = x[i];
    const float p = fmaf(v.x, v.x, v.y * v.y);
    out[i] = (power == 2) ? p : sqrtf(p);
  }
}

// mel[b, m, t] = sum_f basis[m, f] * P[b, t, f] on the matrix cores (v_mfma_f32_16x16x4_f32, exact fp32): the dense
// mel filterbank of librosa.feature.melspectrogram (manager.py:219-222) for every frame length the fused kernel does not
// take.  A wave owns 16 frames x up to MELT_PER_WAVE tiles of 16 mel rows; per group of 16 bins each lane loads 16
// bytes of its frame's power row (B operand) and 16 bytes of each of its basis rows (A operands) and feeds four
// consecutive k-steps from them (the k order inside a group is permuted the same way on both sides); the last,
// partial group of a row (F = n_fft/2 + 1 is odd) is loaded element-wise with zero fill.
constexpr int MELT_PER_WAVE = 4;
typedef float mel_v4f __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void mel_dense_mfma_kernel(const float* __restrict__ P, int64_t B, int64_t T, int F,
                                                             const float* __restrict__ basis, int M,
                                                             float* __restrict__ mel, int64_t tiles_per_clip) {
  const int lane = threadIdx.x & 63;
  const int64_t wid = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int64_t b = wid / tiles_per_clip;
  if (b >= B) return;
  const int64_t t0 = (wid - b * tiles_per_clip) * 16;
  const int mt0 = blockIdx.y * MELT_PER_WAVE;
  const int n = lane & 15, kk = lane >> 4;
  const int64_t t = t0 + n;
  const bool tok = t < T;
  const float* prow = P + (b * T + (tok ? t : T - 1)) * (int64_t)F + 4 * kk;
  const float* arow[MELT_PER_WAVE];
  bool aok[MELT_PER_WAVE];
#pragma unroll
  for (int r = 0; r < MELT_PER_WAVE; ++r) {
    const int m = 16 * (mt0 + r) + n;
    aok[r] = m < M;
    arow[r] = basis + (int64_t)(aok[r] ? m : 0) * F + 4 * kk;
  }
  mel_v4f acc[MELT_PER_WAVE];
#pragma unroll
  for (int r = 0; r < MELT_PER_WAVE; ++r) acc[r] = mel_v4f{0.f, 0.f, 0.f, 0.f};
  const int full = F >> 4;                       // groups of 16 bins that lie entirely inside a row
  auto step = [&](const float (&pb)[4], const float (&pa)[MELT_PER_WAVE][4]) {
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int r = 0; r < MELT_PER_WAVE; ++r) acc[r] = __builtin_amdgcn_mfma_f32_16x16x4f32(pa[r][u], pb[u], acc[r], 0, 0, 0);
  };
  for (int s = 0; s < full; ++s) {
    float pb[4], pa[MELT_PER_WAVE][4];
    {
      // (4-byte aligned 16-byte loads: F is odd, rows are not 16-byte aligned)
      const float* q = prow + 16 * s;
      pb[0] = tok ? q[0] : 0.f; pb[1] = tok ? q[1] : 0.f; pb[2] = tok ? q[2] : 0.f; pb[3] = tok ? q[3] : 0.f;
    }
#pragma unroll
    for (int r = 0; r < MELT_PER_WAVE; ++r) {
      const float* q = arow[r] + 16 * s;
#pragma unroll
      for (int u = 0; u < 4; ++u) pa[r][u] = aok[r] ? q[u] : 0.f;
    }
    step(pb, pa);
  }
  if (full * 16 < F) {                           // the partial last group
    float pb[4], pa[MELT_PER_WAVE][4];
    const int f0 = full * 16 + 4 * kk;
#pragma unroll
    for (int u = 0; u < 4; ++u) pb[u] = (tok && f0 + u < F) ? prow[16 * full + u] : 0.f;
#pragma unroll
    for (int r = 0; r < MELT_PER_WAVE; ++r)
#pragma unroll
      for (int u = 0; u < 4; ++u) pa[r][u] = (aok[r] && f0 + u < F) ? arow[r][16 * full + u] : 0.f;
    step(pb, pa);
  }
  // D[row = 4 kk + i][col = n]: mel row 16 (mt0 + r) + 4 kk + i of frame t0 + n
  if (tok) {
#pragma unroll
    for (int r = 0; r < MELT_PER_WAVE; ++r)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int m = 16 * (mt0 + r) + 4 * kk + i;
        if (m < M) mel[(b * M + m) * T + t] = acc[r][i];
      }
  }
}

// ---------------------------------------------------------------------------------
// Welch: grid (nblk, B); workgroup loops over segments blockIdx.x, += gridDim.x and keeps
// per-bin partial sums in registers; a second kernel combines the partials in fixed order.
constexpr int WELCH_NT = 256;
constexpr int WELCH_MAXPAIR = (MAX_N / 4 + 1 + WELCH_NT - 1) / WELCH_NT;  // mirror pairs (k, M - k) per thread: 9

__global__ __launch_bounds__(WELCH_NT) void welch_partial_kernel(
    const float* __restrict__ x, int64_t L, int64_t ldx, int nperseg, int step, int nfft, int64_t nseg,
    const float* __restrict__ win, const float2* __restrict__ tw, int detrend, float* __restrict__ partial) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  __shared__ float red[2 * (WELCH_NT / 64)];
  const int M = nfft >> 1;
  float2* xa = reinterpret_cast<float2*>(lds);
  float2* xb = xa + M;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int64_t b = blockIdx.y;
  const float* xr = x + b * ldx;
  float acc[WELCH_MAXPAIR], accm[WELCH_MAXPAIR];
#pragma unroll
  for (int j = 0; j < WELCH_MAXPAIR; ++j) { acc[j] = 0.f; accm[j] = 0.f; }
  for (int64_t sg = blockIdx.x; sg < nseg; sg += gridDim.x) {
    const float* seg = xr + sg * (int64_t)step;
    // one pass over the segment: the samples stay in registers while the mean is reduced, then they are detrended,
    // windowed and packed (z[m] = x[2m] + i x[2m+1]) into LDS
    constexpr int PER = (MAX_N / 2 + WELCH_NT - 1) / WELCH_NT;       // packed points per thread (16 at n = 8192)
    float2 raw[PER];
    float s = 0.f, sj = 0.f;                                         // sum x, sum (i - (n-1)/2) x
    const float jc = 0.5f * (float)(nperseg - 1);
#pragma unroll
    for (int j = 0; j < PER; ++j) {
      const int m = tid + j * WELCH_NT;
      const int i0 = 2 * m, i1 = 2 * m + 1;
      raw[j].x = (m < M && i0 < nperseg) ? seg[i0] : 0.f;
      raw[j].y = (m < M && i1 < nperseg) ? seg[i1] : 0.f;
      s += raw[j].x + raw[j].y;
      if (detrend == 2) sj += ((float)i0 - jc) * raw[j].x + ((float)i1 - jc) * raw[j].y;
    }
    // detrend 1: subtract the mean; 2: subtract the least-squares line (scipy.signal.detrend type='linear'), written
    // around the segment centre so that slope and mean decouple: x - mean - slope (i - (n-1)/2),
    // slope = sum (i - c) x / sum (i - c)^2, sum (i - c)^2 = n (n^2 - 1) / 12
    float mean = 0.f, slope = 0.f;
    if (detrend) {
      s = wave_sum(s);
      sj = wave_sum(sj);
      if (lane == 0) { red[w] = s; red[WELCH_NT / 64 + w] = sj; }
      __syncthreads();
      s = 0.f; sj = 0.f;
#pragma unroll
      for (int i = 0; i < WELCH_NT / 64; ++i) { s += red[i]; sj += red[WELCH_NT / 64 + i]; }
      mean = s / (float)nperseg;
      if (detrend == 2 && nperseg > 1) {
        const double nn = (double)nperseg;
        slope = (float)((double)sj / (nn * (nn * nn - 1.0) / 12.0));
      }
    }
#pragma unroll
    for (int j = 0; j < PER; ++j) {
      const int m = tid + j * WELCH_NT;
      if (m < M) {
        const int i0 = 2 * m, i1 = 2 * m + 1;
        const float a = (i0 < nperseg) ? (raw[j].x - mean - slope * ((float)i0 - jc)) * win[i0] : 0.f;
        const float c = (i1 < nperseg) ? (raw[j].y - mean - slope * ((float)i1 - jc)) * win[i1] : 0.f;
        xa[m] = make_float2(a, c);
      }
    }
    __syncthreads();
    float2* Z = block_fft(xa, xb, M, tw + nfft, tid, WELCH_NT);
    // periodogram by mirror pairs: bins k and M - k come out of the same two points Z[k], Z[M-k] and one twiddle
    // W_{2M}^k:  X[k] = (E + w O)/2,  X[M-k] = conj(E - w O)/2,  E = Z[k] + conj(Z[M-k]),  O = -i (Z[k] - conj(Z[M-k]))
#pragma unroll
    for (int j = 0; j < WELCH_MAXPAIR; ++j) {
      const int k = tid + j * WELCH_NT;
      if (k <= (M >> 1)) {
        const float2 zk = Z[k], zm = Z[(M - k) & (M - 1)];
        const float2 E = make_float2(zk.x + zm.x, zk.y - zm.y);
        const float2 O = make_float2(zk.y + zm.y, zm.x - zk.x);
        const float2 wO = cmul(tw[k], O);
        const float ax = 0.5f * (E.x + wO.x), ay = 0.5f * (E.y + wO.y);
        const float bx = 0.5f * (E.x - wO.x), by = 0.5f * (E.y - wO.y);
        acc[j] += fmaf(ax, ax, ay * ay);                      // bin k
        accm[j] += fmaf(bx, bx, by * by);                     // bin M - k (the same bin when k = M/2)
      }
    }
    __syncthreads();
  }
  float* po = partial + ((int64_t)b * gridDim.x + blockIdx.x) * (int64_t)(M + 1);
#pragma unroll
  for (int j = 0; j < WELCH_MAXPAIR; ++j) {
    const int k = tid + j * WELCH_NT;
    if (k <= (M >> 1)) {
      po[k] = acc[j];
      if (M - k != k) po[M - k] = accm[j];
    }
  }
}

// Combine the per-workgroup partial sums in float64, fixed order: 64 bins x 16 strided sub-sums per workgroup
// (thread (k, g) adds partials g, g + 16, ...), then the 16 sub-sums of a bin are added in order through LDS.
constexpr int WF_G = 16;
__global__ __launch_bounds__(64 * WF_G) void welch_final_kernel(const float* __restrict__ partial, int nblk, int F,
                                                                 int64_t nseg, double scale, int odd_nfft,
                                                                 float* __restrict__ psd) {
  __shared__ double sub[WF_G][64];
  const int64_t b = blockIdx.y;
  const int kx = threadIdx.x & 63, g = threadIdx.x >> 6;
  const int k = blockIdx.x * 64 + kx;
  double s = 0.0;
  if (k < F)
    for (int j = g; j < nblk; j += WF_G) s += (double)partial[((int64_t)b * nblk + j) * F + k];
  sub[g][kx] = s;
  __syncthreads();
  if (g != 0 || k >= F) return;
  s = 0.0;
#pragma unroll
  for (int i = 0; i < WF_G; ++i) s += sub[i][kx];
  s = s * scale / (double)nseg;
  if (k > 0 && (odd_nfft || k < F - 1)) s *= 2.0;  // one-sided: double all but DC (and Nyquist)
  psd[b * (int64_t)F + k] = (float)s;
}


// Few streams (config C5: ONE per GPU) leave the kernel above with F / 64 = 33 workgroups for 2048 partial rows: 40 us.
// Two stages then: stage 1 cuts the partial rows into WF_SLICES contiguous slices (grid z), every slice sums its rows in
// order into float64 sub-results; stage 2 adds the slices in order and scales.  Same order of additions inside a slice,
// slices added first to last: deterministic.
constexpr int WF_SLICES = 16;
__global__ __launch_bounds__(64 * WF_G) void welch_slice_kernel(const float* __restrict__ partial, int nblk, int F,
                                                                 double* __restrict__ dsub) {
  __shared__ double sub[WF_G][64];
  const int64_t b = blockIdx.y;
  const int z = blockIdx.z;
  const int kx = threadIdx.x & 63, g = threadIdx.x >> 6;
  const int k = blockIdx.x * 64 + kx;
  const int per = (nblk + WF_SLICES - 1) / WF_SLICES;
  const int j0 = z * per, j1 = (j0 + per < nblk) ? j0 + per : nblk;
  double s = 0.0;
  if (k < F)
    for (int j = j0 + g; j < j1; j += WF_G) s += (double)partial[((int64_t)b * nblk + j) * F + k];
  sub[g][kx] = s;
  __syncthreads();
  if (g != 0 || k >= F) return;
  s = 0.0;
#pragma unroll
  for (int i = 0; i < WF_G; ++i) s += sub[i][kx];
  dsub[((int64_t)b * WF_SLICES + z) * F + k] = s;
}
__global__ __launch_bounds__(256) void welch_final2_kernel(const double* __restrict__ dsub, int F, int64_t nseg, double scale,
                                                           int odd_nfft, float* __restrict__ psd) {
  const int64_t b = blockIdx.y;
  const int k = blockIdx.x * 256 + threadIdx.x;
  if (k >= F) return;
  double s = 0.0;
#pragma unroll
  for (int z = 0; z < WF_SLICES; ++z) s += dsub[((int64_t)b * WF_SLICES + z) * F + k];
  s = s * scale / (double)nseg;
  if (k > 0 && (odd_nfft || k < F - 1)) s *= 2.0;
  psd[b * (int64_t)F + k] = (float)s;
}

// Strided variant used to compose large transforms (four-step) on the host side:
// element e of transform (o, b) lives at in[o*in_os + b*in_bs + e*in_es]; the output may be
// multiplied by the four-step twiddle W_bign^(b*k) (conjugated for the inverse).
__global__ void fft_pow2_strided_kernel(const float2* __restrict__ in, float2* __restrict__ out, int n, int inverse,
                                        const float2* __restrict__ tw, int64_t in_os, int64_t in_bs, int64_t in_es,
                                        int64_t out_os, int64_t out_bs, int64_t out_es, int64_t bign, float scale,
                                        int flags, int64_t mask_n, int64_t in_valid) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float2* x = reinterpret_cast<float2*>(lds);
  float2* y = x + n;
  const int tid = threadIdx.x, nt = blockDim.x;
  const int64_t b = blockIdx.x, o = blockIdx.y;
  const int64_t ibase = o * in_os, irow = b * in_bs, obase = o * out_os + b * out_bs;
  for (int i = tid; i < n; i += nt) {
    float2 v = fft_load(in, ibase, irow + (int64_t)i * in_es, flags, mask_n, in_valid);
    if (inverse) v.y = -v.y;
    x[i] = v;
  }
  __syncthreads();
  float2* r = block_fft(x, y, n, tw, tid, nt);
  for (int k = tid; k < n; k += nt) {
    float2 v = r[k];
    if (bign > 0) {
      const int64_t e = (b * (int64_t)k) % bign;
      double sn, cs;
      sincospi(-2.0 * (double)e / (double)bign, &sn, &cs);
      v = cmul(v, make_float2((float)cs, (float)sn));
    }
    v.x *= scale; v.y *= scale;
    if (inverse) v.y = -v.y;
    fft_store(out, obase + (int64_t)k * out_es, v, flags);
  }
}

// Column-tiled form of the strided transform for the two passes of a four-step FFT.  The transforms of one pass
// are the columns of a matrix whose rows are contiguous (in_bs == 1): a workgroup takes CB adjacent columns, so
// every global access is a run of CB complex values (128 bytes at CB = 16) instead of one 8-byte element per line,
// transposes them into LDS (one column = one natural-order array, pitch n + COLS_PAD), runs the CB transforms side
// by side (256 / CB threads each; the trip counts and barriers of block_fft depend on n only) and stores either
// k-fast (out_es == 1: the transposed layout pass A leaves for pass B) or column-fast (out_bs == 1: final order).
constexpr int COLS_NT = 256;
constexpr int COLS_PAD = 2;
constexpr int COLS_MAXN = 1024;

__device__ __forceinline__ float2 four_step_twiddle(int64_t e, int64_t bign) {
  if (bign <= (1 << 24)) {                       // e < bign: exact in float, and bign is a power of two
    float sn, cs;
    sincospif(-2.0f * ((float)e / (float)bign), &sn, &cs);
    return make_float2(cs, sn);
  }
  double sn, cs;
  sincospi(-2.0 * (double)e / (double)bign, &sn, &cs);
  return make_float2((float)cs, (float)sn);
}

template <bool KFAST>
__global__ __launch_bounds__(COLS_NT) void fft_cols_kernel(const float2* __restrict__ in, float2* __restrict__ out,
                                                           int n, int cb_log, int inverse,
                                                           const float2* __restrict__ tw, int64_t in_os, int64_t in_es,
                                                           int64_t out_os, int64_t out_bs, int64_t out_es,
                                                           int64_t bign, float scale, int flags, int64_t mask_n, int64_t in_valid) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int CB = 1 << cb_log, LP = n + COLS_PAD;
  float2* x = reinterpret_cast<float2*>(lds);
  float2* y = x + CB * LP;
  const int tid = threadIdx.x;
  const int64_t c0 = (int64_t)blockIdx.x << cb_log, o = blockIdx.y;
  const int64_t ibase = o * in_os, obase = o * out_os;
  const int total = n << cb_log;
  for (int idx = tid; idx < total; idx += COLS_NT) {
    const int c = idx & (CB - 1), e = idx >> cb_log;
    const int64_t pos = (int64_t)e * in_es + c0 + c;           // position inside the row (= the bin, for the analytic weights)
    float2 v = fft_load(in, ibase, pos, flags, mask_n, in_valid);
    if (inverse) v.y = -v.y;
    x[c * LP + e] = v;
  }
  __syncthreads();
  const int tpc_log = 8 - cb_log;                              // threads per column
  const int g = tid >> tpc_log, lt = tid & ((1 << tpc_log) - 1);
  const float2* r = block_fft(x + g * LP, y + g * LP, n, tw, lt, 1 << tpc_log) - g * LP;
  int ln = 0;
  while ((1 << ln) < n) ++ln;
  for (int idx = tid; idx < total; idx += COLS_NT) {
    int c, k;
    if (KFAST) { k = idx & (n - 1); c = idx >> ln; }
    else { c = idx & (CB - 1); k = idx >> cb_log; }
    float2 v = r[c * LP + k];
    if (bign > 0) {
      int64_t e = (c0 + c) * (int64_t)k;
      if (e >= bign) e %= bign;                                // (column * k < bign in a four-step split: never taken there)
      v = cmul(v, four_step_twiddle(e, bign));
    }
    v.x *= scale; v.y *= scale;
    if (inverse) v.y = -v.y;
    fft_store(out, obase + (c0 + c) * out_bs + (int64_t)k * out_es, v, flags);
  }
}

// out[i] = a[i] * b[i mod nb]   (conj_b: multiply by conj(b))
__global__ void cmul_kernel(const float2* __restrict__ a, const float2* __restrict__ b, float2* __restrict__ out,
                            int64_t na, int64_t nb, int conj_b) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < na; i += (int64_t)gridDim.x * blockDim.x) {
    const float2 u = a[i], v = b[i % nb];
    out[i] = conj_b ? cmulc(u, v) : cmul(u, v);
  }
}

// out[r, i] = (i < len ? x[r, i] * (win ? win[i] : 1) : 0, 0) for i < n : real -> zero-padded complex rows
__global__ void pack_real_kernel(const float* __restrict__ x, int64_t len, int64_t ldx, const float* __restrict__ win,
                                 float2* __restrict__ out, int64_t n) {
  const int64_t r = blockIdx.y;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    float v = 0.f;
    if (i < len) v = win ? x[r * ldx + i] * win[i] : x[r * ldx + i];
    out[r * n + i] = make_float2(v, 0.f);
  }
}

bool is_pow2(int n) { return n >= 2 && (n & (n - 1)) == 0; }
int fft_threads(int n) { int t = n / 4; if (t < 64) t = 64; if (t > 1024) t = 1024; return t; }

int set_lds(const void* fn, size_t bytes, const char* what) {
  if (bytes > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) {
      set_error("%s: cannot reserve %zu B of LDS: %s", what, bytes, hipGetErrorString(e));
      return SYG_E_LAUNCH;
    }
  }
  return SYG_OK;
}

}  // namespace
int welch_wave_launch(const float* x, int64_t B, int64_t ldx, int step, int64_t nseg, const float* window,
                      const float* twiddle, int detrend, int nblk, float* work, hipStream_t st);   // welch_wave.hip
namespace {
// Partial sums per stream: enough of them to fill the chip whatever the batch (a single one-hour stream -- config C5 --
// gets 2048: the eight waves per CU the wave-per-segment kernel's registers admit; a batch of 1024 clips 16 each); partial j sums the
// segments j, j + nblk, ...
__host__ int welch_nblk(int64_t B) {
  int64_t n = (8192 / B) & ~(int64_t)3;       // (a multiple of 4: the wave-per-segment kernel runs 4 waves per workgroup)
  if (n < 16) n = 16;
  if (n > 2048) n = 2048;
  return (int)n;
}

}  // namespace
}  // namespace syg

using namespace syg;

extern "C" int syg_fft_pow2_c2c_f32(const float* in, float* out, int64_t batch, int n, int inverse,
                                    const float* twiddle, void* stream) {
  SYG_REQUIRE(in && out && twiddle, "fft_pow2: null pointer argument");
  SYG_REQUIRE(is_pow2(n) && n <= MAX_N, "fft_pow2: n must be a power of two in [2, %d] (got %d)", MAX_N, n);
  SYG_REQUIRE(batch >= 1 && batch < (int64_t)0x7fffffff, "fft_pow2: bad batch %lld", (long long)batch);
  const size_t lds = (size_t)n * 2 * sizeof(float2);
  int rc = set_lds((const void*)fft_pow2_kernel, lds, "fft_pow2");
  if (rc) return rc;
  hipLaunchKernelGGL(fft_pow2_kernel, dim3((unsigned)batch), dim3(fft_threads(n)), lds, (hipStream_t)stream,
                     (const float2*)in, (float2*)out, n, inverse, (const float2*)twiddle);
  SYG_CHECK_LAUNCH("fft_pow2");
  return SYG_OK;
}

extern "C" int syg_stft_pow2_c2c_f32(const float* y, int64_t B, int64_t L, int64_t ldy, int n_fft, int hop,
                                     int center, int64_t T, const float* window, const float* twiddle, float* out,
                                     void* stream) {
  SYG_REQUIRE(y && window && twiddle && out, "stft_pow2: null pointer argument");
  SYG_REQUIRE(is_pow2(n_fft) && n_fft >= 8 && n_fft <= 2 * MAX_N,
              "stft_pow2: n_fft must be a power of two in [8, %d] (got %d)", 2 * MAX_N, n_fft);
  SYG_REQUIRE(B >= 1 && L >= 1 && ldy >= L && hop >= 1, "stft_pow2: bad B/L/ldy/hop");
  const int64_t Texp = center ? 1 + L / hop : (L >= n_fft ? 1 + (L - n_fft) / hop : 0);
  SYG_REQUIRE(T >= 1 && T == Texp, "stft_pow2: T=%lld does not match the framing rule (%lld)", (long long)T,
              (long long)Texp);
  SYG_REQUIRE(B * T < (int64_t)0x7fffffff, "stft_pow2: grid too large");
  const int M = n_fft / 2;
  const size_t lds = (size_t)M * 2 * sizeof(float2);
  int rc = set_lds((const void*)stft_pow2_kernel, lds, "stft_pow2");
  if (rc) return rc;
  hipLaunchKernelGGL(stft_pow2_kernel, dim3((unsigned)(B * T)), dim3(fft_threads(M)), lds, (hipStream_t)stream, y, L,
                     ldy, n_fft, hop, center ? n_fft / 2 : 0, T, window, (const float2*)twiddle, (float2*)out);
  SYG_CHECK_LAUNCH("stft_pow2");
  return SYG_OK;
}

extern "C" int syg_cabs_pow_f32(const float* x_c64, int64_t n, int power, float* out, void* stream) {
  SYG_REQUIRE(x_c64 && out, "cabs_pow: null pointer argument");
  SYG_REQUIRE(n >= 0 && (power == 1 || power == 2), "cabs_pow: power must be 1 or 2");
  if (n == 0) return SYG_OK;
  int64_t blocks = (n + 255) / 256;
  if (blocks > 2048 * 8) blocks = 2048 * 8;
  hipLaunchKernelGGL(cabs_pow_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                     (const float2*)x_c64, n, power, out);
  SYG_CHECK_LAUNCH("cabs_pow");
  return SYG_OK;
}

extern "C" int syg_mel_dense_f32(const float* P, int64_t B, int64_t T, int F, const float* basis, int M,
                                 float* mel_out, void* stream) {
  SYG_REQUIRE(P && basis && mel_out, "mel_dense: null pointer argument");
  SYG_REQUIRE(B >= 1 && T >= 1 && F >= 1 && F <= 16385 && M >= 1, "mel_dense: bad shape");
  const int64_t tiles_per_clip = (T + 15) / 16;
  const int64_t waves = B * tiles_per_clip;
  SYG_REQUIRE((waves + 3) / 4 < (int64_t)0x7fffffff && M <= 16 * MELT_PER_WAVE * 65535, "mel_dense: grid too large");
  const dim3 grid((unsigned)((waves + 3) / 4), (unsigned)((M + 16 * MELT_PER_WAVE - 1) / (16 * MELT_PER_WAVE)));
  hipLaunchKernelGGL(mel_dense_mfma_kernel, grid, dim3(256), 0, (hipStream_t)stream, P, B, T, F, basis, M, mel_out,
                     tiles_per_clip);
  SYG_CHECK_LAUNCH("mel_dense");
  return SYG_OK;
}

extern "C" int64_t syg_welch_work_bytes(int64_t B, int nfft) {
  if (B < 1 || nfft < 2) return -1;
  // float partial rows, then (8-byte aligned) the float64 slice sums of the two-stage combine
  const int64_t part = (B * (int64_t)welch_nblk(B) * (nfft / 2 + 1) * (int64_t)sizeof(float) + 7) & ~(int64_t)7;
  return part + B * WF_SLICES * (int64_t)(nfft / 2 + 1) * (int64_t)sizeof(double);
}

// combine the partial rows: one stage when the streams alone fill the chip, two stages otherwise
static int welch_combine(const float* work, int64_t B, int nblk, int F, int64_t nseg, double scale, float* psd_out,
                         hipStream_t st) {
  if (B * ((F + 63) / 64) >= 512 || nblk < 4 * WF_SLICES) {
    hipLaunchKernelGGL(welch_final_kernel, dim3((F + 63) / 64, (unsigned)B), dim3(64 * WF_G), 0, st, work, nblk, F, nseg, scale,
                       0, psd_out);
    SYG_CHECK_LAUNCH("welch_final");
    return SYG_OK;
  }
  const int64_t part = (B * (int64_t)welch_nblk(B) * F * (int64_t)sizeof(float) + 7) & ~(int64_t)7;
  double* dsub = reinterpret_cast<double*>(reinterpret_cast<char*>(const_cast<float*>(work)) + part);
  hipLaunchKernelGGL(welch_slice_kernel, dim3((F + 63) / 64, (unsigned)B, WF_SLICES), dim3(64 * WF_G), 0, st, work, nblk, F, dsub);
  SYG_CHECK_LAUNCH("welch_slice");
  hipLaunchKernelGGL(welch_final2_kernel, dim3((F + 255) / 256, (unsigned)B), dim3(256), 0, st, (const double*)dsub, F, nseg,
                     scale, 0, psd_out);
  SYG_CHECK_LAUNCH("welch_final2");
  return SYG_OK;
}

extern "C" int syg_welch_f32(const float* x, int64_t B, int64_t L, int64_t ldx, int nperseg, int step, int nfft,
                             const float* window, const float* twiddle, int detrend, double scale, float* psd_out,
                             void* work, void* stream) {
  SYG_REQUIRE(x && window && twiddle && psd_out && work, "welch: null pointer argument");
  SYG_REQUIRE(is_pow2(nfft) && nfft >= 8 && nfft <= 2 * MAX_N, "welch: nfft must be a power of two in [8, %d]",
              2 * MAX_N);
  SYG_REQUIRE(nperseg >= 1 && nperseg <= nfft && step >= 1 && step <= nperseg, "welch: bad nperseg/step");
  SYG_REQUIRE(B >= 1 && B <= 65535 && L >= nperseg && ldx >= L, "welch: bad B/L/ldx");
  SYG_REQUIRE(detrend >= 0 && detrend <= 2, "welch: detrend must be 0 (none), 1 (constant) or 2 (linear)");
  const int64_t nseg = (L - (nperseg - step)) / step;
  SYG_REQUIRE(nseg >= 1, "welch: no complete segment");
  const int M = nfft / 2, F = M + 1;
  const size_t lds = (size_t)M * 2 * sizeof(float2);
  int rc = set_lds((const void*)welch_partial_kernel, lds, "welch");
  if (rc) return rc;
  hipStream_t st = (hipStream_t)stream;
  int nblk = welch_nblk(B);
  // nperseg = nfft = 4096 on 16-byte aligned segments (config C5): one wave per segment (welch_wave.hip); its partial
  // sums carry a factor 4
  if (nfft == 4096 && nperseg == 4096 && step % 4 == 0 && ldx % 4 == 0 && ((uintptr_t)x) % 16 == 0) {
    if ((int64_t)nblk > nseg) nblk = (int)((nseg + 3) & ~(int64_t)3);
    rc = welch_wave_launch(x, B, ldx, step, nseg, window, twiddle, detrend, nblk, (float*)work, st);
    if (rc) return rc;
    return welch_combine((const float*)work, B, nblk, F, nseg, 0.25 * scale, psd_out, st);
  }
  if ((int64_t)nblk > nseg) nblk = (int)nseg;
  hipLaunchKernelGGL(welch_partial_kernel, dim3(nblk, (unsigned)B), dim3(WELCH_NT), lds, st, x, L, ldx, nperseg,
                     step, nfft, nseg, window, (const float2*)twiddle, detrend, (float*)work);
  SYG_CHECK_LAUNCH("welch_partial");
  return welch_combine((const float*)work, B, nblk, F, nseg, scale, psd_out, st);
}

extern "C" int syg_fft_pow2_strided_ex_f32(const float* in, float* out, int64_t outer, int64_t batch, int n,
                                            int inverse, const float* twiddle, int64_t in_os, int64_t in_bs,
                                            int64_t in_es, int64_t out_os, int64_t out_bs, int64_t out_es,
                                            int64_t bign, float scale, int flags, int64_t mask_n, int64_t in_valid, void* stream) {
  SYG_REQUIRE(in && out && twiddle, "fft_pow2_strided: null pointer argument");
  SYG_REQUIRE(is_pow2(n) && n <= MAX_N, "fft_pow2_strided: n must be a power of two in [2, %d] (got %d)", MAX_N, n);
  SYG_REQUIRE(batch >= 1 && batch < (int64_t)0x7fffffff && outer >= 1 && outer <= 65535,
              "fft_pow2_strided: bad batch/outer");
  SYG_REQUIRE(in != out, "fft_pow2_strided: in-place operation is not supported");
  SYG_REQUIRE(flags >= 0 && flags <= 7 && (flags & 5) != 5 && mask_n >= 0 && in_valid >= 0, "fft_pow2_strided: bad flags / mask length");
  if (in_bs == 1 && (out_es == 1 || out_bs == 1) && n <= COLS_MAXN && n >= 8) {
    int cb_log = 4;                                            // 16 columns = 128-byte runs
    while (cb_log > 2 && ((int64_t)n << cb_log) > 4096) --cb_log;
    // (two workgroups per CU hide too little: above 40 KB of LDS take 8 columns, see fft_mixed.hip)
    if (cb_log == 4 && (size_t)2 * ((size_t)(n + COLS_PAD) << 4) * sizeof(float2) > 40 * 1024) cb_log = 3;
    if (batch % (1 << cb_log) == 0) {
      const bool kfast = out_es == 1;
      const void* fn = kfast ? (const void*)fft_cols_kernel<true> : (const void*)fft_cols_kernel<false>;
      const size_t lds = (size_t)2 * ((size_t)(n + COLS_PAD) << cb_log) * sizeof(float2);
      int rc = set_lds(fn, lds, "fft_pow2_strided(cols)");
      if (rc) return rc;
      const dim3 grid((unsigned)(batch >> cb_log), (unsigned)outer);
      if (kfast)
        hipLaunchKernelGGL(fft_cols_kernel<true>, grid, dim3(COLS_NT), lds, (hipStream_t)stream, (const float2*)in,
                           (float2*)out, n, cb_log, inverse, (const float2*)twiddle, in_os, in_es, out_os, out_bs,
                           out_es, bign, scale, flags, mask_n, in_valid);
      else
        hipLaunchKernelGGL(fft_cols_kernel<false>, grid, dim3(COLS_NT), lds, (hipStream_t)stream, (const float2*)in,
                           (float2*)out, n, cb_log, inverse, (const float2*)twiddle, in_os, in_es, out_os, out_bs,
                           out_es, bign, scale, flags, mask_n, in_valid);
      SYG_CHECK_LAUNCH("fft_pow2_strided(cols)");
      return SYG_OK;
    }
  }
  const size_t lds = (size_t)n * 2 * sizeof(float2);
  int rc = set_lds((const void*)fft_pow2_strided_kernel, lds, "fft_pow2_strided");
  if (rc) return rc;
  hipLaunchKernelGGL(fft_pow2_strided_kernel, dim3((unsigned)batch, (unsigned)outer), dim3(fft_threads(n)), lds,
                     (hipStream_t)stream, (const float2*)in, (float2*)out, n, inverse, (const float2*)twiddle, in_os,
                     in_bs, in_es, out_os, out_bs, out_es, bign, scale, flags, mask_n, in_valid);
  SYG_CHECK_LAUNCH("fft_pow2_strided");
  return SYG_OK;
}

extern "C" int syg_fft_pow2_strided_c2c_f32(const float* in, float* out, int64_t outer, int64_t batch, int n, int inverse,
                                            const float* twiddle, int64_t in_os, int64_t in_bs, int64_t in_es,
                                            int64_t out_os, int64_t out_bs, int64_t out_es, int64_t bign, float scale,
                                            void* stream) {
  return syg_fft_pow2_strided_ex_f32(in, out, outer, batch, n, inverse, twiddle, in_os, in_bs, in_es, out_os, out_bs, out_es,
                                     bign, scale, 0, 0, 0, stream);
}

extern "C" int syg_cmul_c64(const float* a, const float* b, float* out, int64_t na, int64_t nb, int conj_b,
                            void* stream) {
  SYG_REQUIRE(a && b && out, "cmul: null pointer argument");
  SYG_REQUIRE(na >= 1 && nb >= 1, "cmul: bad sizes");
  int64_t blocks = (na + 255) / 256;
  if (blocks > 16384) blocks = 16384;
  hipLaunchKernelGGL(cmul_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const float2*)a,
                     (const float2*)b, (float2*)out, na, nb, conj_b);
  SYG_CHECK_LAUNCH("cmul");
  return SYG_OK;
}

extern "C" int syg_pack_real_c64(const float* x, int64_t rows, int64_t len, int64_t ldx, const float* window,
                                 float* out, int64_t n, void* stream) {
  SYG_REQUIRE(x && out, "pack_real: null pointer argument");
  SYG_REQUIRE(rows >= 1 && rows <= 65535 && len >= 0 && n >= 1 && ldx >= len, "pack_real: bad sizes");
  int64_t blocks = (n + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(pack_real_kernel, dim3((unsigned)blocks, (unsigned)rows), dim3(256), 0, (hipStream_t)stream, x,
                     len < n ? len : n, ldx, window, (float2*)out, n);
  SYG_CHECK_LAUNCH("pack_real");
  return SYG_OK;
}
