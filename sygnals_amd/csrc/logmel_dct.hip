// power_to_db (per-clip ref=max, amin, top_db clamp) + DCT-II (+ lifter).
//
// Reproduces librosa.power_to_db(S_mel, ref=np.max) (manager.py:223) and
// librosa.feature.mfcc(S=..) -> scipy.fft.dct(type=2, norm='ortho')[:n_mfcc] (cepstral.py:106-115).
// One workgroup per clip.  The clip's [M, T] mel matrix (15 KB at the headline config) is
// L2-resident from the producing kernel; the DCT is the 16x16x4 f32 MFMA with the DCT rows
// as the A operand and 16 frames as the B operand.
#include "common.h"

namespace syg {
namespace {

constexpr int NT = 256;
typedef float v4f __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(NT) void logmel_dct_kernel(float* __restrict__ mel, int M, int64_t T,
                                                         const float* __restrict__ dct, int K,
                                                         const float* __restrict__ lifter, float amin, float top_db,
                                                         int ref_is_max, float ref_value, float* __restrict__ logmel,
                                                         float* __restrict__ mfcc) {
  __shared__ float red[NT / 64];
  __shared__ float s_floor, s_refdb;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int64_t b = blockIdx.x;
  float* src = mel + b * (int64_t)M * T;
  float* dst = logmel ? logmel + b * (int64_t)M * T : src;
  const int ktiles = (K + 15) / 16;
  const int ttiles = (int)((T + 15) / 16);
  const int f = lane & 15, g = lane >> 4;
  const int64_t n = (int64_t)M * T;

  if (ref_is_max == 2) goto dct_stage;  // input is already in dB: DCT only
  {
  // ---- per-clip maximum (ref=np.max and the top_db floor both need it)
  float mx = 0.f;  // power is non-negative
  for (int64_t i = tid; i < n; i += NT) mx = fmaxf(mx, src[i]);
  mx = wave_max(mx);
  if (lane == 0) red[w] = mx;
  __syncthreads();
  if (tid == 0) {
    float m = red[0];
    for (int i = 1; i < NT / 64; ++i) m = fmaxf(m, red[i]);
    const float ref = ref_is_max ? m : fabsf(ref_value);
    // dB values are formed as 10 log10(2) (log2 x - log2 ref): the difference of logs is exact (0) when
    // x == ref, whatever the compiler contracts into FMAs (hardware log2: common.h)
    const float reflog = syg_log2(fmaxf(amin, ref));
    s_refdb = reflog;
    // log_spec.max() - top_db, with log_spec monotone in the power
    s_floor = (top_db >= 0.f) ? SYG_DB_PER_LOG2 * (syg_log2(fmaxf(amin, m)) - reflog) - top_db : -3.4e38f;
  }
  __syncthreads();
  const float reflog = s_refdb, flo = s_floor;
  for (int64_t i = tid; i < n; i += NT) {
    float v = SYG_DB_PER_LOG2 * (syg_log2(fmaxf(amin, src[i])) - reflog);
    dst[i] = fmaxf(v, flo);
  }
  }
  if (mfcc == nullptr) return;
  __threadfence_block();
  __syncthreads();
dct_stage:
  if (ref_is_max == 2) dst = src;

  // ---- DCT: out[k, t] = sum_m dct[k, m] * dB[m, t]   (16x16 output tiles on the MFMA)
  for (int tile = w; tile < ktiles * ttiles; tile += NT / 64) {
    const int kt = tile / ttiles, tt = tile % ttiles;
    const int krow = kt * 16 + f;          // A operand row
    const int64_t tcol = (int64_t)tt * 16 + f;  // B operand column
    v4f acc = {0.f, 0.f, 0.f, 0.f};
    for (int m0 = 0; m0 < M; m0 += 4) {
      const int m = m0 + g;
      const float a = (krow < K && m < M) ? dct[krow * M + m] : 0.f;
      const float bv = (tcol < T && m < M) ? dst[(int64_t)m * T + tcol] : 0.f;
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bv, acc, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int k = kt * 16 + 4 * g + r;
      if (k < K && tcol < T) {
        float v = acc[r];
        if (lifter) v *= lifter[k];
        mfcc[(b * K + k) * T + tcol] = v;
      }
    }
  }
}

// The same with the clip's matrix staged in LDS (M * T * 4 bytes <= 144 KiB: every clip of the BASELINE configurations and of
// the other frame lengths' tile kernels): one read of the mel matrix, the dB matrix written once (the in-place /
// logmel_out contract) and read back from LDS by the DCT, whose rows wait in registers; 512 threads.  The form above read
// the matrix from global memory three times, one dependent L2 round trip per DCT step (32 us per 1024 clips at T = 188).
// STORE_DB = false (syg_mel_mfcc_f32): the dB matrix only exists in LDS -- callers that want the MFCCs alone save its
// M * T * 4 bytes of HBM writes per clip (at T = 751 frames three times the bytes of the MFCCs).
constexpr int NTL = 512;
template <bool STORE_DB>
__global__ __launch_bounds__(NTL) void logmel_dct_lds_kernel(float* __restrict__ mel, int M, int64_t T,
                                                             const float* __restrict__ dct, int K,
                                                             const float* __restrict__ lifter, float amin, float top_db,
                                                             int ref_is_max, float ref_value, float* __restrict__ logmel,
                                                             float* __restrict__ mfcc) {
  extern __shared__ __attribute__((aligned(16))) float db[];     // [M][T]
  __shared__ float red[NTL / 64];
  __shared__ float s_floor, s_refdb;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int64_t b = blockIdx.x;
  float* src = mel + b * (int64_t)M * T;
  float* dst = logmel ? logmel + b * (int64_t)M * T : src;
  const int n = M * (int)T;
  float mx = 0.f;  // power is non-negative
  if ((n & 3) == 0 && ((uintptr_t)src & 15) == 0) {
    for (int i = tid; i < (n >> 2); i += NTL) {
      const float4 q = reinterpret_cast<const float4*>(src)[i];
      reinterpret_cast<float4*>(db)[i] = q;
      mx = fmaxf(fmaxf(mx, fmaxf(q.x, q.y)), fmaxf(q.z, q.w));
    }
  } else {
    for (int i = tid; i < n; i += NTL) { const float q = src[i]; db[i] = q; mx = fmaxf(mx, q); }
  }
  mx = wave_max(mx);
  if (lane == 0) red[w] = mx;
  __syncthreads();
  if (tid == 0) {
    float m = red[0];
    for (int i = 1; i < NTL / 64; ++i) m = fmaxf(m, red[i]);
    const float ref = ref_is_max ? m : fabsf(ref_value);
    const float reflog = syg_log2(fmaxf(amin, ref));
    s_refdb = reflog;
    s_floor = (top_db >= 0.f) ? SYG_DB_PER_LOG2 * (syg_log2(fmaxf(amin, m)) - reflog) - top_db : -3.4e38f;
  }
  __syncthreads();
  const float reflog = s_refdb, flo = s_floor;
  if ((n & 3) == 0 && ((uintptr_t)dst & 15) == 0) {
    for (int i = tid; i < (n >> 2); i += NTL) {
      float4 q = reinterpret_cast<float4*>(db)[i];
      q.x = fmaxf(SYG_DB_PER_LOG2 * (syg_log2(fmaxf(amin, q.x)) - reflog), flo);
      q.y = fmaxf(SYG_DB_PER_LOG2 * (syg_log2(fmaxf(amin, q.y)) - reflog), flo);
      q.z = fmaxf(SYG_DB_PER_LOG2 * (syg_log2(fmaxf(amin, q.z)) - reflog), flo);
      q.w = fmaxf(SYG_DB_PER_LOG2 * (syg_log2(fmaxf(amin, q.w)) - reflog), flo);
      reinterpret_cast<float4*>(db)[i] = q;
      if (STORE_DB) reinterpret_cast<float4*>(dst)[i] = q;
    }
  } else {
    for (int i = tid; i < n; i += NTL) {
      const float v = fmaxf(SYG_DB_PER_LOG2 * (syg_log2(fmaxf(amin, db[i])) - reflog), flo);
      db[i] = v;
      if (STORE_DB) dst[i] = v;
    }
  }
  if (mfcc == nullptr) return;
  __syncthreads();
  // ---- DCT: out[k, t] = sum_m dct[k, m] * dB[m, t]   (16 x 16 output tiles on the MFMA, the A operands of a k tile
  // loaded once per wave: M / 4 registers)
  const int ktiles = (K + 15) / 16, ttiles = (int)((T + 15) / 16);
  const int f = lane & 15, g = lane >> 4;
  constexpr int MAXS = 32;                            // k-steps of four mel rows kept in registers: M <= 128
  const int steps = (M + 3) >> 2;
  for (int kt = 0; kt < ktiles; ++kt) {
    const int krow = kt * 16 + f;
    float a[MAXS];
#pragma unroll
    for (int sidx = 0; sidx < MAXS; ++sidx) {
      const int m = 4 * sidx + g;
      a[sidx] = (sidx < steps && krow < K && m < M) ? dct[krow * M + m] : 0.f;
    }
    for (int tt = w; tt < ttiles; tt += NTL / 64) {
      const int tcol = tt * 16 + f;
      v4f acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int sidx = 0; sidx < MAXS; ++sidx) {
        if (sidx < steps) {
          const int m = 4 * sidx + g;
          const float bv = (tcol < T && m < M) ? db[m * (int)T + tcol] : 0.f;
          acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[sidx], bv, acc, 0, 0, 0);
        }
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int k = kt * 16 + 4 * g + r;
        if (k < K && tcol < T) {
          float v = acc[r];
          if (lifter) v *= lifter[k];
          mfcc[(b * K + k) * T + tcol] = v;
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Config C4's per-clip block in one launch behind the fused STFT kernel: from the clip's mel power [M, T], the
// statistics rows and the contrast tail means, write [K + 2 + R, T]:
//   rows 0 .. K-1        MFCC: power_to_db(ref = max, amin, top_db) -> DCT rows on the matrix cores (the dB matrix only
//                        ever exists in LDS)
//   row  K, K + 1        spectral centroid (Hz) and rolloff (bin index x bin width = the bin's frequency)
//   rows K + 2 .. +R-1   spectral contrast: power_to_db(peak) - power_to_db(valley), each clamped top_db below the
//                        maximum of its own [R, T] matrix (librosa.feature.spectral_contrast)
// -- the columns extract_features(["mfcc", "spectral_centroid", "spectral_rolloff", "spectral_contrast"]) returns
// (manager.py:289-371).  One workgroup per clip.
__global__ __launch_bounds__(NT) void feature_block_kernel(const float* __restrict__ mel, int M, int64_t T,
                                                           const float* __restrict__ dct, int K, float amin,
                                                           float top_db, const float* __restrict__ stats, float binhz,
                                                           const float* __restrict__ pv, int R, float c_amin,
                                                           float c_top_db, float* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) float db[];     // [M][T]
  __shared__ float red[3][NT / 64];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int64_t b = blockIdx.x;
  const int rows = K + 2 + R;
  const float* src = mel + b * (int64_t)M * T;
  const float* pk = pv + (b * 2 + 0) * (int64_t)R * T;
  const float* vl = pv + (b * 2 + 1) * (int64_t)R * T;
  float* ob = out + b * (int64_t)rows * T;
  const int64_t n = (int64_t)M * T, nc = (int64_t)R * T;
  float mx = 0.f, m0 = 0.f, m1 = 0.f;       // powers and magnitudes are non-negative
  const bool have_mel = mel != nullptr;     // (null: rows 0 .. K-1 were written by syg_stft2048_features_tri_f32)
  if (have_mel)
    for (int64_t i = tid; i < n; i += NT) mx = fmaxf(mx, src[i]);
  for (int64_t i = tid; i < nc; i += NT) { m0 = fmaxf(m0, pk[i]); m1 = fmaxf(m1, vl[i]); }
  mx = wave_max(mx); m0 = wave_max(m0); m1 = wave_max(m1);
  if (lane == 0) { red[0][w] = mx; red[1][w] = m0; red[2][w] = m1; }
  __syncthreads();
  mx = red[0][0]; m0 = red[1][0]; m1 = red[2][0];
#pragma unroll
  for (int i = 1; i < NT / 64; ++i) { mx = fmaxf(mx, red[0][i]); m0 = fmaxf(m0, red[1][i]); m1 = fmaxf(m1, red[2][i]); }
  // (the same expressions as logmel_dct_kernel / contrast_db_kernel: the two forms give identical values)
  const float reflog = syg_log2(fmaxf(amin, mx));
  const float flo = (top_db >= 0.f) ? SYG_DB_PER_LOG2 * (syg_log2(fmaxf(amin, mx)) - reflog) - top_db : -3.4e38f;
  if (have_mel)
    for (int64_t i = tid; i < n; i += NT) db[i] = fmaxf(SYG_DB_PER_LOG2 * (syg_log2(fmaxf(amin, src[i])) - reflog), flo);
  const float f0 = (c_top_db >= 0.f) ? 10.f * log10f(fmaxf(c_amin, m0)) - c_top_db : -3.4e38f;
  const float f1 = (c_top_db >= 0.f) ? 10.f * log10f(fmaxf(c_amin, m1)) - c_top_db : -3.4e38f;
  for (int64_t i = tid; i < nc; i += NT) {
    const float a = fmaxf(10.f * log10f(fmaxf(c_amin, pk[i])), f0);
    const float c = fmaxf(10.f * log10f(fmaxf(c_amin, vl[i])), f1);
    ob[(int64_t)(K + 2) * T + i] = a - c;
  }
  const float* st = stats + b * (int64_t)SYG_NSTAT * T;
  for (int64_t t = tid; t < T; t += NT) {
    ob[(int64_t)K * T + t] = st[SYG_STAT_CENTROID * T + t];
    ob[(int64_t)(K + 1) * T + t] = st[SYG_STAT_ROLLOFF_BIN * T + t] * binhz;
  }
  if (!have_mel) return;
  __syncthreads();
  const int ktiles = (K + 15) / 16, ttiles = (int)((T + 15) / 16);
  const int f = lane & 15, g = lane >> 4;
  for (int tile = w; tile < ktiles * ttiles; tile += NT / 64) {
    const int kt = tile / ttiles, tt = tile % ttiles;
    const int krow = kt * 16 + f;
    const int64_t tcol = (int64_t)tt * 16 + f;
    v4f acc = {0.f, 0.f, 0.f, 0.f};
    for (int m0i = 0; m0i < M; m0i += 4) {
      const int m = m0i + g;
      const float a = (krow < K && m < M) ? dct[krow * M + m] : 0.f;
      const float bv = (tcol < T && m < M) ? db[(int64_t)m * T + tcol] : 0.f;
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bv, acc, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int k = kt * 16 + 4 * g + r;
      if (k < K && tcol < T) ob[(int64_t)k * T + tcol] = acc[r];
    }
  }
}

}  // namespace
}  // namespace syg

using namespace syg;

extern "C" int syg_feature_block_f32(const float* mel, int64_t B, int M, int64_t T, const float* dct, int K, float amin,
                                     float top_db, const float* stats, float binhz, const float* contrast_pv, int R,
                                     float c_amin, float c_top_db, float* block_out, void* stream) {
  SYG_REQUIRE((dct || !mel) && stats && contrast_pv && block_out, "feature_block: null pointer argument");
  SYG_REQUIRE(B >= 1 && B < (int64_t)0x7fffffff && M >= 1 && T >= 1 && K >= 1 && K <= M && R >= 1,
              "feature_block: bad shape (B=%lld M=%d T=%lld K=%d R=%d)", (long long)B, M, (long long)T, K, R);
  SYG_REQUIRE(amin >= 1.17549435e-38f && c_amin > 0.f, "feature_block: amin must be a positive normal float");
  const size_t lds = mel ? (size_t)M * (size_t)T * sizeof(float) : 0;       // (mel == NULL: only the statistics / contrast rows)
  SYG_REQUIRE(lds <= 150 * 1024, "feature_block: the clip's dB matrix (%d x %lld) does not fit LDS; use "
              "syg_logmel_dct_f32 + syg_contrast_db_f32", M, (long long)T);
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute((const void*)feature_block_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)lds);
    if (e != hipSuccess) { set_error("feature_block: cannot reserve %zu B of LDS", lds); return SYG_E_LAUNCH; }
  }
  hipLaunchKernelGGL(feature_block_kernel, dim3((unsigned)B), dim3(NT), lds, (hipStream_t)stream, mel, M, T, dct, K, amin,
                     top_db, stats, binhz, contrast_pv, R, c_amin, c_top_db, block_out);
  SYG_CHECK_LAUNCH("feature_block");
  return SYG_OK;
}

extern "C" int syg_logmel_dct_f32(float* mel, int64_t B, int M, int64_t T, const float* dct, int K,
                                  const float* lifter, float amin, float top_db, int ref_is_max, float ref_value,
                                  float* logmel_out, float* mfcc_out, void* stream) {
  SYG_REQUIRE(mel, "logmel_dct: null mel pointer");
  SYG_REQUIRE(B >= 1 && M >= 1 && T >= 1, "logmel_dct: need B, M, T >= 1");
  SYG_REQUIRE(ref_is_max >= 0 && ref_is_max <= 2, "logmel_dct: ref_is_max must be 0, 1 or 2");
  SYG_REQUIRE(ref_is_max == 2 || amin >= 1.17549435e-38f, "logmel_dct: amin must be a positive normal float (>= 1.17549435e-38)");
  SYG_REQUIRE(ref_is_max != 2 || mfcc_out, "logmel_dct: DCT-only mode needs mfcc_out");
  if (mfcc_out) SYG_REQUIRE(dct && K >= 1 && K <= M, "logmel_dct: need dct and 1 <= K <= M (K=%d M=%d)", K, M);
  SYG_REQUIRE(B < (int64_t)0x7fffffff, "logmel_dct: batch too large");
  const size_t lds = (size_t)M * (size_t)T * sizeof(float);
  if (ref_is_max != 2 && lds <= 144 * 1024 && M <= 128) {
    if (lds > 48 * 1024) {
      hipError_t e = hipFuncSetAttribute((const void*)logmel_dct_lds_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e != hipSuccess) { set_error("logmel_dct: cannot reserve %zu B of LDS", lds); return SYG_E_LAUNCH; }
    }
    hipLaunchKernelGGL(logmel_dct_lds_kernel<true>, dim3((unsigned)B), dim3(NTL), lds, (hipStream_t)stream, mel, M, T, dct, K,
                       lifter, amin, top_db, ref_is_max, ref_value, logmel_out, mfcc_out);
    SYG_CHECK_LAUNCH("logmel_dct");
    return SYG_OK;
  }
  hipLaunchKernelGGL(logmel_dct_kernel, dim3((unsigned)B), dim3(NT), 0, (hipStream_t)stream, mel, M, T, dct, K,
                     lifter, amin, top_db, ref_is_max, ref_value, logmel_out, mfcc_out);
  SYG_CHECK_LAUNCH("logmel_dct");
  return SYG_OK;
}

// MFCCs alone from a mel power matrix: power_to_db(ref, amin, top_db) + DCT rows (+ lifter) as syg_logmel_dct_f32, but the
// dB matrix is not written anywhere when the clip's matrix fits the LDS (M * T * 4 <= 144 KiB, M <= 128).  Larger clips take
// syg_logmel_dct_f32's in-place form: `mel` is scratch for this call either way (callers pass a temporary).
extern "C" int syg_mel_mfcc_f32(float* mel, int64_t B, int M, int64_t T, const float* dct, int K, const float* lifter,
                                float amin, float top_db, int ref_is_max, float ref_value, float* mfcc_out, void* stream) {
  SYG_REQUIRE(mel && mfcc_out && dct, "mel_mfcc: null pointer argument");
  SYG_REQUIRE(B >= 1 && M >= 1 && T >= 1 && B < (int64_t)0x7fffffff, "mel_mfcc: need B, M, T >= 1");
  SYG_REQUIRE(ref_is_max == 0 || ref_is_max == 1, "mel_mfcc: ref_is_max must be 0 or 1");
  SYG_REQUIRE(amin >= 1.17549435e-38f, "mel_mfcc: amin must be a positive normal float (>= 1.17549435e-38)");
  SYG_REQUIRE(K >= 1 && K <= M, "mel_mfcc: need 1 <= K <= M (K=%d M=%d)", K, M);
  const size_t lds = (size_t)M * (size_t)T * sizeof(float);
  if (lds <= 144 * 1024 && M <= 128) {
    if (lds > 48 * 1024) {
      hipError_t e = hipFuncSetAttribute((const void*)logmel_dct_lds_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e != hipSuccess) { set_error("mel_mfcc: cannot reserve %zu B of LDS", lds); return SYG_E_LAUNCH; }
    }
    hipLaunchKernelGGL(logmel_dct_lds_kernel<false>, dim3((unsigned)B), dim3(NTL), lds, (hipStream_t)stream, mel, M, T, dct, K,
                       lifter, amin, top_db, ref_is_max, ref_value, (float*)nullptr, mfcc_out);
    SYG_CHECK_LAUNCH("mel_mfcc");
    return SYG_OK;
  }
  return syg_logmel_dct_f32(mel, B, M, T, dct, K, lifter, amin, top_db, ref_is_max, ref_value, nullptr, mfcc_out, stream);
}
