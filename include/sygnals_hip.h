/*
 * sygnals_hip.h -- C ABI of libsygnals_hip.so, the MI355X (gfx950) backend for the
 * sygnals windowed-transform / feature-extraction hot path.
 *
 * The reference (araray/sygnals) is pure Python and has no FFI for this path; its
 * boundary is the set of Python functions cited per entry point below (paths are
 * relative to the reference repository).  A binding a maintainer would add is a
 * ctypes stub -- see INTEGRATION.md.  sygnals_amd/_lib.py is that stub for this repo.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless its name ends in `_host`;
 *   - the caller owns every buffer (inputs, outputs, tables, workspaces); the library
 *     allocates nothing and keeps no global state besides a thread-local error string;
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); calls only
 *     enqueue work on it and never synchronise;
 *   - return value: 0 on success, negative SYG_E_* on error, message via syg_last_error();
 *   - real data is float32, complex data is interleaved float32 (re, im);
 *   - 2-D/3-D arrays are dense row-major with the stated shape unless a leading
 *     dimension (`ld*`, in elements) is given.
 */
#ifndef SYGNALS_HIP_H
#define SYGNALS_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SYG_ABI_VERSION 1

#define SYG_OK 0
#define SYG_E_INVALID (-1)   /* bad argument (shape, size, unsupported parameter) */
#define SYG_E_LAUNCH (-2)    /* HIP launch / runtime error */
#define SYG_E_UNSUPPORTED (-3)

int syg_abi_version(void);
const char* syg_last_error(void);

/* ---------------------------------------------------------------------------------
 * Fused headline path: framed STFT (n_fft = 2048) -> |X|^2 -> mel filterbank.
 * Replaces, per clip, librosa.stft + np.abs + **2 + librosa.feature.melspectrogram as
 * called at sygnals/core/features/manager.py:184-187, 198, 219-222.
 *
 *   y          [B, L] float32, row stride ldy            (clips)
 *   window     [2048] float32  periodic analysis window (already centre-padded)
 *   twiddle    [2048] complex  W_2048^k = exp(-2*pi*i*k/2048)
 *   wpacked    packed block-sparse mel weights (see syg_mel_plan_* below / _tables.py)
 *   plan_host  HOST int32[1 + 4*8]: {n_tiles, tile[8], k0[8], nsteps[8], woff[8]}
 *   mel_out    [B, n_mels, T] float32 mel POWER spectrogram
 *   stats_out  optional [B, SYG_NSTAT, T] float32 per-frame spectral statistics
 *              (NULL to skip), rows in SYG_STAT_* order; replaces the per-frame loop
 *              manager.py:304-316 over frequency_domain.py:24-386
 *   contrast   optional: cplan_host HOST int32[1 + 3*SYG_MAX_BANDS] {n_rows, lo[], hi[], k[]}
 *              and contrast_out [B, 2, n_rows, T] (peak, valley means; NULL to skip);
 *              replaces the band loop of librosa.feature.spectral_contrast reached from
 *              frequency_domain.py:200-207
 *   T          number of frames, = 1 + L/hop (center) or 1 + (L-2048)/hop
 * ------------------------------------------------------------------------------- */
#define SYG_NSTAT 8
#define SYG_STAT_CENTROID 0
#define SYG_STAT_BANDWIDTH 1
#define SYG_STAT_FLATNESS 2
#define SYG_STAT_ROLLOFF_BIN 3
#define SYG_STAT_DOMINANT_BIN 4
#define SYG_STAT_MAG_SUM 5
#define SYG_STAT_POWER_SUM 6
#define SYG_STAT_ROLLOFF_MARGIN 7
#define SYG_MAX_BANDS 16

int syg_stft2048_mel_f32(const float* y, int64_t B, int64_t L, int64_t ldy, int hop, int center, int64_t T,
                         const float* window, const float* twiddle, const float* wpacked,
                         const int32_t* plan_host, int n_mels, float* mel_out,
                         float sr, float roll_percent, float bw_p, float* stats_out,
                         const int32_t* cplan_host, float* contrast_out, void* stream);

/* Same front end, complex STFT output (librosa.stft as called by compute_stft,
 * sygnals/core/dsp.py:167-229).  out [B, T, 1025] complex64, FRAME-major. */
int syg_stft2048_c2c_f32(const float* y, int64_t B, int64_t L, int64_t ldy, int hop, int center, int64_t T,
                         const float* window, const float* twiddle, float* out, void* stream);

/* ---------------------------------------------------------------------------------
 * power_to_db + DCT-II (+ lifter): librosa.power_to_db(S_mel, ref=np.max) at
 * manager.py:223 and librosa.feature.mfcc(S=..) at cepstral.py:106-115.
 *   mel        [B, M, T] mel power; converted to dB IN PLACE unless logmel_out given
 *   dct        [K, M] DCT matrix rows (orthonormal DCT-II rows for the default)
 *   lifter     optional [K] multiplicative lifter (NULL = none)
 *   ref_is_max 1: ref = max over the clip's [M, T] (ref=np.max); 0: ref = ref_value
 *   top_db     < 0 disables the clamp
 *   mfcc_out   [B, K, T]  (NULL: only the dB conversion)
 * ------------------------------------------------------------------------------- */
int syg_logmel_dct_f32(float* mel, int64_t B, int M, int64_t T, const float* dct, int K,
                       const float* lifter, float amin, float top_db, int ref_is_max, float ref_value,
                       float* logmel_out, float* mfcc_out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SYGNALS_HIP_H */
