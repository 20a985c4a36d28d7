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
 *     allocates nothing; its only state is a thread-local error string and the process-wide options of
 *     syg_set_option() below (no caches keyed by shape or device: kernel attributes are set at every launch, the
 *     CU count is queried per call).  Nothing is read from the environment;
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
/* 0 = product build; non-zero = a development variant (ablation / in-kernel timeline builds, -DSYG_ABL=n) whose
 * results are wrong by design: a binding must refuse to use such a library. */
int syg_build_variant(void);
const char* syg_last_error(void);

/* Process-wide options (atomics; read when a launch is planned).  All but the first select between kernels that the
 * tests hold to the same results and exist for those tests; production code leaves them at their defaults.
 *   SYG_OPT_RESERVED_CUS  n >= 0 (default 0): the persistent syg_stft2048_* kernels leave n CUs out of their grid, for a
 *                         collective (RCCL's send / receive workgroups) running beside them
 *   SYG_OPT_STFT_LOAD     -1 (default: staged tiles) | 0 | 1 | 2: frame load path of the syg_stft2048_* kernels
 *   SYG_OPT_SOS_CLIP      1 (default) | 0: clip-resident form of syg_sosfiltfilt_f32 / the chunked form only
 *   SYG_OPT_CQT_STAGED    -1 (default: where it pays) | 0 (never) | 1 | 2 (also at hop = n_fft / 2): staged form of
 *                         syg_cqt_octave_bf16x3_f32
 * syg_set_option returns SYG_OK or SYG_E_INVALID (unknown key / value out of range); syg_get_option the current value. */
#define SYG_OPT_RESERVED_CUS 0
#define SYG_OPT_STFT_LOAD 1
#define SYG_OPT_SOS_CLIP 2
#define SYG_OPT_CQT_STAGED 3
#define SYG_OPT_COUNT 4
int syg_set_option(int key, int value);
int syg_get_option(int key);

/* ---------------------------------------------------------------------------------
 * Fused headline path: framed STFT (n_fft = 2048) -> |X|^2 -> mel filterbank.
 * Replaces, per clip, librosa.stft + np.abs + **2 + librosa.feature.melspectrogram as
 * called at sygnals/core/features/manager.py:184-187, 198, 219-222.
 *
 *   y          [B, L] float32, row stride ldy            (clips)
 *   window     [2048] float32  periodic analysis window (already centre-padded)
 *   twiddle    [2048] complex  W_2048^k = exp(-2*pi*i*k/2048)
 *   wpacked    packed block-sparse mel weights for v_mfma_f32_4x4x1_16b_f32 (sygnals_amd/_tables.py: pack_mel_plan):
 *              [wave][steps / 4][64 lanes][4] A operands (lane l of step i: basis[4 g + (l & 3)][k0 + i] of its slot
 *              l >> 4), then >= 6 groups of zero rows (every wave pre-loads 6 groups unconditionally), then four
 *              int32 tables of 64 entries at float offset table_off: first bin of slot (wave * 4 + s), mel group of the
 *              slot, first slot of group g, slot count of group g (slots of a group are consecutive, ascending bins)
 *   plan_host  HOST int32[5]: {2 (layout), n_waves (8 or 16), steps (multiple of 4), n_groups = ceil(n_mels / 4) <= 64,
 *              table_off}
 *   mel_out    [B, n_mels, T] float32 mel POWER spectrogram
 *   stats_out  optional [B, SYG_NSTAT, T] float32 per-frame spectral statistics
 *              (NULL to skip), rows in SYG_STAT_* order; only the rows selected by stats_mask
 *              (SYG_SM_* bits) are computed and written; replaces the per-frame loop
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
#define SYG_SM_CENTROID 1   /* also MAG_SUM */
#define SYG_SM_BANDWIDTH 2
#define SYG_SM_FLATNESS 4
#define SYG_SM_ROLLOFF 8    /* also POWER_SUM, ROLLOFF_MARGIN */
#define SYG_SM_DOMINANT 16
#define SYG_SM_NO_MARGIN 32 /* with SYG_SM_ROLLOFF: the ROLLOFF_MARGIN row is not computed (callers that only read the bin) */

int syg_stft2048_mel_f32(const float* y, int64_t B, int64_t L, int64_t ldy, int hop, int center, int64_t T,
                         const float* window, const float* twiddle, const float* wpacked,
                         const int32_t* plan_host, int n_mels, float* mel_out,
                         float sr, float roll_percent, float bw_p, int stats_mask, float* stats_out,
                         const int32_t* cplan_host, float* contrast_out, void* stream);

/* Same front end, complex STFT output (librosa.stft as called by compute_stft,
 * sygnals/core/dsp.py:167-229).  out [B, T, 1025] complex64, FRAME-major. */
int syg_stft2048_c2c_f32(const float* y, int64_t B, int64_t L, int64_t ldy, int hop, int center, int64_t T,
                         const float* window, const float* twiddle, float* out, void* stream);

/* ---------------------------------------------------------------------------------
 * Whole MFCC chain in one launch (the headline path): librosa.stft -> |.|^2 -> mel filterbank ->
 * power_to_db(ref=np.max, top_db) -> DCT-II rows (+ lifter), i.e. manager.py:184-187, 198, 219-223 and
 * cepstral.py:106-115.  A workgroup owns whole clips; the clip's [n_mels, T] mel matrix stays in LDS, so
 * only the samples are read from and the MFCCs written to HBM.
 *   y .. n_mels   as syg_stft2048_mel_f32 (the plan must be a 16-wave plan)
 *   dct        [n_mfcc, n_mels] DCT matrix rows;  lifter: optional [n_mfcc] (NULL = none)
 *   amin, top_db (< 0: no clamp), ref_is_max (1: per-clip max, 0: ref_value): as syg_logmel_dct_f32
 *   mel_out    optional [B, n_mels, T] copy of the mel POWER (NULL = not stored)
 *   mfcc_out   [B, n_mfcc, T]
 * Fails (SYG_E_ARG) when n_mels * 16*ceil(T/16) floats do not fit the LDS left beside the transform buffers: ~21 KiB
 * beside the tile stage buffer (e.g. n_mels=40: T <= 128 frames), ~60 KiB in its place -- the launch then loads its frames
 * straight from global memory (n_mels=128: T <= 112); callers otherwise use the two-launch form
 * syg_stft2048_mel_f32 + syg_logmel_dct_f32.
 * ------------------------------------------------------------------------------- */
/* 0: the shape does not fit the one-launch form; 2: it fits beside the stage buffer; 1: in the stage buffer's place */
int syg_stft2048_mfcc_fits(int n_mels, int64_t T, int n_mfcc);
int syg_stft2048_mfcc_f32(const float* y, int64_t B, int64_t L, int64_t ldy, int hop, int center,
                          int64_t T, const float* window, const float* twiddle, const float* wpacked,
                          const int32_t* plan_host, int n_mels, const float* dct, int n_mfcc,
                          const float* lifter, float amin, float top_db, int ref_is_max, float ref_value,
                          float* mel_out, float* mfcc_out, void* stream);

/* The same chain for TRIANGULAR filterbanks (librosa.filters.mel as manager.py:198 / cepstral.py:106 build it), the mel
 * projection by SEGMENT SUMS: between two band edges the weights of the rising and of the falling band are affine in
 * the bin index, so a run of bins contributes a T0 + b T1 (T0 = sum p, T1 = sum i p) -- each wave projects its own
 * power row, no weight matrix is read and the projection needs no workgroup barrier.
 *   segtab     device, 16-byte aligned: the piece table of sygnals_amd._tables.pack_mel_segments, [2][2][64][4] words
 *              (n_segtab = 1024); filterbanks whose pieces do not fit 128 lane slots have no table -- use the matrix form
 *   other arguments as syg_stft2048_mfcc_f32 (no mel copy: the matrix form stores one)
 * Needs TWO mel matrices in LDS (the dB + DCT of a clip runs beside the next clip's first tile):
 * syg_stft2048_mfcc_tri_fits() says whether a shape fits. */
int syg_stft2048_mfcc_tri_fits(int n_mels, int64_t T, int n_mfcc);
int syg_stft2048_mfcc_tri_f32(const float* y, int64_t B, int64_t L, int64_t ldy, int hop, int center,
                              int64_t T, const float* window, const float* twiddle, const float* segtab,
                              int n_segtab, int n_mels, const float* dct, int n_mfcc, const float* lifter,
                              float amin, float top_db, int ref_is_max, float ref_value, float* mfcc_out,
                              void* stream);

/* The statistics / contrast rows of syg_stft2048_mel_f32 AND the clip-resident MFCC of syg_stft2048_mfcc_tri_f32 from ONE
 * launch: extract_features(["mfcc", "spectral_centroid", "spectral_rolloff", "spectral_contrast"]) (BASELINE config C4,
 * manager.py:289-371) with only the samples read and the feature rows written -- the mel matrix never reaches HBM.
 * Arguments as in those two entry points (at least one of stats_out / contrast_out).  The MFCC rows of clip b go to
 * mfcc_out + (b * mfcc_rows_per_clip) * T: with mfcc_rows_per_clip > n_mfcc they are the head of a wider per-clip block
 * whose other rows syg_feature_block_f32(mel = NULL, ...) fills.  Shapes that fit: syg_stft2048_mfcc_tri_fits(). */
int syg_stft2048_features_tri_f32(const float* y, int64_t B, int64_t L, int64_t ldy, int hop, int center, int64_t T,
                                  const float* window, const float* twiddle, const float* segtab, int n_segtab,
                                  int n_mels, const float* dct, int n_mfcc, const float* lifter, float amin,
                                  float top_db, int ref_is_max, float ref_value, float sr, float roll_percent,
                                  float bw_p, int stats_mask, float* stats_out, const int32_t* cplan_host,
                                  float* contrast_out, float* mfcc_out, int mfcc_rows_per_clip, void* stream);

/* The TILE form of the segment-sum projection: samples in, mel POWER out; replaces librosa.stft + np.abs + **2 +
 * librosa.feature.melspectrogram (manager.py:184-187, 198, 219-222) like syg_stft2048_mel_f32, without a weight matrix and
 * without the projection's barriers.  Optional statistics / contrast rows as in syg_stft2048_mel_f32 (manager.py:289-343).
 * Any hop; tiles are shared out evenly over the CUs.
 *   segtab    device, 16-byte aligned.  n_segtab = 2048: sygnals_amd._tables.pack_mel_segments(sr, 2048, n_mels, fmin, fmax,
 *             n_pass=4, row_base=4), [4][2][64][4] 32-bit words -- up to 256 pieces: the reference's default of 128 bands
 *             (sygnals/core/features/manager.py:214; `sygnals features extract -f mfcc`, cli/features_cmd.py:82-90, always
 *             runs it), 64 ... 200 bands at the usual rates.  n_segtab = 1024: the two-pass table of
 *             syg_stft2048_mfcc_tri_f32 (up to 128 pieces, e.g. 40 bands)
 *   waves     16 (one workgroup per CU) or 8 (two per CU)
 *   mel_out   [B, n_mels, T]; stats_out / cplan_host / contrast_out as in syg_stft2048_mel_f32 (NULL to skip) */
int syg_stft2048_mel_tri_f32(const float* y, int64_t B, int64_t L, int64_t ldy, int hop, int center, int64_t T,
                             const float* window, const float* twiddle, const float* segtab, int n_segtab, int n_mels,
                             float* mel_out, float sr, float roll_percent, float bw_p, int stats_mask, float* stats_out,
                             const int32_t* cplan_host, float* contrast_out, int waves, void* stream);

/* The per-frame statistics / contrast tail means of syg_stft2048_mel_f32 WITHOUT the mel spectrogram: spectral_centroid /
 * bandwidth / flatness / rolloff / contrast (manager.py:289-343 -> frequency_domain.py:25-212) only need |X|.  The kernel of
 * syg_stft2048_features_tri_f32 with nothing projected and no clip epilogue; same stats_mask / stats_out [B, SYG_NSTAT, T] /
 * cplan_host / contrast_out [B, 2, n_rows, T] as syg_stft2048_mel_f32, results bit-identical to it.  hop <= 512. */
int syg_stft2048_stats_f32(const float* y, int64_t B, int64_t L, int64_t ldy, int hop, int center, int64_t T,
                           const float* window, const float* twiddle, float sr, float roll_percent, float bw_p,
                           int stats_mask, float* stats_out, const int32_t* cplan_host, float* contrast_out,
                           void* stream);

/* frame_length 4096 (librosa.stft + np.abs(.)**2 + melspectrogram, manager.py:184-187, 198, 219-222): samples in, mel power
 * out, one wave per frame (the 4096-point real transform of syg_welch_f32's wave kernel), the mel projection by segment
 * sums as in syg_stft2048_mfcc_tri_f32 with a FOUR-pass piece table (pack_mel_segments(..., n_pass=4): 2048 words).
 *   y [B, L] (row stride ldy), window [4096] device (16-byte aligned), twiddle: W_4096^k for k = 0 .. 4095
 *   mel_out [B, n_mels, T] */
int syg_stft_mel_w4096_f32(const float* y, int64_t B, int64_t L, int64_t ldy, int hop, int center, int64_t T,
                           const float* window, const float* twiddle, const float* segtab, int n_segtab, int n_mels,
                           float* mel_out, void* stream);

/* frame_length 1024 (the reference's own tests and CLI: tests/test_features_manager.py:183-220, cli/features_cmd.py:35),
 * power 2: samples in, mel power out, free-running waves -- a wave owns two frames per 1024-point complex transform and
 * projects its two power rows by segment sums (no weight matrix, no workgroup barrier per tile).  segtab: the two-row
 * table of sygnals_amd._tables.pack_mel_segments_rows(sr, 1024, n_mels, fmin, fmax) (2048 words, 16-byte aligned).
 * window [1024], twiddle: W_1024^k for k = 0 .. 1023; mel_out [B, n_mels, T].  Other powers and filterbanks without a
 * table: syg_stft_mel_pow2_f32. */
int syg_stft_mel_w1024_seg_f32(const float* y, int64_t B, int64_t L, int64_t ldy, int hop, int center, int64_t T,
                               const float* window, const float* twiddle, const float* segtab, int n_segtab, int n_mels,
                               float* mel_out, void* stream);

/* syg_stft_mel_w1024_seg_f32 with the per-frame statistics / contrast rows of syg_stft2048_mel_f32 from the same launch
 * (bins 0 .. 512, bin frequency k sr / 1024): extract_features(frame_length=1024, [spectral features ...]) -- the call of
 * the reference's own manager tests (tests/test_features_manager.py:58-62, 167-174; manager.py:289-343 over
 * frequency_domain.py:24-386) -- without a spectrogram in HBM.  stats_out [B, SYG_NSTAT, T] (rows selected by stats_mask)
 * and / or cplan_host + contrast_out [B, 2, n_rows, T]: at least one.  The mel block is optional: segtab == NULL and
 * mel_out == NULL compute the rows alone. */
int syg_stft_rows_w1024_f32(const float* y, int64_t B, int64_t L, int64_t ldy, int hop, int center, int64_t T,
                            const float* window, const float* twiddle, const float* segtab, int n_segtab, int n_mels,
                            float* mel_out, float sr, float roll_percent, float bw_p, int stats_mask, float* stats_out,
                            const int32_t* cplan_host, float* contrast_out, void* stream);

/* The same for frame_length 512 and 256 (256: the reference's short-signal tests): a wave owns four / eight frames per
 * transform and projects its four / eight power rows.  segtab: pack_mel_segments_rows(sr, n_fft, n_mels, fmin, fmax,
 * rows=4, row_words=296 (512) / 160 (256), n_pass=1, block=16 (512) / 8 (256)) (2048 words); twiddle: W_1024^k for k = 0 .. 1023. */
int syg_stft_mel_wseg_small_f32(const float* y, int64_t B, int64_t L, int64_t ldy, int n_fft, int hop, int center,
                                int64_t T, const float* window, const float* twiddle, const float* segtab,
                                int n_segtab, int n_mels, float* mel_out, void* stream);

/* The per-frame statistics / contrast rows of syg_stft2048_mel_f32 for frame lengths 512 and 256 (bins 0 .. n_fft / 2, bin
 * frequency k sr / n_fft) from the transform of syg_stft_mel_wseg_small_f32, nothing projected: extract_features(
 * frame_length=512 | 256, [spectral features]) -- 256 is the frame length of the reference's short-signal tests
 * (tests/test_features_manager.py:183-220; manager.py:289-343 over frequency_domain.py:24-386) -- without a spectrogram in
 * HBM.  stats_out [B, SYG_NSTAT, T] (rows selected by stats_mask) and / or cplan_host + contrast_out [B, 2, n_rows, T]:
 * at least one.  window [n_fft]; twiddle: W_1024^k, k = 0 .. 1023. */
int syg_stft_rows_wsmall_f32(const float* y, int64_t B, int64_t L, int64_t ldy, int n_fft, int hop, int center, int64_t T,
                             const float* window, const float* twiddle, float sr, float roll_percent, float bw_p,
                             int stats_mask, float* stats_out, const int32_t* cplan_host, float* contrast_out, void* stream);

/* The same rows for frame length 4096 (2049 bins, bin frequency k sr / 4096) from the launch of syg_stft_mel_w4096_f32
 * (one wave per frame): extract_features(frame_length=4096, [spectral features (+ mfcc)]) -- manager.py:289-343 over
 * frequency_domain.py:24-386 -- without a spectrogram in HBM.  stats_out [B, SYG_NSTAT, T] (rows selected by stats_mask)
 * and / or cplan_host + contrast_out [B, 2, n_rows, T]: at least one; with segtab (2048 words, as for
 * syg_stft_mel_w4096_f32) and mel_out [B, n_mels, T] also the mel power block, with segtab NULL nothing is projected.
 * window [4096]; twiddle: W_4096^k, k = 0 .. 4095. */
int syg_stft_rows_w4096_f32(const float* y, int64_t B, int64_t L, int64_t ldy, int hop, int center, int64_t T,
                            const float* window, const float* twiddle, const float* segtab, int n_segtab, int n_mels,
                            float* mel_out, float sr, float roll_percent, float bw_p, int stats_mask, float* stats_out,
                            const int32_t* cplan_host, float* contrast_out, void* stream);

/* ---------------------------------------------------------------------------------
 * power_to_db + DCT-II (+ lifter): librosa.power_to_db(S_mel, ref=np.max) at
 * manager.py:223 and librosa.feature.mfcc(S=..) at cepstral.py:106-115.
 *   mel        [B, M, T] mel power; converted to dB IN PLACE unless logmel_out given
 *   dct        [K, M] DCT matrix rows (orthonormal DCT-II rows for the default)
 *   lifter     optional [K] multiplicative lifter (NULL = none)
 *   ref_is_max 1: ref = max over the clip's [M, T] (ref=np.max); 0: ref = ref_value;
 *              2: `mel` already holds dB values (mfcc(S=log_mel)): DCT only, mel untouched
 *   top_db     < 0 disables the clamp
 *   mfcc_out   [B, K, T]  (NULL: only the dB conversion)
 * ------------------------------------------------------------------------------- */
int syg_logmel_dct_f32(float* mel, int64_t B, int M, int64_t T, const float* dct, int K,
                       const float* lifter, float amin, float top_db, int ref_is_max, float ref_value,
                       float* logmel_out, float* mfcc_out, void* stream);
/* MFCCs alone (librosa.feature.mfcc(S=power_to_db(S_mel, ref=np.max)), cepstral.py:106-115 behind manager.py:223): as above,
 * but the dB matrix is written nowhere when the clip's matrix fits the LDS; `mel` is scratch for the call. */
int syg_mel_mfcc_f32(float* mel, int64_t B, int M, int64_t T, const float* dct, int K, const float* lifter, float amin,
                     float top_db, int ref_is_max, float ref_value, float* mfcc_out, void* stream);

/* ---------------------------------------------------------------------------------
 * Generic batched power-of-two FFT in LDS (n = 2^k, 2 <= n <= 8192): scipy.fft.fft /
 * ifft as called by compute_fft / compute_ifft, sygnals/core/dsp.py:104, 151.
 *   in/out   [batch, n] complex64 (may alias); inverse != 0 scales by 1/n
 *   twiddle  [n] complex W_n^k = exp(-2*pi*i*k/n)
 * ------------------------------------------------------------------------------- */
int syg_fft_pow2_c2c_f32(const float* in, float* out, int64_t batch, int n, int inverse,
                         const float* twiddle, void* stream);

/* Strided form used to compose transforms longer than 8192 points (four-step) and Bluestein
 * for arbitrary n (scipy.fft.fft(x, n) accepts any n, dsp.py:104).  Element e of transform
 * (o, b) is at in[o*in_os + b*in_bs + e*in_es] (complex elements); if bign > 0 output k of
 * transform b is multiplied by W_bign^(b*k); every output is multiplied by `scale`.  in != out. */
int syg_fft_pow2_strided_c2c_f32(const float* in, float* out, int64_t outer, int64_t batch, int n, int inverse,
                                 const float* twiddle, int64_t in_os, int64_t in_bs, int64_t in_es,
                                 int64_t out_os, int64_t out_bs, int64_t out_es, int64_t bign, float scale,
                                 void* stream);

/* The same for lengths n = 2^a 3^b 5^c 7^d <= 8192 (mixed-radix Stockham: one second of audio at 48 / 44.1 / 16 kHz is
 * such a length, none a power of two); twiddle [n] = W_n^k.  syg_fft_mixed_plan returns the number of passes (0: n has
 * another prime factor -> Bluestein) and, if radices_host is given, their radices.  Longer 7-smooth lengths are
 * composed four-step from two such transforms by the caller, exactly like the power-of-two case. */
int syg_fft_mixed_plan(int64_t n, int32_t* radices_host, int max_passes);
int syg_fft_mixed_strided_c2c_f32(const float* in, float* out, int64_t outer, int64_t batch, int n, int inverse,
                                  const float* twiddle, int64_t in_os, int64_t in_bs, int64_t in_es,
                                  int64_t out_os, int64_t out_bs, int64_t out_es, int64_t bign, float scale,
                                  void* stream);
/* The two strided transforms with FUSED ENDS (the analytic signal / envelope of scipy.signal.hilbert as transforms.py:119-151
 * and dsp.py:565-636 call it, without its three element-wise passes):
 *   flags & 1 (SYG_FFT_REAL_IN)   `in` is a REAL array (same element indexing), imaginary part 0
 *   flags & 2 (SYG_FFT_ABS_OUT)   `out` is a REAL array that receives |result|
 *   flags & 4 (SYG_FFT_PAIR_IN)   `in` is a REAL array read as the complex sequence (x[2 p], x[2 p + 1]), zero from sample
 *                                 in_valid on; in_os is then the row stride in floats (the packed real-input transform of
 *                                 syg_rconv_spectrum_c64's convolutions without a packing pass)
 *   mask_n > 0                    every loaded element is multiplied by the analytic-signal weight of its position p inside
 *                                 the row (the row is a length-mask_n spectrum): 1 at p = 0 and 2 p = mask_n, 2 for
 *                                 2 p < mask_n, 0 above */
int syg_fft_pow2_strided_ex_f32(const float* in, float* out, int64_t outer, int64_t batch, int n, int inverse,
                                const float* twiddle, int64_t in_os, int64_t in_bs, int64_t in_es, int64_t out_os,
                                int64_t out_bs, int64_t out_es, int64_t bign, float scale, int flags, int64_t mask_n,
                                int64_t in_valid, void* stream);
int syg_fft_mixed_strided_ex_f32(const float* in, float* out, int64_t outer, int64_t batch, int n, int inverse,
                                 const float* twiddle, int64_t in_os, int64_t in_bs, int64_t in_es, int64_t out_os,
                                 int64_t out_bs, int64_t out_es, int64_t bign, float scale, int flags, int64_t mask_n,
                                 int64_t in_valid, void* stream);

/* out[i] = a[i] * b[i mod nb] (complex64; conj_b != 0 multiplies by conj(b)); may be in place. */
int syg_cmul_c64(const float* a, const float* b, float* out, int64_t na, int64_t nb, int conj_b, void* stream);

/* Real rows -> zero-padded / truncated complex rows with an optional window (apply_window,
 * dsp.py:641-691, then the implicit pad/truncate of fft(x, n)): out [rows, n] complex64. */
int syg_pack_real_c64(const float* x, int64_t rows, int64_t len, int64_t ldx, const float* window, float* out,
                      int64_t n, void* stream);

/* Generic framed STFT for any power-of-two n_fft in [8, 16384] (slow path of
 * compute_stft and of extract_features for frame lengths other than 2048).
 *   twiddle [n_fft + n_fft/2] complex: W_nfft^k (k < n_fft) followed by W_{nfft/2}^k
 *   out [B, T, 1 + n_fft/2] complex64, frame-major. */
int syg_stft_pow2_c2c_f32(const float* y, int64_t B, int64_t L, int64_t ldy, int n_fft, int hop, int center,
                          int64_t T, const float* window, const float* twiddle, float* out, void* stream);

/* Elementwise helpers on frame-major spectra: out[i] = |X[i]|^power (power 1 or 2). */
int syg_cabs_pow_f32(const float* x_c64, int64_t n, int power, float* out, void* stream);

/* Dense mel projection for the generic path: mel[b, m, t] = sum_f basis[m, f] * P[b, t, f]. */
int syg_mel_dense_f32(const float* P, int64_t B, int64_t T, int F, const float* basis, int M,
                      float* mel_out, void* stream);

/* ---------------------------------------------------------------------------------
 * Fused front end for the OTHER power-of-two frame lengths (n_fft = 64 ... 1024; the reference's own tests and CLI use
 * 1024 and 256: tests/test_features_manager.py:183-220, cli/features_cmd.py:35): librosa.stft -> |.|^power -> mel
 * filterbank in one launch, no spectrogram in HBM (manager.py:184-187, 198, 219-222).
 *   y .. window   as syg_stft_pow2_c2c_f32;  twiddle [n_fft + n_fft/2] complex (W_nfft^k, then W_{nfft/2}^k), for
 *              n_fft = 512 and 256 followed by W_1024^k (1024 entries: four / eight frames share one 1024-point wave
 *              transform)
 *   basis_p    [16*ceil(n_mels/16), Fp] the dense filterbank, zero padded: Fp = (1 + n_fft/2) rounded up to a multiple
 *              of 16; 16-byte aligned
 *   power      1 (magnitude) or 2 (power)
 *   mel_out    [B, n_mels, T]
 * ------------------------------------------------------------------------------- */
int syg_stft_mel_pow2_f32(const float* y, int64_t B, int64_t L, int64_t ldy, int n_fft, int hop, int center,
                          int64_t T, const float* window, const float* twiddle, const float* basis_p, int Fp,
                          int n_mels, int power, float* mel_out, void* stream);

/* The whole MFCC chain for those frame lengths in one launch (a workgroup owns a clip, the clip's mel matrix stays in
 * LDS): ... -> power_to_db(ref=np.max, top_db) -> DCT-II rows (+ lifter) (manager.py:223, cepstral.py:106-115).
 * Arguments as syg_stft_mel_pow2_f32 (power 2) + those of syg_stft2048_mfcc_f32.  syg_stft_mfcc_pow2_fits() says
 * whether the clip's mel matrix fits the LDS; when not, use syg_stft_mel_pow2_f32 + syg_logmel_dct_f32. */
int syg_stft_mfcc_pow2_fits(int n_fft, int n_mels, int64_t T, int n_mfcc);
int syg_stft_mfcc_pow2_f32(const float* y, int64_t B, int64_t L, int64_t ldy, int n_fft, int hop, int center,
                           int64_t T, const float* window, const float* twiddle, const float* basis_p, int Fp,
                           int n_mels, const float* dct, int n_mfcc, const float* lifter, float amin, float top_db,
                           int ref_is_max, float ref_value, float* mel_out, float* mfcc_out, void* stream);

/* Per-frame spectral statistics of frame-major magnitude spectra mag [N, F] with bin
 * frequencies freqs [F] (frequency_domain.py:24-386).  stats_out [SYG_NSTAT, N]. */
int syg_spectral_stats_f32(const float* mag, int64_t N, int F, const float* freqs, float roll_percent,
                           float bw_p, float* stats_out, void* stream);

/* Spectral-contrast peak/valley means of frame-major magnitude spectra mag [N, F]:
 * for each band row r: mean of the k[r] smallest / largest values of bins lo[r]..hi[r]-1.
 *   out [2, n_rows, N] (peak then valley). */
int syg_contrast_pv_f32(const float* mag, int64_t N, int F, const int32_t* cplan_host, float* out, void* stream);

/* contrast[b, r, t] = power_to_db(peak) - power_to_db(valley) (ref 1, amin, top_db clamp per
 * [R, T] matrix), the last step of librosa.feature.spectral_contrast (frequency_domain.py:200-207).
 *   pv [B, 2, R, T] (peak, valley) -> out [B, R, T]; top_db < 0 disables the clamp;
 *   amin <= 0 selects librosa's linear=True: out = peak - valley. */
int syg_contrast_db_f32(const float* pv, int64_t B, int R, int64_t T, float amin, float top_db, float* out,
                        void* stream);

/* ---------------------------------------------------------------------------------
 * Zero-phase SOS filtering: scipy.signal.sosfiltfilt(sos, x) (padtype='odd') as called by
 * apply_sos_filter, sygnals/core/filters.py:85-115.
 *   x, y        [B, L] float32 (row strides ldx, ldy); y may alias x
 *   sos_host    HOST float64 [n_sections, 6]
 *   zi_host     HOST float64 [n_sections, 2]  sosfilt_zi(sos)
 *   padlen      edge extension length (3*ntaps rule), must be < L
 *   work        device workspace of syg_sosfiltfilt_work_bytes(B, L, padlen, n_sections) bytes; that is 0 for
 *               clips the clip-resident form takes (L + 2 padlen <= 65536, <= 4 sections): work may then be NULL
 * ------------------------------------------------------------------------------- */
int64_t syg_sosfiltfilt_work_bytes(int64_t B, int64_t L, int padlen, int n_sections);
int syg_sosfiltfilt_f32(const float* x, int64_t B, int64_t L, int64_t ldx, const double* sos_host,
                        const double* zi_host, int n_sections, int padlen, float* y, int64_t ldy,
                        void* work, void* stream);

/* ---------------------------------------------------------------------------------
 * Welch PSD: scipy.signal.welch(..., return_onesided=True) as called by
 * compute_psd_welch, sygnals/core/dsp.py:545-555.  nfft power of two in [8, 16384],
 * nperseg <= nfft.  twiddle as for syg_stft_pow2_c2c_f32 ([nfft + nfft/2] complex).
 *   x        [B, L] float32
 *   window   [nperseg] float32
 *   detrend  0 none, 1 constant (per-segment mean removal), 2 linear (per-segment least-squares line,
 *            scipy.signal.detrend type='linear')
 *   scale    density: 1/(fs*sum(w^2)); spectrum: 1/sum(w)^2   (computed by the caller)
 *   psd_out  [B, 1 + nfft/2] float32 ; partial sums in work (syg_welch_work_bytes)
 * ------------------------------------------------------------------------------- */
int64_t syg_welch_work_bytes(int64_t B, int nfft);
int syg_welch_f32(const float* x, int64_t B, int64_t L, int64_t ldx, int nperseg, int step, int nfft,
                  const float* window, const float* twiddle, int detrend, double scale, float* psd_out,
                  void* work, void* stream);

/* ---------------------------------------------------------------------------------
 * BASELINE config C4's per-clip feature block in one launch behind syg_stft2048_mel_f32 (mel + centroid + rolloff +
 * contrast tail means): block_out [B, K + 2 + R, T] float32 with rows
 *   0 .. K-1      MFCC = DCT rows x power_to_db(mel, ref = max of the clip, amin, top_db)   (manager.py:219-223,
 *                 cepstral.py:106-115; the dB matrix stays in LDS: M * T * 4 bytes <= 150 KiB)
 *   K, K + 1      spectral centroid (Hz), spectral rolloff (Hz = bin x binhz)              (manager.py:304-316)
 *   K + 2 ..      spectral contrast dB rows from contrast_pv [B, 2, R, T] (peak, valley)    (frequency_domain.py:200-207)
 * -- the columns extract_features(["mfcc", "spectral_centroid", "spectral_rolloff", "spectral_contrast"]) returns, in
 * order: the [B/W, 22, 94] block a rank contributes to config C4's gather.  stats: [B, SYG_NSTAT, T].
 * mel == NULL (dct may then be NULL too): rows 0 .. K-1 are left as they are -- syg_stft2048_features_tri_f32 has
 * written the MFCCs there -- and only the statistics / contrast rows are filled.
 * ------------------------------------------------------------------------------- */
int syg_feature_block_f32(const float* mel, int64_t B, int M, int64_t T, const float* dct, int K, float amin,
                          float top_db, const float* stats, float binhz, const float* contrast_pv, int R,
                          float c_amin, float c_top_db, float* block_out, void* stream);

/* ---------------------------------------------------------------------------------
 * Time-domain frame features, one value per frame:
 *   rows 0..6: mean |x|, population std, skewness (scipy.stats.skew bias=False), excess kurtosis
 *              (scipy.stats.kurtosis fisher, bias=False), max |x|, crest factor, Shannon entropy of
 *              np.histogram(frame, num_bins) -- sygnals/core/features/time_domain.py:23-227 applied per frame
 *              of the zero-padded signal by manager.py:264-286
 *   row 7:     RMS energy  -- core/audio/features.py:73-131 (librosa.feature.rms, zero padding)
 *   row 8:     zero-crossing rate -- core/audio/features.py:26-71 (librosa.feature.zero_crossing_rate,
 *              edge padding, threshold 1e-10)
 *   y [B, L] float32 (row stride ldy); T as syg_stft2048_mel_f32's framing rule with n_fft = frame_length;
 *   mask: bit r selects row r (unselected rows are left untouched);  out [B, SYG_NFSTAT, T] float32.
 * syg_rms_from_spec_f32: librosa.feature.rms(S=...) for rows of magnitudes S [rows, F] -> out [rows].
 * ------------------------------------------------------------------------------- */
#define SYG_NFSTAT 9
int syg_frame_stats_f32(const float* y, int64_t B, int64_t L, int64_t ldy, int frame_length, int hop,
                        int center, int64_t T, int num_bins, int mask, float* out, void* stream);
int syg_rms_from_spec_f32(const float* S, int64_t rows, int F, int frame_length, float* out, void* stream);

/* ---------------------------------------------------------------------------------
 * Constant-Q transform building blocks: librosa.cqt as called by compute_cqt,
 * sygnals/core/dsp.py:276-284 (recursive per-octave algorithm; the host composes the octaves).
 *   syg_decimate2_f32   y[b, n] = scale * sum_j taps[j] * x[b, 2n + (ntaps-1)/2 - j], n < ceil(L/2)
 *                       (zero outside the signal) -- the decimation between octaves
 *   syg_cqt_octave_f32  rectangular-window centred STFT frame (n_fft = 2^k <= 4096, hop) of y [B, L],
 *                       times the frequency-domain basis [n_filt, n_fft/2+1] complex64 ->
 *                       out[b * out_bstride + (row0 + f) * T + t] complex64; hull_host (host int32
 *                       [2 * n_filt], may be NULL = dense): first non-zero bin and run length of every
 *                       basis row (librosa sparsifies the basis; only the run is multiplied); twiddle as for
 *                       syg_stft_pow2_c2c_f32 ([n_fft + n_fft/2] complex)
 * ------------------------------------------------------------------------------- */
int syg_decimate2_f32(const float* x, int64_t B, int64_t L, int64_t ldx, const float* taps, int ntaps, float scale,
                      float* y, int64_t ldy, void* stream);
 /* syg_decimate2_chain_f32: `levels` (1..4) successive decimations in one pass; y / ldy are HOST arrays of `levels`
 *   device pointers / row strides, level s of length ceil(L / 2^(s+1)); an entry of y may be NULL when that level is
 *   not wanted (41 taps only; the last level is always written).  Bit-identical to `levels` calls of
 *   syg_decimate2_f32: with the CQT's 41-tap filter a workgroup carries its tile through all levels in LDS, so every
 *   level is written once and only x is read (the recursive octave walk of librosa.cqt, core/dsp.py:276-284, needs
 *   every level).  */
int syg_decimate2_chain_f32(const float* x, int64_t B, int64_t L, int64_t ldx, const float* taps, int ntaps, float scale,
                            int levels, float* const* y, const int64_t* ldy, void* stream);
 /* syg_cqt_octave_gemm_f32: the same octave as ONE framed matrix product on the matrix cores (n_fft 128 / 256 / 512):
 *   out[f, t] = sum_n y[t hop - n_fft/2 + n] * g_f[n],  g_f[n] = sum_k basis[f, k] exp(-2 pi i k n / n_fft)
 *   (the frequency-domain rows taken to the time domain by the caller, in float64 -- the same linear map).
 *   gpacked float32 [row tile][n_fft/16][4][64]: A operands of v_mfma_f32_16x16x4_f32, row r = 2 f + (0: re, 1: im):
 *   entry (mt, s, u, lane) = G[16 s + 4 (lane >> 4) + u][16 mt + (lane & 15)] (0 past 2 n_filt rows).  */
int syg_cqt_octave_gemm_f32(const float* y, int64_t B, int64_t L, int64_t ldy, int n_fft, int hop, int64_t T,
                            const float* gpacked, int n_filt, float* out, int64_t out_bstride, int row0, void* stream);
 /* syg_cqt_octave_bf16x3_f32: the same product with both operands split into three bfloat16 terms and the six products
 *   above 2^-24 accumulated in fp32 on v_mfma_f32_16x16x32_bf16 (fp32-equivalent: error <= 3 x 2^-24 |a b| per product,
 *   parity tests at 1e-5 like the fp32 form; 3/8 of its matrix-pipe time).  n_fft 128 / 256, n_filt <= 16.
 *   gsplit: bfloat16 [3 terms hi, mid, lo][row tile][n_fft/32][64 lanes][8]: entry (p, mt, s, lane, j) = term p of
 *   float32(G[32 s + 8 (lane >> 4) + j][16 mt + (lane & 15)]); 16-byte aligned.  */
int syg_cqt_octave_bf16x3_f32(const float* y, int64_t B, int64_t L, int64_t ldy, int n_fft, int hop, int64_t T,
                              const void* gsplit, int n_filt, float* out, int64_t out_bstride, int row0, void* stream);

/* The whole transform of compute_cqt (sygnals/core/dsp.py:231-289) in ONE launch for its usual shape -- hop_length 512, one
 * early decimation, n_oct <= 7 octaves of n_filt <= 16 filters at frame length 256 and hop 256 >> o that share one operand
 * table (sygnals_amd.ops.cqt_pack_bf16x3: [3][2][8][64] 16-byte entries): the stream is read once, every decimation level
 * lives in LDS only (syg_decimate2_chain_f32 + n_oct x syg_cqt_octave_bf16x3_f32 write and re-read them through HBM), the
 * same values to the rounding of the float32 sums (the decimator's symmetric taps are paired, the products' k range is
 * summed in four parts).  taps: the 41-tap half-band decimator (zero at the even offsets from its centre, symmetric);
 * scale: sqrt(2); row0_host[n_oct]: first output row of octave o; out [B, n_bins, T] complex (float pairs),
 * out_bstride in complex elements, T <= the smallest centred frame count over the octaves (1 + L_o / hop_o). */
int syg_cqt_fused_f32(const float* y, int64_t B, int64_t L, int64_t ldy, const float* taps, int ntaps, float scale,
                      const void* gsplit, int n_filt, int n_oct, const int32_t* row0_host, int64_t T, float* out,
                      int64_t out_bstride, void* stream);
int syg_cqt_octave_f32(const float* y, int64_t B, int64_t L, int64_t ldy, int n_fft, int hop, int64_t T,
                       const float* twiddle, const float* basis, int n_filt, const int32_t* hull_host,
                       float* out, int64_t out_bstride, int row0, void* stream);

/* ---------------------------------------------------------------------------------
 * FFT-backed 1-D operations; the host composes them with the power-of-two complex FFT above.
 *   syg_pack_rows_f32       (row stride ldx may be smaller than len: overlapping rows = frames of one signal)
 *                           out[r, i] = (x[r, j] - mean_r) * window[i] for i < min(len, n), 0 up to n;
 *                           j = reverse ? len-1-i : i; detrend 0 none, 1 mean_r, 2 least-squares line (float64 sums, need `work` of
 *                           syg_pack_rows_work_bytes(rows) bytes); cplx != 0 writes complex rows (value, 0).
 *                           A real row of n floats is at the same time the packed row z[m] = x[2m] + i x[2m+1]
 *                           of n/2 complex elements consumed by syg_rconv_spectrum_c64.
 *   syg_rconv_spectrum_c64  za [rows, H], zb [rows_b (1 or rows), H] (any H >= 2): length-H complex FFTs of two packed real
 *                           rows of 2H samples -> out [rows, H] (may alias za): the packed transform of their
 *                           circular convolution; its inverse length-H FFT, read as 2H floats, is the real
 *                           result.  scipy.signal.fftconvolve / correlate as called at sygnals/core/dsp.py:333,
 *                           :388 (correlation = convolution with the reversed second input).
 *   syg_analytic_mask_c64   in place X[r, k] *= h[k], h = scipy.signal.hilbert's mask (1, 2.., [1], 0..)
 *                           -- sygnals/core/transforms.py:146, sygnals/core/dsp.py:609
 *   syg_psd_onesided_f32    out[r, k] = scale * |X[r, k]|^2 * (1 for DC and the even-n Nyquist bin, else 2),
 *                           k <= n/2, from the full spectrum X [rows, n] -- scipy.signal.periodogram as called
 *                           at sygnals/core/dsp.py:484-492
 *   syg_col_mean_f32        acc[k] (float64 [F], caller-owned) = (first ? 0 : acc[k]) + sum_r in[r, k]; on the `last`
 *                           call out[k] = acc[k] / divisor: the average of per-segment periodograms for Welch with
 *                           a segment length that is not a power of two (segments = overlapping rows, row stride
 *                           ldx = nperseg - noverlap, through syg_pack_rows_f32 -> FFT -> syg_psd_onesided_f32)
 * ------------------------------------------------------------------------------- */
int64_t syg_pack_rows_work_bytes(int64_t rows);
int syg_pack_rows_f32(const float* x, int64_t rows, int64_t len, int64_t ldx, const float* window, int detrend,
                      int reverse, int cplx, float* out, int64_t n, void* work, void* stream);
/* syg_pack_rows_f32 for the frames of several clips at once: row r starts at
 * x + (r / rows_per_group) * group_stride + (r % rows_per_group) * ldx  (rows_per_group = 0: x + r * ldx). */
int syg_pack_frames_f32(const float* x, int64_t rows, int64_t len, int64_t ldx, int64_t rows_per_group,
                        int64_t group_stride, const float* window, int detrend, int reverse, int cplx, float* out,
                        int64_t n, void* work, void* stream);
int syg_rconv_spectrum_c64(const float* za, const float* zb, int64_t rows, int64_t rows_b, int64_t H, float* out,
                           void* stream);
int syg_analytic_mask_c64(float* X, int64_t rows, int64_t n, void* stream);
int syg_psd_onesided_f32(const float* X, int64_t rows, int64_t n, double scale, float* out, void* stream);
int syg_col_mean_f32(const float* in, int64_t rows, int64_t F, double* acc, int first, int last, double divisor,
                     float* out, void* stream);

/* ---------------------------------------------------------------------------------
 * Batched audio ingest: integer PCM frames as stored in a WAV file -> float32 mono clips, after the
 * host-to-device copy (half the PCIe bytes of float32 for 16-bit audio).  Replaces librosa.load's conversion
 * reached through sygnals/core/audio/io.py:84-90 (scale 2^-(bits-1), mono = mean over channels) and the mix-down of
 * sygnals/cli/features_cmd.py:66-68.
 *   pcm   [rows, frames, channels] interleaved; int16 (bits 16), int32 (bits 32; 24-bit files left-justified),
 *         uint8 (bits 8, offset 128); row stride ld in ELEMENTS
 *   out   [rows, frames] float32, row stride ldo:  out[r, i] = mean_c pcm[r, i, c] / 2^(bits-1)
 * ------------------------------------------------------------------------------- */
int syg_pcm_to_f32(const void* pcm, int bits, int64_t rows, int64_t frames, int channels, int64_t ld, float* out,
                   int64_t ldo, void* stream);

/* ---------------------------------------------------------------------------------
 * Feature formatting for ML on the device.  x is a [n, F] float32 matrix (frames x features);
 * statistics are float64 and NaN-aware like scikit-learn's fit (NaNs ignored; all-NaN column -> NaN).
 *   syg_col_stats_f32      out [5, F] float64: count, mean, population variance, min, max per column -- the fit
 *                          of StandardScaler / MinMaxScaler, sygnals/core/ml_utils/scaling.py:108-118
 *   syg_affine_cols_f32    out[r, c] = (x[r, c] - sub[c]) * mul[c] + add[c] (float64 arithmetic): every scaler's
 *                          transform, scaling.py:118, 133
 *   syg_col_quantiles_f32  out [nq, F] float64 = np.nanpercentile(x, 100 q, axis=0) (linear interpolation), q in
 *                          [0, 1]; n <= 32768 -- RobustScaler's centre and scale, scaling.py:114
 *   syg_zoom_f32           scipy.ndimage.zoom(img [H, W], order 0 | 1, mode='nearest') to [H2, W2] --
 *                          format_features_as_image, sygnals/core/ml_utils/formatters.py:296-316
 * ------------------------------------------------------------------------------- */
int syg_col_stats_f32(const float* x, int64_t n, int64_t F, double* out, void* stream);
int syg_affine_cols_f32(const float* x, int64_t n, int64_t F, const double* sub, const double* mul, const double* add,
                        float* out, void* stream);
int syg_col_quantiles_f32(const float* x, int64_t n, int64_t F, const double* q, int nq, double* out, void* stream);
int syg_zoom_f32(const float* img, int H, int W, int H2, int W2, int order, float* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SYGNALS_HIP_H */
