"""ctypes binding of libsygnals_hip.so (the C ABI declared in include/sygnals_hip.h).

The product path has NO CPU fallback: `lib()` raises if the shared library has not been
built, and every op raises if no GPU is present.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# SYGNALS_AMD_LIB: development only (timeline builds, A/B of a kernel against an earlier one) -- a library built somewhere else
LIB_PATH = os.environ.get("SYGNALS_AMD_LIB") or os.path.join(_HERE, "lib", "libsygnals_hip.so")

_p = C.c_void_p
_i = C.c_int
_l = C.c_int64
_f = C.c_float
_d = C.c_double

# name -> (restype, argtypes); must list every symbol declared in include/sygnals_hip.h
SIGNATURES = {
    "syg_abi_version": (_i, []),
    "syg_build_variant": (_i, []),
    "syg_last_error": (C.c_char_p, []),
    "syg_set_option": (_i, [_i, _i]),
    "syg_get_option": (_i, [_i]),
    "syg_stft2048_mel_f32": (_i, [_p, _l, _l, _l, _i, _i, _l, _p, _p, _p, _p, _i, _p, _f, _f, _f, _i, _p, _p, _p, _p]),
    "syg_stft2048_c2c_f32": (_i, [_p, _l, _l, _l, _i, _i, _l, _p, _p, _p, _p]),
    "syg_stft2048_mfcc_fits": (_i, [_i, _l, _i]),
    "syg_stft2048_mfcc_tri_fits": (_i, [_i, _l, _i]),
    "syg_stft2048_mfcc_tri_f32": (_i, [_p, _l, _l, _l, _i, _i, _l, _p, _p, _p, _i, _i, _p, _i, _p, _f, _f, _i, _f, _p, _p]),
    "syg_stft2048_mfcc_f32": (_i, [_p, _l, _l, _l, _i, _i, _l, _p, _p, _p, _p, _i, _p, _i, _p, _f, _f, _i, _f, _p, _p, _p]),
    "syg_stft2048_features_tri_f32": (_i, [_p, _l, _l, _l, _i, _i, _l, _p, _p, _p, _i, _i, _p, _i, _p, _f, _f, _i, _f, _f, _f, _f, _i, _p, _p,
                                           _p, _p, _i, _p]),
    "syg_stft2048_mel_tri_f32": (_i, [_p, _l, _l, _l, _i, _i, _l, _p, _p, _p, _i, _i, _p, _f, _f, _f, _i, _p, _p, _p, _i, _p]),
    "syg_stft2048_stats_f32": (_i, [_p, _l, _l, _l, _i, _i, _l, _p, _p, _f, _f, _f, _i, _p, _p, _p, _p]),
    "syg_stft_mel_w4096_f32": (_i, [_p, _l, _l, _l, _i, _i, _l, _p, _p, _p, _i, _i, _p, _p]),
    "syg_stft_mel_w1024_seg_f32": (_i, [_p, _l, _l, _l, _i, _i, _l, _p, _p, _p, _i, _i, _p, _p]),
    "syg_stft_rows_w1024_f32": (_i, [_p, _l, _l, _l, _i, _i, _l, _p, _p, _p, _i, _i, _p, _f, _f, _f, _i, _p, _p, _p, _p]),
    "syg_stft_mel_wseg_small_f32": (_i, [_p, _l, _l, _l, _i, _i, _i, _l, _p, _p, _p, _i, _i, _p, _p]),
    "syg_stft_rows_wsmall_f32": (_i, [_p, _l, _l, _l, _i, _i, _i, _l, _p, _p, _f, _f, _f, _i, _p, _p, _p, _p]),
    "syg_stft_rows_w4096_f32": (_i, [_p, _l, _l, _l, _i, _i, _l, _p, _p, _p, _i, _i, _p, _f, _f, _f, _i, _p, _p, _p, _p]),
    "syg_cqt_fused_f32": (_i, [_p, _l, _l, _l, _p, _i, _f, _p, _i, _i, _p, _l, _p, _l, _p]),
    "syg_mel_mfcc_f32": (_i, [_p, _l, _i, _l, _p, _i, _p, _f, _f, _i, _f, _p, _p]),
    "syg_fft_pow2_strided_ex_f32": (_i, [_p, _p, _l, _l, _i, _i, _p, _l, _l, _l, _l, _l, _l, _l, _f, _i, _l, _l, _p]),
    "syg_fft_mixed_strided_ex_f32": (_i, [_p, _p, _l, _l, _i, _i, _p, _l, _l, _l, _l, _l, _l, _l, _f, _i, _l, _l, _p]),
    "syg_frame_stats_f32": (_i, [_p, _l, _l, _l, _i, _i, _i, _l, _i, _i, _p, _p]),
    "syg_rms_from_spec_f32": (_i, [_p, _l, _i, _i, _p, _p]),
    "syg_feature_block_f32": (_i, [_p, _l, _i, _l, _p, _i, _f, _f, _p, _f, _p, _i, _f, _f, _p, _p]),
    "syg_logmel_dct_f32": (_i, [_p, _l, _i, _l, _p, _i, _p, _f, _f, _i, _f, _p, _p, _p]),
    "syg_fft_pow2_c2c_f32": (_i, [_p, _p, _l, _i, _i, _p, _p]),
    "syg_fft_pow2_strided_c2c_f32": (_i, [_p, _p, _l, _l, _i, _i, _p, _l, _l, _l, _l, _l, _l, _l, _f, _p]),
    "syg_cmul_c64": (_i, [_p, _p, _p, _l, _l, _i, _p]),
    "syg_pack_real_c64": (_i, [_p, _l, _l, _l, _p, _p, _l, _p]),
    "syg_stft_pow2_c2c_f32": (_i, [_p, _l, _l, _l, _i, _i, _i, _l, _p, _p, _p, _p]),
    "syg_cabs_pow_f32": (_i, [_p, _l, _i, _p, _p]),
    "syg_mel_dense_f32": (_i, [_p, _l, _l, _i, _p, _i, _p, _p]),
    "syg_stft_mel_pow2_f32": (_i, [_p, _l, _l, _l, _i, _i, _i, _l, _p, _p, _p, _i, _i, _i, _p, _p]),
    "syg_stft_mfcc_pow2_fits": (_i, [_i, _i, _l, _i]),
    "syg_stft_mfcc_pow2_f32": (_i, [_p, _l, _l, _l, _i, _i, _i, _l, _p, _p, _p, _i, _i, _p, _i, _p, _f, _f, _i, _f, _p, _p, _p]),
    "syg_spectral_stats_f32": (_i, [_p, _l, _i, _p, _f, _f, _p, _p]),
    "syg_contrast_pv_f32": (_i, [_p, _l, _i, _p, _p, _p]),
    "syg_contrast_db_f32": (_i, [_p, _l, _i, _l, _f, _f, _p, _p]),
    "syg_decimate2_f32": (_i, [_p, _l, _l, _l, _p, _i, _f, _p, _l, _p]),
    "syg_decimate2_chain_f32": (_i, [_p, _l, _l, _l, _p, _i, _f, _i, _p, _p, _p]),
    "syg_cqt_octave_f32": (_i, [_p, _l, _l, _l, _i, _i, _l, _p, _p, _i, _p, _p, _l, _i, _p]),
    "syg_cqt_octave_bf16x3_f32": (_i, [_p, _l, _l, _l, _i, _i, _l, _p, _i, _p, _l, _i, _p]),
    "syg_cqt_octave_gemm_f32": (_i, [_p, _l, _l, _l, _i, _i, _l, _p, _i, _p, _l, _i, _p]),
    "syg_sosfiltfilt_work_bytes": (_l, [_l, _l, _i, _i]),
    "syg_sosfiltfilt_f32": (_i, [_p, _l, _l, _l, _p, _p, _i, _i, _p, _l, _p, _p]),
    "syg_welch_work_bytes": (_l, [_l, _i]),
    "syg_welch_f32": (_i, [_p, _l, _l, _l, _i, _i, _i, _p, _p, _i, _d, _p, _p, _p]),
    "syg_pack_rows_work_bytes": (_l, [_l]),
    "syg_pack_rows_f32": (_i, [_p, _l, _l, _l, _p, _i, _i, _i, _p, _l, _p, _p]),
    "syg_pack_frames_f32": (_i, [_p, _l, _l, _l, _l, _l, _p, _i, _i, _i, _p, _l, _p, _p]),
    "syg_rconv_spectrum_c64": (_i, [_p, _p, _l, _l, _l, _p, _p]),
    "syg_analytic_mask_c64": (_i, [_p, _l, _l, _p]),
    "syg_psd_onesided_f32": (_i, [_p, _l, _l, _d, _p, _p]),
    "syg_col_mean_f32": (_i, [_p, _l, _l, _p, _i, _i, _d, _p, _p]),
    "syg_fft_mixed_plan": (_i, [_l, _p, _i]),
    "syg_fft_mixed_strided_c2c_f32": (_i, [_p, _p, _l, _l, _i, _i, _p, _l, _l, _l, _l, _l, _l, _l, _f, _p]),
    "syg_col_stats_f32": (_i, [_p, _l, _l, _p, _p]),
    "syg_affine_cols_f32": (_i, [_p, _l, _l, _p, _p, _p, _p, _p]),
    "syg_col_quantiles_f32": (_i, [_p, _l, _l, _p, _i, _p, _p]),
    "syg_zoom_f32": (_i, [_p, _i, _i, _i, _i, _i, _p, _p]),
    "syg_pcm_to_f32": (_i, [_p, _i, _l, _l, _i, _l, _p, _l, _p]),
}

_lib = None


class SygnalsHipError(RuntimeError):
    """Raised when the HIP library is missing or a C-ABI call reports an error."""


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise SygnalsHipError(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(sygnals_amd has no CPU fallback)")
        h = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(h, name)          # AttributeError if the library does not export it
            fn.restype = res
            fn.argtypes = args
        ver = h.syg_abi_version()
        if ver != 1:
            raise SygnalsHipError(f"libsygnals_hip.so ABI version {ver} != 1")
        var = h.syg_build_variant()
        if var != 0 and os.environ.get("SYGNALS_AMD_ALLOW_VARIANT") != str(var):
            raise SygnalsHipError(f"{LIB_PATH} is a development variant (syg_build_variant() = {var}: stamps or ablations, results not for use); "
                                  "rebuild the product library with build_lib.sh")
        _lib = h
    return _lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = lib().syg_last_error()
        raise SygnalsHipError(f"{what} failed (rc={rc}): {msg.decode() if msg else ''}")
