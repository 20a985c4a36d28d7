"""Device-backed mirror of sygnals/core/features/manager.py: extract_features (:78-445).

Same validation, frame-count rule (:149-157), frame times (:166-169), output naming
(mfcc_{i} :365-369, contrast_band_{i} / contrast_delta :337-343), float64 outputs and
DataFrame / dict_of_arrays formats.  The STFT is computed once per call and shared by every
requested feature (the reference caches it, tests/test_features_manager.py:137-160); for
frame_length 2048 everything from the samples to mel power, per-frame statistics and contrast
tail means comes out of ONE fused kernel launch.  `extract_features_batch` is the batched form
(leading clip axis) the reference lacks (manager.py:144-145).
"""
from __future__ import annotations

import logging
from typing import Any, Dict, List, Optional

import numpy as np
import torch

from ... import _tables as T
from ... import ops
from ..._lib import SygnalsHipError

logger = logging.getLogger(__name__)

_SPECTRUM_BASED = ("spectral_centroid", "spectral_bandwidth", "spectral_flatness", "spectral_rolloff",
                   "dominant_frequency")
_SPECTROGRAM_BASED = ("spectral_contrast",)
_MELSPEC_BASED = ("mfcc",)
# time-domain frame features (time_domain.py) + RMS / ZCR (audio/features.py): one device launch for all of them
_FRAME_BASED = ("mean_amplitude", "std_dev_amplitude", "skewness", "kurtosis", "peak_amplitude", "crest_factor",
                "signal_entropy", "zero_crossing_rate", "rms_energy")
# names the reference knows (manager.py:38-69) that are not offloaded (pitch-based placeholders)
_REFERENCE_ONLY = ("hnr", "jitter", "shimmer")
_DEVICE_FEATURES = set(_SPECTRUM_BASED) | set(_SPECTROGRAM_BASED) | set(_MELSPEC_BASED) | set(_FRAME_BASED)
_ALL_KNOWN_FEATURES = _DEVICE_FEATURES | set(_REFERENCE_ONLY)


class FeatureExtractionError(Exception):
    """Raised for non-recoverable errors during feature extraction (manager.py:72)."""


def mel_power_batch(y, sr, n_fft, hop, center, window, n_mels, fmin, fmax, power=2.0, win_length=None):
    """[B, L] device clips -> mel spectrogram [B, n_mels, T] of |STFT|^power."""
    if power == 2.0 and ops.fused_mel_ok(sr, n_fft, n_mels, fmin, fmax):
        mel, _, _ = ops.stft2048_mel(y, sr, hop, center, window, 2048 if win_length is None else win_length,
                                     n_mels, fmin, fmax)
        return mel
    if power not in (1.0, 2.0):
        raise SygnalsHipError("mel power must be 1.0 or 2.0 on the device")
    if power == 2.0:
        mel = ops.stft_mel_segments(y, sr, n_fft, hop, center, window, win_length, n_mels, fmin, fmax)
        if mel is not None:
            return mel
    if ops.fused_pow2_ok(n_fft, n_mels):
        return ops.stft_mel_pow2(y, sr, n_fft, hop, center, window, win_length, n_mels, fmin, fmax, int(power))
    X = ops.stft_any(y, n_fft, hop, center, window, win_length)
    P = ops.cabs_pow(X, int(power))
    cfg = ops.mel_config(sr, n_fft, n_mels, fmin, fmax)
    return ops.mel_dense(P, cfg.basis)


def extract_features_batch(y, sr: int, features: List[str], frame_length: int = 2048, hop_length: int = 512,
                           center: bool = True, window: str = "hann",
                           feature_params: Optional[Dict[str, Dict[str, Any]]] = None, to_host: bool = True):
    """Batched extract_features: y [B, L] (array or device tensor) -> {'time': [T], name: [B, T]}.

    With to_host=False the per-feature values stay on the device as float32 tensors.

    The per-feature semantics are the reference's (manager.py:242-397): features are processed in the order given;
    a failing feature is logged and skipped, the others still come out (:394-397); `spectral_bandwidth` requested
    without `spectral_centroid` also emits the centroid it depends on (:296-301); when the STFT has fewer frames
    than the frame-count rule (odd frame_length, :186-194) `time` is re-made from the STFT frame count and rows
    of another length are dropped by the final length check (:408-420).  What is NOT per feature: a missing HIP
    library or GPU raises before anything is computed (no CPU fallback).
    """
    feature_params = feature_params or {}
    if features == ["all"]:
        features = sorted(_DEVICE_FEATURES)
    unknown = [f for f in features if f not in _ALL_KNOWN_FEATURES]
    if unknown:
        raise ValueError(f"Unknown feature(s) requested: {unknown}. Available: {sorted(_ALL_KNOWN_FEATURES)}")
    ops.require_gpu()
    yd = y if hasattr(y, "is_cuda") else ops.to_device_f32(np.asarray(y))
    if yd.dim() != 2:
        raise ValueError("Batched input 'y' must have shape [B, L].")
    B, L = yd.shape
    Tn = ops.num_frames(L, frame_length, hop_length, center)
    if Tn <= 0:
        logger.warning("Signal is too short for the given frame/hop length and centering setting. No frames generated.")
        return {"time": np.array([], dtype=np.float64)}
    off = frame_length // 2 if center else 0
    res: Dict[str, Any] = {"time": ((np.arange(Tn) * hop_length + off) / float(sr)).astype(np.float64)}

    _bits = {"spectral_centroid": 1, "spectral_bandwidth": 2 | 1, "spectral_flatness": 4, "spectral_rolloff": 8,
             "dominant_frequency": 16}
    want_stats = 0
    for f in features:
        want_stats |= _bits.get(f, 0)
    want_contrast = "spectral_contrast" in features
    want_mfcc = "mfcc" in features
    mp = feature_params.get("mfcc", {})
    cp = feature_params.get("spectral_contrast", {})
    freqs = np.fft.rfftfreq(frame_length, 1.0 / sr)

    # ---- lazy device products, computed once and shared (the reference's _S_mag / _S_mel_log caches, :173-227)
    cache: Dict[str, Any] = {}

    def time_rows():
        if "tstats" not in cache:
            from .time_domain import time_features_frames
            want_time = [f for f in features if f in _FRAME_BASED]
            nb = int(feature_params.get("signal_entropy", {}).get("num_bins", 10))
            tstats = time_features_frames(yd, frame_length, hop_length, center, nb, want_time)
            # the reference frames the signal padded by frame_length // 2 on both sides (manager.py:268-271, librosa's
            # rms / zcr likewise): with an ODD frame_length that is one sample short of the last frame whenever hop
            # divides len(y); those rows come out one frame short and are NaN-padded (manager.py:378-386)
            t_rows = Tn
            if center:
                t_rows = min(Tn, 1 + (L + 2 * (frame_length // 2) - frame_length) // hop_length)
            cache["tstats"] = ({k: v[:, :t_rows] for k, v in tstats.items()}, t_rows)
        return cache["tstats"]

    def stft_products():
        """(mel power | None, stats | None, contrast tail means | None, STFT frame count): ONE fused launch for
        frame_length 2048; an STFT failure is remembered and re-raised for every feature that needs it."""
        if "stft" in cache:
            if isinstance(cache["stft"], Exception):
                raise cache["stft"]
            return cache["stft"]
        try:
            n_mels = mp.get("n_mels", 128)
            fmin, fmax = mp.get("fmin", 0.0), mp.get("fmax", sr / 2.0)
            power = mp.get("power", 2.0)
            roll = feature_params.get("spectral_rolloff", {}).get("roll_percent", 0.85)
            bw_p = float(feature_params.get("spectral_bandwidth", {}).get("p", 2))
            # a bad parameter of ONE feature must only cost that feature (frequency_domain.py:116, 314 raise inside the
            # per-feature try of manager.py:304-316): the shared launch runs with the default in its place and the
            # feature's own branch below raises
            cache["roll_ok"], cache["bw_ok"] = 0.0 <= roll <= 1.0, bw_p > 0
            if not cache["roll_ok"]:
                roll = 0.85
            if not cache["bw_ok"]:
                bw_p = 2.0
            cplan = None
            if want_contrast:
                cplan = T.contrast_plan(freqs, sr, cp.get("n_bands", 6), cp.get("fmin", 200.0), cp.get("quantile", 0.02))
            mel = stats = cpv = None
            n_mfcc, lifter = int(mp.get("n_mfcc", 13)), float(mp.get("lifter", 0.0))
            # the MFCC rows straight from the fused launch (the clip's mel matrix stays in LDS) when the request is the
            # default cepstrum (DCT-II ortho on the power mel spectrogram) -- mfcc is the only mel-based feature here
            # (ops.settings.one_launch_features = False: the mel launch + logmel_dct, for comparisons)
            std_mfcc = (want_mfcc and frame_length == 2048 and power == 2.0 and mp.get("dct_type", 2) == 2 and
                        mp.get("norm", "ortho") == "ortho" and 1 <= n_mfcc <= n_mels and ops.fused_waves() == 16 and
                        ops.settings.one_launch_features)
            one = None
            if std_mfcc and (want_stats or want_contrast) and lifter == 0.0:
                one = features_one_launch(yd, sr, hop_length, center, window, n_mels, fmin, fmax, n_mfcc, want_stats, roll, bw_p, cplan)
            if one is not None:
                cache["mfcc_dev"], stats, cpv = one
                t_stft = Tn
            elif (std_mfcc and not (want_stats or want_contrast) and ops.fused_mel_ok(sr, 2048, n_mels, fmin, fmax)
                  and ops.mfcc_fused_fits(n_mels, Tn, n_mfcc)):
                cache["mfcc_dev"] = ops.stft2048_mfcc(yd, sr, hop_length, center, window, n_mels, n_mfcc, fmin, fmax, lifter)[0]
                t_stft = Tn
            elif (not want_mfcc and (want_stats or want_contrast) and frame_length == 2048 and ops.stft2048_stats_fits(hop_length, yd.shape[1])):
                # no mel-based feature asked for: transform + row functions, nothing projected (syg_stft2048_stats_f32)
                stats, cpv = ops.stft2048_stats(yd, sr, hop_length, center, window, 2048, want_stats, roll, bw_p, cplan)
                t_stft = Tn
            elif power == 2.0 and ops.fused_mel_ok(sr, frame_length, n_mels if want_mfcc else 16, fmin, fmax):
                mel, stats, cpv = ops.stft2048_mel(yd, sr, hop_length, center, window, 2048, n_mels if want_mfcc else 16,
                                                   fmin, fmax, want_stats, roll, bw_p, cplan)
                t_stft = Tn
            elif (frame_length == 1024 and (want_stats or want_contrast) and
                  (not want_mfcc or (power in (1.0, 2.0) and ops.fused_pow2_ok(1024, n_mels)))):
                # frame length 1024 with spectral features (the reference's own manager tests: tests/test_features_manager.py:
                # 58-62, 167-174): the rows from the segment-sum kernel's launch, no spectrogram in HBM; the mel block from the
                # same launch where the filterbank has a piece table (power 2), else from the dense-matrix kernel
                seg_mel = want_mfcc and power == 2.0 and ops.w1024_segtab(sr, n_mels, fmin, fmax) is not None
                mel, stats, cpv = ops.stft_rows_w1024(yd, sr, hop_length, center, window, None, n_mels if seg_mel else None,
                                                      fmin, fmax, want_stats, roll, bw_p, cplan)
                if want_mfcc and not seg_mel:
                    mel = ops.stft_mel_pow2(yd, sr, 1024, hop_length, center, window, None, n_mels, fmin, fmax, int(power))
                t_stft = Tn
            elif (frame_length == 4096 and (want_stats or want_contrast) and
                  (not want_mfcc or (power == 2.0 and ops.w4096_segtab(sr, n_mels, fmin, fmax) is not None))):
                # frame length 4096: rows (and the mel block, power 2 with a piece table) from the one-wave-per-frame launch
                mel, stats, cpv = ops.stft_rows_w4096(yd, sr, hop_length, center, window, None, n_mels if want_mfcc else None,
                                                      fmin, fmax, want_stats, roll, bw_p, cplan)
                t_stft = Tn
            elif (frame_length in (512, 256) and (want_stats or want_contrast) and
                  (not want_mfcc or (power in (1.0, 2.0) and ops.fused_pow2_ok(frame_length, n_mels)))):
                # frame lengths 512 / 256 (256: the reference's short-signal tests): the rows from the segment-sum kernel's
                # transform (one launch, no spectrogram in HBM); the mel block, if an MFCC is asked for too, from its own launch
                stats, cpv = ops.stft_rows_wsmall(yd, sr, frame_length, hop_length, center, window, None, want_stats, roll, bw_p, cplan)
                if want_mfcc:
                    mel = ops.stft_mel_segments(yd, sr, frame_length, hop_length, center, window, None, n_mels, fmin, fmax) if power == 2.0 else None
                    if mel is None:
                        mel = ops.stft_mel_pow2(yd, sr, frame_length, hop_length, center, window, None, n_mels, fmin, fmax, int(power))
                t_stft = Tn
            elif (want_mfcc and not (want_stats or want_contrast) and power == 2.0 and
                  (mel := ops.stft_mel_segments(yd, sr, frame_length, hop_length, center, window, None, n_mels, fmin, fmax)) is not None):
                # only the mel spectrogram is needed: the segment-sum kernel of this frame length (1024 / 512 / 256 / 4096)
                t_stft = mel.shape[2]
            elif want_mfcc and not (want_stats or want_contrast) and power in (1.0, 2.0) and ops.fused_pow2_ok(frame_length, n_mels):
                # ... or the dense-matrix kernel of the other power-of-two frame lengths
                mel = ops.stft_mel_pow2(yd, sr, frame_length, hop_length, center, window, None, n_mels, fmin, fmax, int(power))
                t_stft = mel.shape[2]
            else:
                X = ops.stft_any(yd, frame_length, hop_length, center, window)
                t_stft, F = X.shape[1], X.shape[2]
                if want_stats or want_contrast:
                    mag = ops.cabs_pow(X, 1).reshape(B * t_stft, F)
                    if want_stats:
                        stats = ops.spectral_stats(mag, ops.to_device_f32(freqs), roll, bw_p).reshape(8, B, t_stft).permute(1, 0, 2)
                    if want_contrast:
                        R = int(cplan[0])
                        cpv = ops.contrast_pv(mag, cplan).reshape(2, R, B, t_stft).permute(2, 0, 1, 3).contiguous()
                if want_mfcc:
                    if power not in (1.0, 2.0):
                        raise SygnalsHipError("mel power must be 1.0 or 2.0 on the device")
                    P = ops.cabs_pow(X, int(power))
                    mel = ops.mel_dense(P, ops.mel_config(sr, frame_length, n_mels, fmin, fmax).basis)
        except Exception as e:  # mirrors manager.py:201-202, 225-226 (raised inside the per-feature try there too)
            cache["stft"] = FeatureExtractionError(f"Error calculating STFT: {e}")
            raise cache["stft"]
        if t_stft != len(res["time"]):
            # manager.py:186-194: the STFT's own frame count wins and `time` is re-made from it
            logger.warning(f"STFT frames ({t_stft}) mismatch calculated frame times ({len(res['time'])}). "
                           f"Adjusting frame count and times to match STFT output.")
            res["time"] = ((np.arange(t_stft) * hop_length + frame_length // 2) / float(sr)).astype(np.float64)
        cache["stft"] = (mel, stats, cpv, t_stft)
        return cache["stft"]

    def host(t):
        return t.cpu().numpy().astype(np.float64) if to_host else t

    def fit(name, v, n):
        """Pad with NaN / truncate a [B, len] row block to n frames (manager.py:376-387)."""
        have = v.shape[1]
        if have == n:
            return v
        logger.warning(f"Feature '{name}' array length ({have}) mismatch expected frames ({n}). Adjusting length "
                       f"(padding with NaN or truncating).")
        if have > n:
            return v[:, :n]
        if isinstance(v, np.ndarray):
            out = np.full((v.shape[0], n), np.nan, dtype=np.float64)
            out[:, :have] = v
            return out
        out = torch.full((v.shape[0], n), float("nan"), dtype=torch.float32, device=v.device)
        out[:, :have] = v
        return out

    processed = set()
    for name in features:
        if name in processed:
            continue
        try:
            items = []
            cur_T = len(res["time"])
            if name in _REFERENCE_ONLY:
                raise SygnalsHipError(f"feature '{name}' is not offloaded to the device backend")
            if name in _FRAME_BASED:
                rows, _ = time_rows()
                items.append((name, host(rows[name])))
            elif name in _SPECTRUM_BASED:
                _, stats, _, cur_T = stft_products()
                if name == "spectral_rolloff" and not cache["roll_ok"]:
                    raise ValueError("roll_percent must be between 0.0 and 1.0.")
                if name == "spectral_bandwidth" and not cache["bw_ok"]:
                    raise ValueError("Order 'p' for spectral bandwidth must be positive.")
                if name == "spectral_bandwidth" and "spectral_centroid" not in res:
                    logger.warning("Feature 'spectral_bandwidth' requires 'spectral_centroid', calculating it first.")
                    res["spectral_centroid"] = host(stats[:, 0])
                    processed.add("spectral_centroid")
                if name == "spectral_centroid":
                    items.append((name, host(stats[:, 0])))
                elif name == "spectral_bandwidth":
                    items.append((name, host(stats[:, 1])))
                elif name == "spectral_flatness":
                    items.append((name, host(stats[:, 2])))
                else:
                    row = 3 if name == "spectral_rolloff" else 4
                    if to_host:
                        items.append((name, freqs[stats[:, row].cpu().numpy().astype(np.int64)]))
                    else:
                        items.append((name, stats[:, row] * float(sr / frame_length)))
            elif name == "spectral_contrast":
                _, _, cpv, cur_T = stft_products()
                cdb = ops.contrast_db(cpv, linear=bool(cp.get("linear", False)))
                R = cdb.shape[1]
                for i in range(R - 1):
                    items.append((f"contrast_band_{i}", host(cdb[:, i])))
                items.append(("contrast_delta", host(cdb[:, R - 1])))
            elif name == "mfcc":
                mel, _, _, cur_T = stft_products()
                if "mfcc_dev" in cache:
                    mf = cache["mfcc_dev"]
                else:
                    _, mf = ops.logmel_dct(mel, mp.get("n_mfcc", 13), mp.get("dct_type", 2), mp.get("norm", "ortho"),
                                           float(mp.get("lifter", 0.0)), ref="max", keep_mel=True)
                for i in range(mf.shape[1]):
                    items.append((f"mfcc_{i}", host(mf[:, i])))
            for nm, arr in items:
                res[nm] = fit(nm, arr, cur_T)
                processed.add(nm)
            processed.add(name)
        except Exception as e:                       # manager.py:394-397: log and continue with the next feature
            logger.error(f"Error extracting feature '{name}': {e}")

    # manager.py:401-420: only rows whose length equals the final frame count survive
    final_T = len(res["time"])
    out: Dict[str, Any] = {"time": res["time"]}
    for nm, arr in res.items():
        if nm == "time":
            continue
        if arr.shape[1] == final_T:
            out[nm] = arr
        else:
            logger.error(f"Internal Error: Final length mismatch for feature '{nm}' ({arr.shape[1]} vs {final_T}). "
                         f"Skipping feature in output.")
    return out


def extract_features(y, sr: int, features: List[str], frame_length: int = 2048, hop_length: int = 512,
                     center: bool = True, window: str = "hann",
                     feature_params: Optional[Dict[str, Dict[str, Any]]] = None, output_format: str = "dataframe"):
    """Single-signal extract_features with the reference's signature and return formats."""
    y = np.asarray(y)
    if features != ["all"]:
        unknown = [f for f in features if f not in _ALL_KNOWN_FEATURES]
        if unknown:
            raise ValueError(f"Unknown feature(s) requested: {unknown}. Available: {sorted(_ALL_KNOWN_FEATURES)}")
    if y.ndim != 1:
        raise ValueError("Input audio signal 'y' must be a 1D array.")
    if output_format not in ("dataframe", "dict_of_arrays"):
        raise ValueError(f"Unsupported output format: {output_format}. Choose 'dataframe' or 'dict_of_arrays'.")
    import pandas as pd
    empty = pd.DataFrame() if output_format == "dataframe" else {"time": np.array([], dtype=np.float64)}
    if ops.num_frames(len(y), frame_length, hop_length, center) <= 0:
        logger.warning("Signal is too short for the given frame/hop length and centering setting. No frames generated.")
        return empty
    res = extract_features_batch(y[None, :].astype(np.float32), sr, features, frame_length, hop_length, center, window,
                                 feature_params)
    final = {"time": res.pop("time")}
    for k, v in res.items():
        final[k] = np.asarray(v[0], dtype=np.float64)
    if len(final) <= 1:
        logger.warning("No features were successfully extracted or passed final checks.")
        return empty if output_format == "dataframe" else {"time": final["time"]}
    if output_format == "dict_of_arrays":
        return final
    idx = pd.to_timedelta(final.pop("time"), unit="s")
    df = pd.DataFrame(final, index=idx)
    df.index.name = "time"
    return df


def features_one_launch(y, sr, hop_length, center, window, n_mels, fmin, fmax, n_mfcc, smask, roll_percent, bw_p, cplan):
    """MFCC rows + statistics rows + contrast tail means of [B, L] device clips from ONE launch
    (syg_stft2048_features_tri_f32: frame_length 2048, power 2, DCT-II ortho, no lifter, ref = max, top_db 80 -- the
    defaults of manager.py:219-227 / cepstral.py:20-120), or None when the shape has no segment-sum form (the caller then
    takes the mel launch).  Returns (mfcc [B, n_mfcc, T], stats [B, 8, T] | None, contrast_pv [B, 2, R, T] | None)."""
    import ctypes as C
    from ..._lib import check, lib
    B, L = y.shape
    Tn = ops.num_frames(L, 2048, hop_length, center)
    if not (hop_length <= 512 and ops.fused_waves() == 16 and 1 <= n_mfcc <= n_mels <= 127 and (smask or cplan is not None)):
        return None
    cfg = ops.mel_config(sr, 2048, n_mels, fmin, fmax, waves=16)
    if cfg.segtab is None or not lib().syg_stft2048_mfcc_tri_fits(int(n_mels), int(Tn), int(n_mfcc)):
        return None
    if y.stride(1) != 1:
        y = y.contiguous()
    mf = torch.empty((B, n_mfcc, Tn), dtype=torch.float32, device=y.device)
    stats = torch.zeros((B, 8, Tn), dtype=torch.float32, device=y.device) if smask else None
    cpv = cph = None
    if cplan is not None:
        cph = np.ascontiguousarray(cplan, np.int32)
        cpv = torch.empty((B, 2, int(cph[0]), Tn), dtype=torch.float32, device=y.device)
    dct = ops._cached(("dct", n_mfcc, n_mels, 2, "ortho"), lambda: ops._dev(T.dct_matrix(n_mfcc, n_mels, 2, "ortho")))
    rc = lib().syg_stft2048_features_tri_f32(
        ops._ptr(y), B, L, y.stride(0), hop_length, int(center), Tn, ops._ptr(ops.window_dev(window, 2048, 2048)),
        ops._ptr(ops.twiddle_dev(2048)), ops._ptr(cfg.segtab), int(cfg.segtab.numel()), n_mels, ops._ptr(dct), n_mfcc, None,
        1e-10, 80.0, 1, 1.0, float(sr), float(roll_percent), float(bw_p), (smask | 32) if smask else 1, ops._ptr(stats),
        cph.ctypes.data_as(C.c_void_p) if cph is not None else None, ops._ptr(cpv), ops._ptr(mf), n_mfcc,
        C.c_void_p(ops._stream_ptr()))
    check(rc, "syg_stft2048_features_tri_f32")
    return mf, stats, cpv


# ------------------------------------------------------------------ config C4: the packed per-clip feature block
# feature_block's default where the filterbank has a piece table: ONE fused launch with the segment-sum projection
# (syg_stft2048_features_tri_f32, MODE 7) + the small rows kernel -- 701 against 731 us per 2048 clips for the mel ->
# feature_block pair on one box
TRI_FEATURES_DEFAULT = True


def feature_block(y, sr: int, hop_length: int = 512, n_mels: int = 40, n_mfcc: int = 13, roll_percent: float = 0.85,
                  n_bands: int = 6, fmin_contrast: float = 200.0, quantile: float = 0.02, out=None, one_launch=None,
                  projection: str = "auto", tri_waves: int = 16):
    """BASELINE config C4 on the device: y [B, L] float32 device clips -> [B, n_mfcc + 2 + (n_bands + 1), T] float32,
    rows = mfcc_0..mfcc_{n-1}, spectral_centroid (Hz), spectral_rolloff (Hz), contrast_band_0..{n_bands-1},
    contrast_delta -- the columns `extract_features(["mfcc", "spectral_centroid", "spectral_rolloff",
    "spectral_contrast"])` returns (manager.py:289-371), one dense block per rank for the gather to rank 0
    ([B/W, 22, 94]).  frame_length 2048 (the fused kernel)."""
    ops.require_gpu()
    B, L = y.shape
    Tn = ops.num_frames(L, 2048, hop_length, True)
    freqs = np.fft.rfftfreq(2048, 1.0 / sr)
    cplan = ops._cached(("cplan", float(sr), n_bands, float(fmin_contrast), float(quantile)),
                        lambda: T.contrast_plan(freqs, sr, n_bands, fmin_contrast, quantile))
    R = int(cplan[0])
    rows = n_mfcc + 2 + R
    if out is None:
        out = torch.empty((B, rows, Tn), dtype=torch.float32, device=y.device)
    dct = ops._cached(("dct", n_mfcc, n_mels, 2, "ortho"), lambda: ops._dev(T.dct_matrix(n_mfcc, n_mels, 2, "ortho")))
    import ctypes as C
    from ..._lib import check, lib
    st = C.c_void_p(ops._stream_ptr())
    # one_launch: everything from ONE fused launch -- samples in; MFCC rows straight into the head of the block,
    # statistics rows and contrast tail means out (the mel matrix stays in LDS) -- then the small kernel that turns those
    # into the block's other rows: syg_stft2048_features_tri_f32 (each wave projects its own row by segment sums; the clip
    # epilogue runs on waves that have no frame).  Needs a two-pass piece table for the filterbank and the clip's mel
    # matrix in LDS (n_mels = 40: yes; 128: no).  one_launch=None picks it where it applies, True insists, False takes
    # the two launches below.  (Round 3's matrix form of the one launch lost to the two launches, 735 vs 724 us per 2048
    # clips, and is gone.)
    cfg = ops.mel_config(sr, 2048, n_mels, 0.0, None, waves=16)
    tri_ok = (cfg.segtab is not None and hop_length <= 512 and ops.fused_waves() == 16
              and bool(lib().syg_stft2048_mfcc_tri_fits(int(n_mels), int(Tn), int(n_mfcc))))
    if one_launch not in (None, True, False, "segments"):
        raise ValueError("one_launch must be None, True, False or 'segments'")
    if one_launch in (True, "segments") and not tri_ok:
        raise SygnalsHipError("feature_block: no one-launch form for this shape (two-pass piece table + the clip's mel matrix in LDS)")
    if one_launch is None:
        one_launch = tri_ok and TRI_FEATURES_DEFAULT
    if one_launch:
        stats = torch.empty((B, 8, Tn), dtype=torch.float32, device=y.device)       # (only the rows read below are written)
        cpv = torch.empty((B, 2, R, Tn), dtype=torch.float32, device=y.device)
        rc = lib().syg_stft2048_features_tri_f32(
            ops._ptr(y), B, L, y.stride(0), hop_length, 1, Tn, ops._ptr(ops.window_dev("hann", 2048, 2048)),
            ops._ptr(ops.twiddle_dev(2048)), ops._ptr(cfg.segtab), int(cfg.segtab.numel()), n_mels, ops._ptr(dct), n_mfcc, None,
            1e-10, 80.0, 1, 1.0, float(sr), float(roll_percent), 2.0, 1 | 8 | 32, ops._ptr(stats),
            np.ascontiguousarray(cplan, np.int32).ctypes.data_as(C.c_void_p), ops._ptr(cpv), ops._ptr(out), rows, st)
        check(rc, "syg_stft2048_features_tri_f32")
        rc = lib().syg_feature_block_f32(None, B, n_mels, Tn, None, n_mfcc, 1e-10, 80.0, ops._ptr(stats), float(sr) / 2048.0,
                                         ops._ptr(cpv), R, 1e-10, 80.0, ops._ptr(out), st)
        check(rc, "syg_feature_block_f32")
        return out
    # two launches: mel + rows (matrix form, or -- projection="segments" -- the tile form of the segment-sum projection with
    # `tri_waves` waves per workgroup), then the block kernel
    mel, stats, cpv = ops.stft2048_mel(y, sr, hop_length, True, "hann", 2048, n_mels, 0.0, None, 1 | 8 | 32, roll_percent,
                                       2.0, cplan, projection, tri_waves)
    rc = lib().syg_feature_block_f32(ops._ptr(mel), B, n_mels, Tn, ops._ptr(dct), n_mfcc, 1e-10, 80.0, ops._ptr(stats),
                                     float(sr) / 2048.0, ops._ptr(cpv), R, 1e-10, 80.0, ops._ptr(out), st)
    check(rc, "syg_feature_block_f32")
    return out


def feature_block_dominant(y, sr, hop_length, n_mels, n_mfcc):
    """(name, callable, feature rows it carries) of the dominant kernel of feature_block (bench.py's roofline leg)."""
    freqs = np.fft.rfftfreq(2048, 1.0 / sr)
    cplan = T.contrast_plan(freqs, sr)
    B, L = y.shape
    Tn = ops.num_frames(L, 2048, hop_length, True)
    cfg = ops.mel_config(sr, 2048, n_mels, 0.0, None, waves=16)
    from ..._lib import check, lib
    if (TRI_FEATURES_DEFAULT and cfg.segtab is not None and hop_length <= 512 and ops.fused_waves() == 16
            and bool(lib().syg_stft2048_mfcc_tri_fits(int(n_mels), int(Tn), int(n_mfcc)))):
        import ctypes as C
        R = int(cplan[0])
        rows = n_mfcc + 2 + R
        out = torch.empty((B, rows, Tn), dtype=torch.float32, device=y.device)
        stats = torch.empty((B, 8, Tn), dtype=torch.float32, device=y.device)
        cpv = torch.empty((B, 2, R, Tn), dtype=torch.float32, device=y.device)
        dct = ops._cached(("dct", n_mfcc, n_mels, 2, "ortho"), lambda: ops._dev(T.dct_matrix(n_mfcc, n_mels, 2, "ortho")))
        cph = np.ascontiguousarray(cplan, np.int32)
        args = (ops._ptr(y), B, L, y.stride(0), hop_length, 1, Tn, ops._ptr(ops.window_dev("hann", 2048, 2048)),
                ops._ptr(ops.twiddle_dev(2048)), ops._ptr(cfg.segtab), int(cfg.segtab.numel()), n_mels, ops._ptr(dct), n_mfcc,
                None, 1e-10, 80.0, 1, 1.0, float(sr), 0.85, 2.0, 1 | 8 | 32, ops._ptr(stats), cph.ctypes.data_as(C.c_void_p),
                ops._ptr(cpv), ops._ptr(out), rows)

        def run():
            check(lib().syg_stft2048_features_tri_f32(*args, C.c_void_p(ops._stream_ptr())), "syg_stft2048_features_tri_f32")
            return out, stats, cpv, cph
        return ("stft2048_kernel<16,2,7> (16 waves, staged tiles, per-wave mel projection by segment sums, clip-resident MFCC "
                "+ centroid + rolloff + contrast tail means)", run, n_mfcc + 2 + 2 * R)
    return ("stft2048_kernel<16,2,1> (16 waves, staged tiles, mel + centroid + rolloff + contrast tail means)",
            lambda: ops.stft2048_mel(y, sr, hop_length, True, "hann", 2048, n_mels, 0.0, None, 1 | 8, 0.85, 2.0, cplan),
            n_mels + 3 + 2 * int(cplan[0]))
