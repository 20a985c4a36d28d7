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
    if n_fft == 2048 and power == 2.0 and n_mels <= 16 * ops.fused_waves():
        mel, _, _ = ops.stft2048_mel(y, sr, hop, center, window, 2048 if win_length is None else win_length,
                                     n_mels, fmin, fmax)
        return mel
    if power not in (1.0, 2.0):
        raise SygnalsHipError("mel power must be 1.0 or 2.0 on the device")
    X = ops.stft_any(y, n_fft, hop, center, window, win_length)
    P = ops.cabs_pow(X, int(power))
    cfg = ops.mel_config(sr, n_fft, n_mels, fmin, fmax)
    return ops.mel_dense(P, cfg.basis)


def extract_features_batch(y, sr: int, features: List[str], frame_length: int = 2048, hop_length: int = 512,
                           center: bool = True, window: str = "hann",
                           feature_params: Optional[Dict[str, Dict[str, Any]]] = None, to_host: bool = True):
    """Batched extract_features: y [B, L] (array or device tensor) -> {'time': [T], name: [B, T]}.

    With to_host=False the per-feature values stay on the device as float32 tensors.
    """
    feature_params = feature_params or {}
    if features == ["all"]:
        features = sorted(_DEVICE_FEATURES)
    unknown = [f for f in features if f not in _ALL_KNOWN_FEATURES]
    if unknown:
        raise ValueError(f"Unknown feature(s) requested: {unknown}. Available: {sorted(_ALL_KNOWN_FEATURES)}")
    host_only = [f for f in features if f in _REFERENCE_ONLY]
    if host_only:
        raise SygnalsHipError(f"feature(s) {host_only} are not offloaded to the device backend")
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
    n_mels = mp.get("n_mels", 128)
    fmin, fmax = mp.get("fmin", 0.0), mp.get("fmax", sr / 2.0)
    power = mp.get("power", 2.0)
    roll = feature_params.get("spectral_rolloff", {}).get("roll_percent", 0.85)
    bw_p = float(feature_params.get("spectral_bandwidth", {}).get("p", 2))
    if not 0.0 <= roll <= 1.0:
        raise ValueError("roll_percent must be between 0.0 and 1.0.")
    if bw_p <= 0:
        raise ValueError("Order 'p' for spectral bandwidth must be positive.")
    freqs = np.fft.rfftfreq(frame_length, 1.0 / sr)
    cplan = None
    cp = feature_params.get("spectral_contrast", {})
    if want_contrast:
        cplan = T.contrast_plan(freqs, sr, cp.get("n_bands", 6), cp.get("fmin", 200.0), cp.get("quantile", 0.02))

    tstats = None
    want_time = [f for f in features if f in _FRAME_BASED]
    if want_time:
        from .time_domain import time_features_frames
        nb = int(feature_params.get("signal_entropy", {}).get("num_bins", 10))
        tstats = time_features_frames(yd, frame_length, hop_length, center, nb, want_time)
        # the reference frames the signal padded by frame_length // 2 on both sides (manager.py:268-271, librosa's
        # rms / zcr likewise): with an ODD frame_length that is one sample short of the last frame whenever hop
        # divides len(y), and the missing value is NaN-padded (manager.py:378-386).  Same observable result here.
        if center:
            t_ref = 1 + (L + 2 * (frame_length // 2) - frame_length) // hop_length
            if t_ref < Tn:
                for v in tstats.values():
                    v[:, t_ref:] = float("nan")
    need_stft = bool(want_stats or want_contrast or want_mfcc)

    mel = stats = cpv = None
    t_stft = Tn
    try:
        if not need_stft:
            pass
        elif frame_length == 2048 and power == 2.0 and n_mels <= 16 * ops.fused_waves():
            mel, stats, cpv = ops.stft2048_mel(yd, sr, hop_length, center, window, 2048, n_mels if want_mfcc else 16,
                                               fmin, fmax, want_stats, roll, bw_p, cplan)
        else:
            X = ops.stft_any(yd, frame_length, hop_length, center, window)
            F = X.shape[2]
            if X.shape[1] < Tn:
                # odd frame_length: the centred STFT has one frame less than the manager's frame count whenever hop
                # divides len(y); the reference NaN-pads rows that come out short (manager.py:378-386), as above
                t_stft = X.shape[1]
                X = torch.cat([X, torch.zeros((B, Tn - t_stft, F, 2), dtype=X.dtype, device=X.device)], dim=1)
            if want_stats or want_contrast:
                mag = ops.cabs_pow(X, 1).reshape(B * Tn, F)
                if want_stats:
                    stats = ops.spectral_stats(mag, ops.to_device_f32(freqs), roll, bw_p).reshape(8, B, Tn).permute(1, 0, 2)
                if want_contrast:
                    R = int(cplan[0])
                    cpv = ops.contrast_pv(mag, cplan).reshape(2, R, B, Tn).permute(2, 0, 1, 3).contiguous()
            if want_mfcc:
                if power not in (1.0, 2.0):
                    raise SygnalsHipError("mel power must be 1.0 or 2.0 on the device")
                P = ops.cabs_pow(X, int(power))
                mel = ops.mel_dense(P, ops.mel_config(sr, frame_length, n_mels, fmin, fmax).basis)
    except SygnalsHipError:
        raise
    except Exception as e:  # mirrors manager.py:201-202, 225-226
        raise FeatureExtractionError(f"Error calculating STFT: {e}")

    def host(t):
        return t.cpu().numpy().astype(np.float64) if to_host else t

    def short(v):                                   # STFT-based rows past the last frame the STFT has
        if t_stft < Tn:
            v = v.astype(np.float64) if isinstance(v, np.ndarray) else v.clone().float()
            v[:, t_stft:] = float("nan")
        return v

    for name in features:
        if name in res:
            continue
        if name in _FRAME_BASED:
            res[name] = host(tstats[name])
        elif name in _SPECTRUM_BASED:
            if name == "spectral_centroid":
                res[name] = short(host(stats[:, 0]))
            elif name == "spectral_bandwidth":
                res[name] = short(host(stats[:, 1]))
            elif name == "spectral_flatness":
                res[name] = short(host(stats[:, 2]))
            elif name == "spectral_rolloff":
                idx = stats[:, 3].cpu().numpy().astype(np.int64)
                res[name] = short(freqs[idx] if to_host else stats[:, 3] * float(sr / frame_length))
            elif name == "dominant_frequency":
                idx = stats[:, 4].cpu().numpy().astype(np.int64)
                res[name] = short(freqs[idx] if to_host else stats[:, 4] * float(sr / frame_length))
        elif name == "spectral_contrast":
            cdb, host_c = ops.contrast_db(cpv, linear=bool(cp.get("linear", False))), host
            R = cdb.shape[1]
            for i in range(R - 1):
                res[f"contrast_band_{i}"] = short(host_c(cdb[:, i]))
            res["contrast_delta"] = short(host_c(cdb[:, R - 1]))
        elif name == "mfcc":
            _, mf = ops.logmel_dct(mel, mp.get("n_mfcc", 13), mp.get("dct_type", 2), mp.get("norm", "ortho"),
                                   float(mp.get("lifter", 0.0)), ref="max")
            for i in range(mf.shape[1]):
                res[f"mfcc_{i}"] = short(host(mf[:, i]))
    return res


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
