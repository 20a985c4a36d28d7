"""Device-backed mirror of sygnals/core/features/frequency_domain.py.

Single-frame functions keep the reference signatures (:24-386) and edge-case constants
(empty -> 0.0, all-zero -> 0.0 / last frequency); `spectral_stats_frames` is the batched
form the feature manager uses (one wave per frame instead of one Python call per frame,
manager.py:304-316).
"""
from __future__ import annotations

import logging
from typing import Any, Dict, Optional

import numpy as np

from ... import _tables as T
from ... import ops

logger = logging.getLogger(__name__)

ST_CENTROID, ST_BANDWIDTH, ST_FLATNESS, ST_ROLLOFF_BIN, ST_DOMINANT_BIN = 0, 1, 2, 3, 4


def _check_pair(magnitude_spectrum, frequencies):
    m = np.asarray(magnitude_spectrum)
    f = np.asarray(frequencies)
    if m.shape != f.shape:
        raise ValueError(f"Spectrum shape {m.shape} and frequencies shape {f.shape} must match.")
    return m, f


def _stats_1frame(m, f, roll_percent=0.85, p=2.0):
    if np.any(m < 0):
        logger.warning("Input magnitude_spectrum contains negative values. Using absolute values.")
    st = ops.spectral_stats(ops.to_device_f32(m[None, :]), ops.to_device_f32(f), roll_percent, p)
    return st[:, 0].cpu().numpy().astype(np.float64)


def spectral_centroid(magnitude_spectrum, frequencies) -> np.float64:
    m, f = _check_pair(magnitude_spectrum, frequencies)
    if m.size == 0:
        return np.float64(0.0)
    return np.float64(_stats_1frame(m, f)[ST_CENTROID])


def spectral_bandwidth(magnitude_spectrum, frequencies, centroid: Optional[np.float64] = None, p: int = 2
                       ) -> np.float64:
    m, f = _check_pair(magnitude_spectrum, frequencies)
    if p <= 0:
        raise ValueError("Order 'p' for spectral bandwidth must be positive.")
    if m.size == 0:
        return np.float64(0.0)
    if centroid is not None:
        # deviation around a caller-supplied centre: a centroid of |f - c|^p weights, i.e. the
        # same weighted-mean kernel applied to the transformed axis
        dev = np.abs(np.asarray(f, dtype=np.float64) - float(centroid)) ** p
        st = _stats_1frame(m, dev)
        total = st[5]
        return np.float64(0.0) if total < np.finfo(np.float64).eps else np.float64(max(st[ST_CENTROID], 0.0) ** (1.0 / p))
    return np.float64(_stats_1frame(m, f, p=float(p))[ST_BANDWIDTH])


def spectral_flatness(magnitude_spectrum) -> np.float64:
    m = np.asarray(magnitude_spectrum)
    if m.size == 0:
        return np.float64(0.0)
    return np.float64(_stats_1frame(m, np.zeros_like(m, dtype=np.float64))[ST_FLATNESS])


def spectral_rolloff(magnitude_spectrum, frequencies, roll_percent: float = 0.85) -> np.float64:
    m, f = _check_pair(magnitude_spectrum, frequencies)
    if not 0.0 <= roll_percent <= 1.0:
        raise ValueError("roll_percent must be between 0.0 and 1.0.")
    if m.size == 0:
        return np.float64(0.0)
    return np.float64(f[int(_stats_1frame(m, f, roll_percent=roll_percent)[ST_ROLLOFF_BIN])])


def dominant_frequency(magnitude_spectrum, frequencies) -> np.float64:
    m, f = _check_pair(magnitude_spectrum, frequencies)
    if m.size == 0:
        return np.float64(0.0)
    return np.float64(f[int(_stats_1frame(m, f)[ST_DOMINANT_BIN])])


def spectral_contrast(S, sr: int, n_bands: int = 6, fmin: float = 200.0, freqs=None, **kwargs: Any) -> np.ndarray:
    """[n_bands + 1, T] float64, librosa.feature.spectral_contrast semantics (frequency_domain.py:147-212)."""
    S = np.asarray(S)
    if S.ndim != 2:
        raise ValueError("Input S must be a 2D spectrogram (frequency x time).")
    if np.any(S < 0):
        logger.warning("Input spectrogram S contains negative values. Using absolute values.")
    quantile = kwargs.pop("quantile", 0.02)
    linear = kwargs.pop("linear", False)
    if kwargs:
        raise TypeError(f"unsupported spectral_contrast arguments: {sorted(kwargs)}")
    if freqs is None:
        freqs = np.fft.rfftfreq(2 * (S.shape[0] - 1), 1.0 / sr)
    freqs = np.atleast_1d(np.asarray(freqs, dtype=np.float64))
    if freqs.ndim != 1 or len(freqs) != S.shape[0]:
        raise ValueError(f"freq.shape={freqs.shape} does not match dimensions of S.shape={S.shape}")
    plan = T.contrast_plan(freqs, sr, n_bands, fmin, quantile)
    mag = ops.to_device_f32(np.ascontiguousarray(S.T))               # frame-major [T, F]
    pv = ops.contrast_pv(mag, plan)                                   # [2, R, T]
    return ops.contrast_db(pv[None], linear=bool(linear))[0].cpu().numpy().astype(np.float64)


FREQUENCY_DOMAIN_FEATURES: Dict[str, Any] = {
    "spectral_centroid": spectral_centroid,
    "spectral_bandwidth": spectral_bandwidth,
    "spectral_flatness": spectral_flatness,
    "spectral_rolloff": spectral_rolloff,
    "dominant_frequency": dominant_frequency,
}
