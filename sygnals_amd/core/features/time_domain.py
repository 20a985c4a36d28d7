"""Device-backed mirror of sygnals/core/features/time_domain.py (:23-227).

The seven functions keep the reference's single-frame signatures and edge-case constants (empty frame -> 0.0,
fewer than 2 / 4 points -> 0.0 for skewness / kurtosis); the arithmetic of a frame runs on the device
(`syg_frame_stats_f32`, one wave per frame, float64 accumulation).  `time_features_frames` is the batched form the
feature manager uses instead of the per-frame Python loop of manager.py:264-286.
"""
from __future__ import annotations

import logging
from typing import Any, Dict

import numpy as np

from ... import ops

logger = logging.getLogger(__name__)

_ROWS = {name: i for i, name in enumerate(ops.FS_ROWS)}


def _one(frame, row: int, num_bins: int = 10) -> np.float64:
    f = np.asarray(frame, dtype=np.float64)
    st = ops.frame_stats(ops.to_device_f32(f[None, :]), f.size, 1, False, num_bins, 1 << row)
    return np.float64(st[0, row, 0].item())


def mean_amplitude(frame) -> np.float64:
    return np.float64(0.0) if np.size(frame) == 0 else _one(frame, 0)


def std_dev_amplitude(frame) -> np.float64:
    return np.float64(0.0) if np.size(frame) == 0 else _one(frame, 1)


def skewness(frame) -> np.float64:
    return np.float64(0.0) if np.size(frame) < 2 else _one(frame, 2)


def kurtosis_val(frame) -> np.float64:
    return np.float64(0.0) if np.size(frame) < 4 else _one(frame, 3)


def peak_amplitude(frame) -> np.float64:
    return np.float64(0.0) if np.size(frame) == 0 else _one(frame, 4)


def crest_factor(frame) -> np.float64:
    return np.float64(0.0) if np.size(frame) == 0 else _one(frame, 5)


def signal_entropy(frame, num_bins: int = 10) -> np.float64:
    if np.size(frame) < 2 or num_bins < 1:
        return np.float64(0.0)
    return _one(frame, 6, num_bins)


TIME_DOMAIN_FEATURES: Dict[str, Any] = {
    "mean_amplitude": mean_amplitude,
    "std_dev_amplitude": std_dev_amplitude,
    "skewness": skewness,
    "kurtosis": kurtosis_val,
    "peak_amplitude": peak_amplitude,
    "crest_factor": crest_factor,
    "signal_entropy": signal_entropy,
}


def time_features_frames(y, frame_length: int = 2048, hop_length: int = 512, center: bool = True,
                         num_bins: int = 10, names=None):
    """y [B, L] device tensor -> {name: [B, T] float32 device tensor} for the requested time-domain rows
    (plus 'rms_energy' / 'zero_crossing_rate')."""
    names = list(ops.FS_ROWS) if names is None else list(names)
    mask = 0
    for n in names:
        mask |= 1 << _ROWS[n]
    st = ops.frame_stats(y, frame_length, hop_length, center, num_bins, mask)
    return {n: st[:, _ROWS[n]] for n in names}
