"""Device-backed mirror of sygnals/core/audio/features.py: zero_crossing_rate (:26-71) and rms_energy (:73-131).

The reference forwards both to librosa (`librosa.feature.zero_crossing_rate`: EDGE padding, threshold 1e-10,
first sample of a frame never counts; `librosa.feature.rms`: zero padding, or from a magnitude spectrogram with DC /
Nyquist halved).  Here one wave per frame computes them (`syg_frame_stats_f32`, `syg_rms_from_spec_f32`).
The pitch-based placeholders of that file (hnr, jitter, shimmer) are not offloaded.
"""
from __future__ import annotations

import logging
from typing import Any, Optional

import numpy as np

from ... import ops

logger = logging.getLogger(__name__)

_ROW_RMS, _ROW_ZCR = 7, 8


def zero_crossing_rate(y, frame_length: int = 2048, hop_length: int = 512, center: bool = True, **kwargs: Any):
    y = np.asarray(y)
    if y.ndim != 1:
        raise ValueError("Input audio data must be a 1D array.")
    if kwargs:
        raise TypeError(f"zero_crossing_rate: unsupported librosa arguments on the device backend: {sorted(kwargs)}")
    logger.debug(f"Calculating Zero Crossing Rate: frame={frame_length}, hop={hop_length}, center={center}")
    st = ops.frame_stats(ops.to_device_f32(y[None, :]), frame_length, hop_length, center, mask=1 << _ROW_ZCR)
    return st[0, _ROW_ZCR].cpu().numpy().astype(np.float64)


def rms_energy(y=None, *, S=None, frame_length: int = 2048, hop_length: int = 512, center: bool = True,
               pad_mode: str = "constant", **kwargs: Any):
    if S is None and y is None:
        raise ValueError("Either audio time series 'y' or magnitude spectrogram 'S' must be provided.")
    if y is not None and np.asarray(y).ndim != 1:
        raise ValueError("Input audio data 'y' must be a 1D array.")
    if S is not None and np.asarray(S).ndim != 2:
        raise ValueError("Input spectrogram 'S' must be a 2D array.")
    if kwargs:
        raise TypeError(f"rms_energy: unsupported librosa arguments on the device backend: {sorted(kwargs)}")
    logger.debug(f"Calculating RMS Energy: frame={frame_length}, hop={hop_length}, center={center}")
    if S is not None:                       # librosa ignores y when S is given
        Sm = np.asarray(S)                   # (the kernel squares the values: |S|^2 needs no abs on the host)
        if Sm.shape[0] != frame_length // 2 + 1:
            raise ValueError(f"Since S.shape[-2] is {Sm.shape[0]}, frame_length is expected to be "
                             f"{2 * Sm.shape[0] - 2} or {2 * Sm.shape[0] - 1}; found {frame_length}")
        out = ops.rms_from_spec(ops.to_device_f32(np.ascontiguousarray(Sm.T)), frame_length)
        return out.cpu().numpy().astype(np.float64)
    if pad_mode != "constant":
        raise ValueError("rms_energy: only pad_mode='constant' is offloaded")
    st = ops.frame_stats(ops.to_device_f32(np.asarray(y)[None, :]), frame_length, hop_length, center, mask=1 << _ROW_RMS)
    return st[0, _ROW_RMS].cpu().numpy().astype(np.float64)
