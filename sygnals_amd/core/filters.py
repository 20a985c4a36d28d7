"""Butterworth SOS design (host, float64) and zero-phase application (device).

Mirror of sygnals/core/filters.py: design_butterworth_sos :22-83 (validation strings are
pinned by the reference's tests/test_filters.py:73-87), apply_sos_filter :85-115 and the
four convenience filters :119-211.  The design stays on the host in float64 (24
coefficients); filtering runs in sosfilt.hip for single signals and [B, L] batches.
"""
from __future__ import annotations

import logging
from typing import Tuple, Union

import numpy as np
from scipy.signal import butter

from .. import _tables as T
from .. import ops

logger = logging.getLogger(__name__)


def design_butterworth_sos(cutoff: Union[float, Tuple[float, float]], fs: float, order: int, filter_type: str):
    nyquist = 0.5 * fs
    if isinstance(cutoff, (int, float)):
        if not 0 < cutoff < nyquist:
            raise ValueError(f"Cutoff frequency ({cutoff} Hz) must be strictly between 0 and Nyquist ({nyquist} Hz).")
        wn = cutoff / nyquist
    elif isinstance(cutoff, tuple) and len(cutoff) == 2:
        low, high = cutoff
        if not (0 < low < nyquist and 0 < high < nyquist):
            raise ValueError(f"Both low ({low} Hz) and high ({high} Hz) cutoff frequencies must be strictly "
                             f"between 0 and Nyquist ({nyquist} Hz).")
        if low >= high:
            raise ValueError(f"Low cutoff ({low} Hz) must be less than high cutoff ({high} Hz).")
        wn = (low / nyquist, high / nyquist)
    else:
        raise TypeError("cutoff must be a float (for low/high pass) or a tuple of two floats (for band pass/stop).")
    logger.debug("Designing %s-order Butterworth %s filter, cutoff %s Hz, fs %s Hz", order, filter_type, cutoff, fs)
    return butter(order, wn, btype=filter_type, analog=False, output="sos").astype(np.float64, copy=False)


def _steady_state(sos: np.ndarray) -> np.ndarray:
    """Per-section DF2T state for a unit step input (what scipy.signal.sosfilt_zi returns)."""
    zi = np.zeros((sos.shape[0], 2))
    gain = 1.0
    for s, (b0, b1, b2, a0, a1, a2) in enumerate(sos):
        b0, b1, b2, a1, a2 = b0 / a0, b1 / a0, b2 / a0, a1 / a0, a2 / a0
        # z0 = b1 - a1*y + z1, z1 = b2 - a2*y with y = b0 + z0 at steady state for x = 1
        y = (b0 + b1 + b2) / (1.0 + a1 + a2)
        z1 = b2 - a2 * y
        z0 = b1 - a1 * y + z1
        zi[s] = gain * np.array([z0, z1])
        gain *= y
    return zi


def apply_sos_filter_batch(sos, data):
    """[B, L] float32 device tensor (or array) -> filtered device tensor, sosfiltfilt semantics."""
    sos = np.asarray(sos, dtype=np.float64)
    if sos.ndim != 2 or sos.shape[1] != 6:
        raise ValueError("Input sos must be a 2D array with shape (n_sections, 6).")
    x = data if hasattr(data, "is_cuda") else ops.to_device_f32(np.asarray(data))
    if x.dim() != 2:
        raise ValueError("Batched input must have shape [B, L].")
    return ops.sosfiltfilt(x, sos, _steady_state(sos), T.butter_padlen(sos))


def apply_sos_filter(sos, data):
    data = np.asarray(data)
    sos = np.asarray(sos, dtype=np.float64)
    if data.ndim != 1:
        raise ValueError("Input data for filtering must be a 1D array.")
    if sos.ndim != 2 or sos.shape[1] != 6:
        raise ValueError("Input sos must be a 2D array with shape (n_sections, 6).")
    logger.debug("Applying SOS filter (zero-phase) with %d sections.", sos.shape[0])
    y = apply_sos_filter_batch(sos, data[None, :].astype(np.float32))
    return y[0].cpu().numpy().astype(np.float64)


def low_pass_filter(data, cutoff: float, fs: float, order: int = 5):
    return apply_sos_filter(design_butterworth_sos(cutoff, fs, order, "lowpass"), data)


def high_pass_filter(data, cutoff: float, fs: float, order: int = 5):
    return apply_sos_filter(design_butterworth_sos(cutoff, fs, order, "highpass"), data)


def band_pass_filter(data, low_cutoff: float, high_cutoff: float, fs: float, order: int = 5):
    return apply_sos_filter(design_butterworth_sos((low_cutoff, high_cutoff), fs, order, "bandpass"), data)


def band_stop_filter(data, low_cutoff: float, high_cutoff: float, fs: float, order: int = 5):
    return apply_sos_filter(design_butterworth_sos((low_cutoff, high_cutoff), fs, order, "bandstop"), data)
