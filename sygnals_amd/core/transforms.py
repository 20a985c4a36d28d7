"""Device-backed mirror of the FFT-based part of sygnals/core/transforms.py: hilbert_transform :119-151.

The wavelet (PyWavelets) and numerical-Laplace functions of that module are outside the hot path.
"""
from __future__ import annotations

import numpy as np

from .. import ops
from .dsp import _c128, analytic_batch


def hilbert_transform(data) -> np.ndarray:
    """Analytic signal x + i*H(x) (scipy.signal.hilbert), complex128."""
    data = np.asarray(data)
    if data.ndim != 1:
        raise ValueError("Input data must be a 1D array.")
    if data.size == 0:
        raise ValueError("N must be positive.")
    return _c128(analytic_batch(ops.to_device_f32(data[None, :])))[0].astype(np.complex128, copy=False)
