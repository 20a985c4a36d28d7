"""Device-backed mirror of sygnals/core/features/cepstral.py (mfcc :20-120)."""
from __future__ import annotations

import logging
from typing import Any, Dict, Optional

import numpy as np

from ... import ops

logger = logging.getLogger(__name__)


def mfcc(y=None, sr: Optional[int] = None, S=None, n_mfcc: int = 13, dct_type: int = 2, norm: Optional[str] = "ortho",
         lifter: float = 0.0, **kwargs: Any) -> np.ndarray:
    """MFCCs [n_mfcc, T] float64 from a signal `y` or a log-power mel spectrogram `S`.

    With `y`, follows librosa.feature.mfcc(y=..): power_to_db(melspectrogram(y, sr, **kwargs))
    with ref=1.0 (the manager path uses ref=np.max instead, see manager.py:223).
    """
    if S is None and y is None:
        raise ValueError("Either audio time series 'y' or Mel spectrogram 'S' must be provided.")
    if S is None and sr is None:
        raise ValueError("Sampling rate 'sr' must be provided when calculating MFCCs from time series 'y'.")
    if S is not None and y is not None:
        logger.warning("Both 'y' and 'S' provided for MFCC calculation. Using pre-computed 'S'.")
    if lifter < 0:
        raise ValueError(f"MFCC lifter={lifter} must be a non-negative number")
    if S is not None:
        S = np.asarray(S)
        if S.ndim != 2:
            raise ValueError("S must be a 2D log-power mel spectrogram (n_mels x frames).")
        L = ops.to_device_f32(S[None])
        _, mf = ops.logmel_dct(L, n_mfcc, dct_type, norm, float(lifter), ref="db")   # DCT stage only
        return mf[0].cpu().numpy().astype(np.float64)
    y = np.asarray(y)
    if y.ndim != 1:
        raise ValueError("Input audio signal 'y' must be a 1D array.")
    n_fft = kwargs.get("n_fft", 2048)
    hop = kwargs.get("hop_length", 512)
    from .manager import mel_power_batch
    mel = mel_power_batch(ops.to_device_f32(y[None, :]), sr, n_fft, hop, kwargs.get("center", True),
                          kwargs.get("window", "hann"), kwargs.get("n_mels", 128), kwargs.get("fmin", 0.0),
                          kwargs.get("fmax"), kwargs.get("power", 2.0), win_length=kwargs.get("win_length"))
    _, mf = ops.logmel_dct(mel, n_mfcc, dct_type, norm, float(lifter), ref=1.0)
    return mf[0].cpu().numpy().astype(np.float64)


CEPSTRAL_FEATURES: Dict[str, Any] = {"mfcc": mfcc}
