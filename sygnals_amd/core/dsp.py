"""Device-backed mirror of sygnals/core/dsp.py (FFT/IFFT, STFT, Welch, windowing).

Same signatures, dtypes (float64 / complex128 out) and error behaviour as the reference:
compute_fft :40-112, compute_ifft :114-162, compute_stft :167-229, compute_psd_welch :495-560,
apply_window :641-691.  Arithmetic runs in fp32 on the device (parity gate 1e-5, peak-relative).
"""
from __future__ import annotations

import logging
from typing import Optional, Tuple, Union

import numpy as np
from scipy.signal import get_window

from .. import ops
from .._lib import SygnalsHipError

logger = logging.getLogger(__name__)


def _c128(t) -> np.ndarray:
    a = t.cpu().numpy().astype(np.float64)
    return a[..., 0] + 1j * a[..., 1]


def _symmetric_window(window_type: str, n: int) -> np.ndarray:
    try:
        return get_window(window_type, n, fftbins=False)      # symmetric, dsp.py:676
    except ValueError as e:
        logger.error("Invalid window type '%s': %s", window_type, e)
        raise ValueError(f"Invalid window type '{window_type}'.") from e


def apply_window(data, window_type: str = "hann") -> np.ndarray:
    data = np.asarray(data)
    if data.ndim != 1:
        raise ValueError("Input data must be a 1D array.")
    w = _symmetric_window(window_type, data.shape[0])
    x = ops.to_device_f32(data[None, :])
    z = ops.pack_real(x, data.shape[0], ops._dev(w.astype(np.float32)))
    return z[0, :, 0].cpu().numpy().astype(np.float64)


def compute_fft(data, fs: Union[int, float] = 1.0, n: Optional[int] = None, window: Optional[str] = "hann"
                ) -> Tuple[np.ndarray, np.ndarray]:
    data = np.asarray(data)
    if data.ndim != 1:
        raise ValueError("Input data must be a 1D array.")
    wdev = None
    if window:
        try:
            wdev = ops._dev(_symmetric_window(window, data.shape[0]).astype(np.float32))
        except ValueError as e:
            raise ValueError(f"Invalid window type '{window}': {e}") from e
    if n is None:
        n = data.shape[0]
    z = ops.pack_real(ops.to_device_f32(data[None, :]), int(n), wdev)
    spectrum = _c128(ops.fft_any(z))[0]
    freqs = np.fft.fftfreq(int(n), d=1 / fs)
    return freqs.astype(np.float64, copy=False), spectrum.astype(np.complex128, copy=False)


def compute_ifft(spectrum, n: Optional[int] = None) -> np.ndarray:
    spectrum = np.asarray(spectrum)
    if spectrum.ndim != 1:
        raise ValueError("Input spectrum must be a 1D array.")
    if n is None:
        n = spectrum.shape[0]
    n = int(n)
    s = np.zeros(n, dtype=np.complex128)           # ifft(x, n) truncates / zero-pads the spectrum
    m = min(n, spectrum.shape[0])
    s[:m] = spectrum[:m]
    z = ops._dev(np.stack([s.real, s.imag], axis=-1).astype(np.float32)[None])
    t = _c128(ops.fft_any(z, inverse=True))[0]
    peak = float(np.max(np.abs(t))) if t.size else 0.0
    imag_max = float(np.max(np.abs(t.imag))) if t.size else 0.0
    # reference warns above 1e-9 absolute (float64); the fp32 device floor is ~1e-7 of the peak
    if imag_max > max(1e-9, 1e-5 * peak):
        logger.warning("Significant imaginary part found in IFFT result (max abs: %.2e). "
                       "Input spectrum might not have conjugate symmetry.", imag_max)
    return np.real(t).astype(np.float64, copy=False)


def compute_stft(y, n_fft: int = 2048, hop_length: Optional[int] = None, win_length: Optional[int] = None,
                 window: str = "hann", center: bool = True, pad_mode: str = "constant") -> np.ndarray:
    y = np.asarray(y)
    if y.ndim != 1:
        raise ValueError("Input data must be a 1D array.")
    if win_length is None:
        win_length = n_fft
    if hop_length is None:
        hop_length = win_length // 4
    if center and pad_mode != "constant":
        y = np.pad(y, n_fft // 2, mode=pad_mode)   # data movement only; zeros are padded on the device
        center = False
    if not center and y.shape[0] < n_fft:
        raise ValueError(f"n_fft={n_fft} is too large for input signal of length={y.shape[0]}")
    X = ops.stft_any(ops.to_device_f32(y[None, :]), n_fft, hop_length, center, window, win_length)
    return np.ascontiguousarray(_c128(X)[0].T).astype(np.complex128, copy=False)   # [F, T] like librosa


def compute_cqt(y, sr: int, hop_length: Optional[int] = 512, fmin: Optional[float] = None, n_bins: int = 84,
                bins_per_octave: int = 12, **kwargs) -> np.ndarray:
    """Constant-Q transform, complex128 [n_bins, 1 + len(y)//hop_length] (dsp.py:231-289).

    Accepted librosa keyword arguments: tuning (default 0.0), filter_scale (1), sparsity (0.01); the fixed choices
    norm=1, window='hann', scale=True, pad_mode='constant' may be passed but not changed.  The decimation filter
    differs from librosa's soxr_hq resampler (see sygnals_amd/_cqt.py).
    """
    y = np.asarray(y)
    if y.ndim != 1:
        raise ValueError("Input data must be a 1D array.")
    fixed = {"norm": 1, "window": "hann", "scale": True, "pad_mode": "constant"}
    for k, v in fixed.items():
        if k in kwargs and kwargs.pop(k) != v:
            raise SygnalsHipError(f"compute_cqt: only {k}={v!r} runs on the device")
    kwargs.pop("res_type", None)
    tuning = kwargs.pop("tuning", 0.0)
    filter_scale = kwargs.pop("filter_scale", 1.0)
    sparsity = kwargs.pop("sparsity", 0.01)
    if kwargs:
        raise TypeError(f"compute_cqt: unsupported arguments {sorted(kwargs)}")
    if tuning is None:
        raise SygnalsHipError("compute_cqt: tuning estimation (tuning=None) is not offloaded; pass a number")
    out = ops.cqt(ops.to_device_f32(y[None, :]), sr, int(hop_length), fmin, n_bins, bins_per_octave, tuning,
                  filter_scale, sparsity)
    return _c128(out)[0].astype(np.complex128, copy=False)


def compute_psd_welch(x, fs: float = 1.0, window: str = "hann", nperseg: Optional[int] = None,
                      noverlap: Optional[int] = None, nfft: Optional[int] = None,
                      detrend: Union[str, bool] = "constant", scaling: str = "density"
                      ) -> Tuple[np.ndarray, np.ndarray]:
    x = np.asarray(x)
    if x.ndim != 1:
        raise ValueError("Input data must be a 1D array.")
    f, p = welch_batch(ops.to_device_f32(x[None, :]), fs, window, nperseg, noverlap, nfft, detrend, scaling)
    return f, p[0].cpu().numpy().astype(np.float64)


def welch_batch(x, fs=1.0, window="hann", nperseg=None, noverlap=None, nfft=None, detrend="constant",
                scaling="density"):
    """[B, L] device tensor -> (freqs float64 [F], Pxx device tensor [B, F]); scipy.signal.welch rules."""
    L = x.shape[1]
    if nperseg is None:
        nperseg = 256
    if nperseg > L:
        logger.warning("nperseg = %d is greater than input length = %d, using nperseg = %d", nperseg, L, L)
        nperseg = L
    nperseg = int(nperseg)
    if nfft is None:
        nfft = nperseg
    elif nfft < nperseg:
        raise ValueError("nfft must be greater than or equal to nperseg.")
    if noverlap is None:
        noverlap = nperseg // 2
    if noverlap >= nperseg:
        raise ValueError("noverlap must be less than nperseg.")
    if scaling not in ("density", "spectrum"):
        raise ValueError(f"Unknown scaling: {scaling!r}")
    if detrend not in ("constant", False, None, "none"):
        raise SygnalsHipError("welch: only detrend='constant' or False run on the device")
    if not ops.is_pow2(int(nfft)) or nfft < 8 or nfft > 16384:
        raise SygnalsHipError(f"welch: nfft={nfft} must be a power of two in [8, 16384] on the device")
    w = get_window(window, nperseg)
    scale = 1.0 / (fs * (w * w).sum()) if scaling == "density" else 1.0 / w.sum() ** 2
    p = ops.welch(x, nperseg, int(noverlap), int(nfft), w, detrend == "constant", scale)
    return np.fft.rfftfreq(int(nfft), 1 / fs).astype(np.float64), p
