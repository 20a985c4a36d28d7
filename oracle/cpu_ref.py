"""cpu_ref -- float64 CPU restatement of the sygnals feature-extraction hot path.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Never imported by sygnals_amd.

Every function cites the reference file:line (relative to /root/reference) whose
behaviour it restates.  Where the reference delegates to SciPy/NumPy the same
SciPy/NumPy routine is called here (it *is* the reference's arithmetic).  Where
the reference delegates to librosa (not vendored, not installed; pinned only as
``librosa>=0.10.0`` in pyproject.toml:38) the published librosa algorithm is
restated in NumPy.

PARITY STATUS
  pinned   : design_butterworth_sos, apply_sos_filter/*_pass_filter, compute_fft,
             compute_ifft, apply_window, compute_psd_welch, spectral_centroid,
             spectral_bandwidth, spectral_flatness, spectral_rolloff,
             dominant_frequency, the seven time-domain frame functions
             (mean/std/skewness/kurtosis/peak/crest/entropy), apply_convolution,
             compute_correlation, compute_autocorrelation, compute_psd_periodogram,
             hilbert_transform, amplitude_envelope(hilbert), apply_scaling,
             format_feature_sequences, format_features_as_image  -- checked against
             tests/golden/ref_*.npz, which were produced by executing the
             reference's own functions.
  UNPINNED : stft, mel_filterbank, power_to_db, melspectrogram, mfcc,
             spectral_contrast, frames_to_time, cqt, rms_energy, zero_crossing_rate  ("parity unpinned": no
             runnable librosa, no numeric golden values in the reference's
             tests).  Anchors: librosa's documented examples (mel_frequencies,
             hz_to_mel, mel_to_hz, fft_frequencies), closed-form KATs, and
             scipy.signal.ShortTimeFFT / scipy.fft.dct cross-checks
             (tests/test_oracle_*.py).
"""
from __future__ import annotations

import numpy as np
import scipy.fft
import scipy.signal

EPS64 = np.finfo(np.float64).eps  # frequency_domain.py:21


# --------------------------------------------------------------------------
# a1  STFT  (dsp.py:167-229, manager.py:184-187 -> librosa.stft)
# --------------------------------------------------------------------------
def fft_window(window, win_length: int, n_fft: int) -> np.ndarray:
    """Periodic window of win_length, zero-padded (centred) to n_fft.

    librosa.stft: ``get_window(window, win_length, fftbins=True)`` then
    ``util.pad_center(.., size=n_fft)``.
    """
    if isinstance(window, (str, tuple, float, int)):
        w = scipy.signal.get_window(window, win_length, fftbins=True)
    else:
        w = np.asarray(window, dtype=np.float64)
        if w.shape != (win_length,):
            raise ValueError(f"window array must have length {win_length}")
    w = np.asarray(w, dtype=np.float64)
    if win_length < n_fft:
        lpad = (n_fft - win_length) // 2
        w = np.pad(w, (lpad, n_fft - win_length - lpad))
    return w


def num_frames(length: int, n_fft: int, hop_length: int, center: bool) -> int:
    """manager.py:149-157 (same rule librosa's framing yields)."""
    if center:
        return 1 + length // hop_length
    if length >= n_fft:
        return 1 + (length - n_fft) // hop_length
    return 0


def frame_signal(y: np.ndarray, n_fft: int, hop_length: int, center: bool,
                 pad_mode: str = "constant") -> np.ndarray:
    """Return frames as a (T, n_fft) float64 array (librosa.util.frame, transposed)."""
    y = np.asarray(y, dtype=np.float64)
    if center:
        y = np.pad(y, n_fft // 2, mode=pad_mode)
    if y.shape[0] < n_fft:
        raise ValueError(f"Input is too short (n={y.shape[0]}) for frame_length={n_fft}")
    T = 1 + (y.shape[0] - n_fft) // hop_length
    idx = np.arange(n_fft)[None, :] + hop_length * np.arange(T)[:, None]
    return y[idx]


def stft(y, n_fft=2048, hop_length=None, win_length=None, window="hann",
         center=True, pad_mode="constant") -> np.ndarray:
    """complex128 [1 + n_fft//2, T], freq-major, like librosa.stft."""
    y = np.asarray(y, dtype=np.float64)
    if y.ndim != 1:
        raise ValueError("Input data must be a 1D array.")  # dsp.py:211-212
    if win_length is None:
        win_length = n_fft
    if hop_length is None:
        hop_length = win_length // 4
    w = fft_window(window, win_length, n_fft)
    frames = frame_signal(y, n_fft, hop_length, center, pad_mode)
    return np.fft.rfft(frames * w[None, :], axis=1).T.astype(np.complex128)


def fft_frequencies(sr, n_fft) -> np.ndarray:
    """librosa.fft_frequencies == rfftfreq (manager.py:199)."""
    return np.fft.rfftfreq(n_fft, 1.0 / sr)


def frames_to_time(frames, sr, hop_length, n_fft=None) -> np.ndarray:
    """manager.py:166-169: (i*hop + n_fft//2)/sr when n_fft is given."""
    off = int(n_fft // 2) if n_fft is not None else 0
    return (np.asarray(frames) * hop_length + off) / float(sr)


# --------------------------------------------------------------------------
# a3  mel filterbank (manager.py:219-222 -> librosa.feature.melspectrogram)
# --------------------------------------------------------------------------
_F_SP = 200.0 / 3
_MIN_LOG_HZ = 1000.0
_MIN_LOG_MEL = _MIN_LOG_HZ / _F_SP          # 15.0
_LOGSTEP = np.log(6.4) / 27.0


def hz_to_mel(f, htk=False):
    f = np.asanyarray(f, dtype=np.float64)
    if htk:
        return 2595.0 * np.log10(1.0 + f / 700.0)
    m = f / _F_SP
    big = f >= _MIN_LOG_HZ
    with np.errstate(divide="ignore", invalid="ignore"):
        m = np.where(big, _MIN_LOG_MEL + np.log(np.maximum(f, 1e-300) / _MIN_LOG_HZ) / _LOGSTEP, m)
    return m


def mel_to_hz(m, htk=False):
    m = np.asanyarray(m, dtype=np.float64)
    if htk:
        return 700.0 * (10.0 ** (m / 2595.0) - 1.0)
    f = _F_SP * m
    big = m >= _MIN_LOG_MEL
    return np.where(big, _MIN_LOG_HZ * np.exp(_LOGSTEP * (m - _MIN_LOG_MEL)), f)


def mel_frequencies(n_mels=128, fmin=0.0, fmax=11025.0, htk=False):
    return mel_to_hz(np.linspace(hz_to_mel(fmin, htk), hz_to_mel(fmax, htk), n_mels), htk)


def mel_filterbank(sr, n_fft, n_mels=128, fmin=0.0, fmax=None, htk=False,
                   norm="slaney") -> np.ndarray:
    """float32 [n_mels, 1 + n_fft//2] exactly as librosa.filters.mel stores it.

    The triangles are evaluated in float64, *stored* float32, then the Slaney
    area normalisation multiplies the float32 values (product formed in
    float64, rounded to float32 again) -- both roundings are reproduced.
    """
    if fmax is None:
        fmax = sr / 2.0
    F = 1 + n_fft // 2
    freqs = fft_frequencies(sr, n_fft)
    edges = mel_frequencies(n_mels + 2, fmin, fmax, htk)
    width = np.diff(edges)
    W = np.zeros((n_mels, F), dtype=np.float32)
    for m in range(n_mels):
        rising = (freqs - edges[m]) / width[m]
        falling = (edges[m + 2] - freqs) / width[m + 1]
        W[m] = np.maximum(0.0, np.minimum(rising, falling)).astype(np.float32)
    if norm == "slaney":
        enorm = 2.0 / (edges[2:n_mels + 2] - edges[:n_mels])
        W = (W.astype(np.float64) * enorm[:, None]).astype(np.float32)
    elif norm is not None:
        raise ValueError("norm must be 'slaney' or None in this restatement")
    return W


def melspectrogram(S_power, sr, n_fft=None, n_mels=128, fmin=0.0, fmax=None) -> np.ndarray:
    """[n_mels, T] float64 = basis(f32->f64) @ S_power  (einsum 'ft,mf->mt')."""
    S_power = np.asarray(S_power, dtype=np.float64)
    if n_fft is None or n_fft // 2 + 1 != S_power.shape[0]:
        n_fft = 2 * (S_power.shape[0] - 1)
    B = mel_filterbank(sr, n_fft, n_mels, fmin, fmax)
    return B.astype(np.float64) @ S_power


# --------------------------------------------------------------------------
# a4  power_to_db (manager.py:223)
# --------------------------------------------------------------------------
def power_to_db(S, ref=1.0, amin=1e-10, top_db=80.0) -> np.ndarray:
    S = np.asarray(S, dtype=np.float64)
    if amin <= 0:
        raise ValueError("amin must be strictly positive")
    ref_value = ref(S) if callable(ref) else np.abs(ref)
    out = 10.0 * np.log10(np.maximum(amin, S))
    out = out - 10.0 * np.log10(np.maximum(amin, ref_value))
    if top_db is not None:
        if top_db < 0:
            raise ValueError("top_db must be non-negative")
        out = np.maximum(out, out.max() - top_db)
    return out


# --------------------------------------------------------------------------
# a5  MFCC (cepstral.py:20-120 -> librosa.feature.mfcc -> scipy.fft.dct)
# --------------------------------------------------------------------------
def mfcc(y=None, sr=None, S=None, n_mfcc=13, dct_type=2, norm="ortho", lifter=0.0,
         **kwargs) -> np.ndarray:
    if S is None and y is None:
        raise ValueError("Either audio time series 'y' or Mel spectrogram 'S' must be provided.")  # cepstral.py:94-95
    if S is None and sr is None:
        raise ValueError("Sampling rate 'sr' must be provided when calculating MFCCs from time series 'y'.")  # :96-97
    if S is None:
        # librosa.feature.mfcc(y=..): power_to_db(melspectrogram(y, sr, **kw)), ref=1.0
        n_fft = kwargs.get("n_fft", 2048)
        hop = kwargs.get("hop_length", 512)
        X = stft(y, n_fft=n_fft, hop_length=hop, win_length=kwargs.get("win_length"),
                 window=kwargs.get("window", "hann"), center=kwargs.get("center", True))
        P = np.abs(X) ** kwargs.get("power", 2.0)
        M = melspectrogram(P, sr, n_fft, kwargs.get("n_mels", 128), kwargs.get("fmin", 0.0),
                           kwargs.get("fmax"))
        S = power_to_db(M)
    S = np.asarray(S, dtype=np.float64)
    C = scipy.fft.dct(S, axis=-2, type=dct_type, norm=norm)[..., :n_mfcc, :]
    if lifter > 0:
        li = np.sin(np.pi * np.arange(1, 1 + n_mfcc, dtype=np.float64) / lifter)
        C = C * (1 + (lifter / 2) * li[:, None])
    elif lifter < 0:
        raise ValueError(f"MFCC lifter={lifter} must be a non-negative number")
    return C.astype(np.float64)


def log_mel_manager(y, sr, n_fft=2048, hop_length=512, center=True, window="hann",
                    n_mels=128, fmin=0.0, fmax=None, power=2.0):
    """The manager's cached chain: manager.py:177-227 (ref=np.max, top_db 80)."""
    S_mag = np.abs(stft(y, n_fft, hop_length, n_fft, window, center))
    M = melspectrogram(S_mag ** power, sr, n_fft, n_mels, fmin, sr / 2.0 if fmax is None else fmax)
    return power_to_db(M, ref=np.max), S_mag


def mfcc_manager(y, sr, n_fft=2048, hop_length=512, center=True, window="hann",
                 n_mels=128, n_mfcc=13, fmin=0.0, fmax=None, power=2.0, lifter=0.0):
    """[n_mfcc, T] -- the graded path a1->a2->a3->a4->a5 for one clip."""
    L, _ = log_mel_manager(y, sr, n_fft, hop_length, center, window, n_mels, fmin, fmax, power)
    return mfcc(S=L, sr=sr, n_mfcc=n_mfcc, lifter=lifter)


# --------------------------------------------------------------------------
# a6/a7/a9 per-frame spectral features (frequency_domain.py:24-386)
# --------------------------------------------------------------------------
def spectral_centroid(mag, freqs) -> np.float64:
    mag = np.asarray(mag, dtype=np.float64); freqs = np.asarray(freqs, dtype=np.float64)
    if mag.shape != freqs.shape:
        raise ValueError(f"Spectrum shape {mag.shape} and frequencies shape {freqs.shape} must match.")
    if mag.size == 0:
        return np.float64(0.0)
    mag = np.abs(mag)
    s = np.sum(mag)
    if s < EPS64:
        return np.float64(0.0)
    return np.float64(np.sum(freqs * mag) / s)


def spectral_bandwidth(mag, freqs, centroid=None, p=2) -> np.float64:
    mag = np.asarray(mag, dtype=np.float64); freqs = np.asarray(freqs, dtype=np.float64)
    if mag.shape != freqs.shape:
        raise ValueError(f"Spectrum shape {mag.shape} and frequencies shape {freqs.shape} must match.")
    if p <= 0:
        raise ValueError("Order 'p' for spectral bandwidth must be positive.")
    if mag.size == 0:
        return np.float64(0.0)
    mag = np.abs(mag)
    s = np.sum(mag)
    if s < EPS64:
        return np.float64(0.0)
    if centroid is None:
        centroid = spectral_centroid(mag, freqs)
    dev = np.sum(mag * np.abs(freqs - centroid) ** p)
    dev = max(dev, 0.0)
    return np.float64((dev / s) ** (1.0 / p))


def spectral_flatness(mag) -> np.float64:
    mag = np.asarray(mag, dtype=np.float64)
    if mag.size == 0:
        return np.float64(0.0)
    mag = np.abs(mag)
    gm = np.exp(np.mean(np.log(mag + EPS64)))
    am = np.mean(mag)
    if am < EPS64:
        return np.float64(0.0)
    return np.float64(np.clip(gm / am, 0.0, 1.0))


def spectral_rolloff(mag, freqs, roll_percent=0.85) -> np.float64:
    mag = np.asarray(mag, dtype=np.float64); freqs = np.asarray(freqs, dtype=np.float64)
    if mag.shape != freqs.shape:
        raise ValueError(f"Spectrum shape {mag.shape} and frequencies shape {freqs.shape} must match.")
    if not 0.0 <= roll_percent <= 1.0:
        raise ValueError("roll_percent must be between 0.0 and 1.0.")
    if mag.size == 0:
        return np.float64(0.0)
    pw = np.abs(mag) ** 2                       # on POWER, frequency_domain.py:325
    tot = np.sum(pw)
    if tot < EPS64:
        return np.float64(freqs[-1])
    hit = np.nonzero(np.cumsum(pw) >= roll_percent * tot)[0]
    if hit.size == 0:
        return np.float64(freqs[-1])
    return np.float64(freqs[hit[0]])


def dominant_frequency(mag, freqs) -> np.float64:
    mag = np.asarray(mag, dtype=np.float64); freqs = np.asarray(freqs, dtype=np.float64)
    if mag.shape != freqs.shape:
        raise ValueError(f"Spectrum shape {mag.shape} and frequencies shape {freqs.shape} must match.")
    if mag.size == 0:
        return np.float64(0.0)
    return np.float64(freqs[int(np.argmax(mag))])


def spectral_stats_frames(S_mag, freqs, roll_percent=0.85, p=2):
    """Vectorised a6/a7/a9 over all columns of S_mag [F, T] (same formulas).

    Returns dict of [T] arrays plus 'rolloff_bin', 'rolloff_margin',
    'dominant_bin', 'dominant_margin' (float64 decision margins used by the
    parity gate for bin-valued outputs).
    """
    S = np.abs(np.asarray(S_mag, dtype=np.float64))
    f = np.asarray(freqs, dtype=np.float64)[:, None]
    F, T = S.shape
    s = S.sum(axis=0)
    ok = s >= EPS64
    ssafe = np.where(ok, s, 1.0)
    cen = np.where(ok, (f * S).sum(axis=0) / ssafe, 0.0)
    bw = np.where(ok, ((S * np.abs(f - cen[None, :]) ** p).sum(axis=0) / ssafe) ** (1.0 / p), 0.0)
    am = S.mean(axis=0)
    gm = np.exp(np.log(S + EPS64).mean(axis=0))
    flat = np.where(am >= EPS64, np.clip(gm / np.where(am >= EPS64, am, 1.0), 0.0, 1.0), 0.0)
    pw = S ** 2
    tot = pw.sum(axis=0)
    cs = np.cumsum(pw, axis=0)
    thr = roll_percent * tot
    rb = np.argmax(cs >= thr[None, :], axis=0)
    rb = np.where(tot < EPS64, F - 1, rb)
    # decision margin: distance of the threshold from the two neighbouring cumsums
    lo = np.where(rb > 0, cs[np.maximum(rb - 1, 0), np.arange(T)], -np.inf)
    hi = cs[rb, np.arange(T)]
    with np.errstate(invalid="ignore", divide="ignore"):
        rmargin = np.minimum(hi - thr, thr - lo) / np.where(tot > 0, tot, 1.0)
    db = np.argmax(S, axis=0)
    top2 = np.sort(S, axis=0)[-2:, :] if F >= 2 else np.vstack([S, S])
    with np.errstate(invalid="ignore", divide="ignore"):
        dmargin = (top2[1] - top2[0]) / np.where(top2[1] > 0, top2[1], 1.0)
    fr = np.asarray(freqs, dtype=np.float64)
    return {
        "spectral_centroid": cen, "spectral_bandwidth": bw, "spectral_flatness": flat,
        "spectral_rolloff": fr[rb], "dominant_frequency": fr[db],
        "rolloff_bin": rb, "rolloff_margin": rmargin, "dominant_bin": db, "dominant_margin": dmargin,
    }


# --------------------------------------------------------------------------
# a8 spectral contrast (frequency_domain.py:147-212 -> librosa.feature.spectral_contrast)
# --------------------------------------------------------------------------
def contrast_bands(freqs, sr, n_bands=6, fmin=200.0, quantile=0.02):
    """Per-band (bin index array, k) following librosa's band rules."""
    freqs = np.atleast_1d(np.asarray(freqs, dtype=np.float64))
    if n_bands < 1 or int(n_bands) != n_bands:
        raise ValueError("n_bands must be a positive integer")
    if not 0.0 < quantile < 1.0:
        raise ValueError("quantile must lie in the range (0, 1)")
    if fmin <= 0:
        raise ValueError("fmin must be a positive number")
    octa = np.zeros(n_bands + 2)
    octa[1:] = fmin * (2.0 ** np.arange(0, n_bands + 1))
    if np.any(octa[:-1] >= 0.5 * sr):
        raise ValueError("Frequency band exceeds Nyquist. Reduce either fmin or n_bands.")
    out = []
    for k in range(n_bands + 1):
        sel = np.logical_and(freqs >= octa[k], freqs <= octa[k + 1])
        idx = np.flatnonzero(sel)
        if k > 0:
            sel[idx[0] - 1] = True
        if k == n_bands:
            sel[idx[-1] + 1:] = True
        cnt = int(np.sum(sel))
        bins = np.flatnonzero(sel)
        if k < n_bands:
            bins = bins[:-1]
        kk = int(max(np.rint(quantile * cnt), 1))
        out.append((bins, kk))
    return out


def spectral_contrast(S, sr, n_bands=6, fmin=200.0, freqs=None, quantile=0.02, linear=False):
    S = np.asarray(S, dtype=np.float64)
    if S.ndim != 2:
        raise ValueError("Input S must be a 2D spectrogram (frequency x time).")  # frequency_domain.py:191-192
    S = np.abs(S)
    if freqs is None:
        freqs = fft_frequencies(sr, 2 * (S.shape[0] - 1))
    freqs = np.atleast_1d(freqs)
    if freqs.ndim != 1 or len(freqs) != S.shape[0]:
        raise ValueError(f"freq.shape={freqs.shape} does not match dimensions of S.shape={S.shape}")
    bands = contrast_bands(freqs, sr, n_bands, fmin, quantile)
    T = S.shape[1]
    valley = np.zeros((n_bands + 1, T)); peak = np.zeros((n_bands + 1, T))
    for k, (bins, kk) in enumerate(bands):
        srt = np.sort(S[bins, :], axis=0)
        valley[k] = np.mean(srt[:kk], axis=0)
        peak[k] = np.mean(srt[-kk:], axis=0)
    if linear:
        return peak - valley
    return power_to_db(peak) - power_to_db(valley)


# --------------------------------------------------------------------------
# a10/a11 fft / ifft / window (dsp.py:40-162, 641-691)
# --------------------------------------------------------------------------
def apply_window(data, window_type="hann"):
    data = np.asarray(data, dtype=np.float64)
    if data.ndim != 1:
        raise ValueError("Input data must be a 1D array.")
    try:
        w = scipy.signal.get_window(window_type, data.shape[0], fftbins=False)  # symmetric, dsp.py:676
    except ValueError as e:
        raise ValueError(f"Invalid window type '{window_type}'.") from e
    return data * w


def compute_fft(data, fs=1.0, n=None, window="hann"):
    data = np.asarray(data, dtype=np.float64)
    if data.ndim != 1:
        raise ValueError("Input data must be a 1D array.")
    x = apply_window(data, window) if window else data
    if n is None:
        n = x.shape[0]
    return scipy.fft.fftfreq(n, d=1 / fs).astype(np.float64), scipy.fft.fft(x, n=n).astype(np.complex128)


def compute_ifft(spectrum, n=None):
    spectrum = np.asarray(spectrum)
    if spectrum.ndim != 1:
        raise ValueError("Input spectrum must be a 1D array.")
    if n is None:
        n = spectrum.shape[0]
    return np.real(scipy.fft.ifft(spectrum, n=n)).astype(np.float64)


# --------------------------------------------------------------------------
# a12/a13 Butterworth SOS + zero-phase filtering (filters.py:22-211)
# --------------------------------------------------------------------------
def design_butterworth_sos(cutoff, fs, order, filter_type):
    nyq = 0.5 * fs
    if isinstance(cutoff, (int, float)):
        if not 0 < cutoff < nyq:
            raise ValueError(f"Cutoff frequency ({cutoff} Hz) must be strictly between 0 and Nyquist ({nyq} Hz).")
        wn = cutoff / nyq
    elif isinstance(cutoff, tuple) and len(cutoff) == 2:
        lo, hi = cutoff
        if not (0 < lo < nyq and 0 < hi < nyq):
            raise ValueError(f"Both low ({lo} Hz) and high ({hi} Hz) cutoff frequencies must be strictly between 0 and Nyquist ({nyq} Hz).")
        if lo >= hi:
            raise ValueError(f"Low cutoff ({lo} Hz) must be less than high cutoff ({hi} Hz).")
        wn = (lo / nyq, hi / nyq)
    else:
        raise TypeError("cutoff must be a float (for low/high pass) or a tuple of two floats (for band pass/stop).")
    return scipy.signal.butter(order, wn, btype=filter_type, analog=False, output="sos").astype(np.float64)


def sosfiltfilt_padlen(sos) -> int:
    """scipy.signal.sosfiltfilt default edge length (padtype='odd', padlen=None)."""
    sos = np.asarray(sos, dtype=np.float64)
    ntaps = 2 * sos.shape[0] + 1
    ntaps -= min(int((sos[:, 2] == 0).sum()), int((sos[:, 5] == 0).sum()))
    return 3 * ntaps


def sosfilt_zi(sos) -> np.ndarray:
    """Steady-state DF2T initial state per section for a unit step (scipy.signal.sosfilt_zi)."""
    sos = np.asarray(sos, dtype=np.float64)
    zi = np.zeros((sos.shape[0], 2))
    scale = 1.0
    for s in range(sos.shape[0]):
        b = sos[s, :3] / sos[s, 3]
        a = sos[s, 3:] / sos[s, 3]
        # solve (I - A^T) z = b[1:] - a[1:] b0 for the DF2T companion matrix
        IminusA = np.array([[1.0 + a[1], -1.0], [a[2], 1.0]])
        rhs = np.array([b[1] - a[1] * b[0], b[2] - a[2] * b[0]])
        zi[s] = scale * np.linalg.solve(IminusA, rhs)
        scale *= b.sum() / a.sum()
    return zi


def sosfilt_explicit(sos, x, zi):
    """Direct-form-II-transposed cascade, pure Python/NumPy scalar loop (small inputs only)."""
    sos = np.asarray(sos, dtype=np.float64)
    y = np.array(x, dtype=np.float64)
    z = np.array(zi, dtype=np.float64)
    for s in range(sos.shape[0]):
        b0, b1, b2, a0, a1, a2 = sos[s]
        b0, b1, b2, a1, a2 = b0 / a0, b1 / a0, b2 / a0, a1 / a0, a2 / a0
        z0, z1 = z[s]
        for n in range(y.shape[0]):
            xn = y[n]
            yn = b0 * xn + z0
            z0 = b1 * xn - a1 * yn + z1
            z1 = b2 * xn - a2 * yn
            y[n] = yn
        z[s] = (z0, z1)
    return y, z


def sosfiltfilt_explicit(sos, x):
    """Restatement of scipy.signal.sosfiltfilt(sos, x) (odd extension, zi-scaled, fwd+bwd)."""
    x = np.asarray(x, dtype=np.float64)
    edge = sosfiltfilt_padlen(sos)
    if x.shape[0] <= edge:
        raise ValueError(f"The length of the input vector x must be greater than padlen, which is {edge}.")
    ext = np.concatenate([2 * x[0] - x[edge:0:-1], x, 2 * x[-1] - x[-2:-(edge + 2):-1]])
    zi = sosfilt_zi(sos)
    y, _ = sosfilt_explicit(sos, ext, zi * ext[0])
    y, _ = sosfilt_explicit(sos, y[::-1], zi * y[-1])
    return y[::-1][edge:-edge]


def apply_sos_filter(sos, data):
    data = np.asarray(data, dtype=np.float64); sos = np.asarray(sos, dtype=np.float64)
    if data.ndim != 1:
        raise ValueError("Input data for filtering must be a 1D array.")
    if sos.ndim != 2 or sos.shape[1] != 6:
        raise ValueError("Input sos must be a 2D array with shape (n_sections, 6).")
    return scipy.signal.sosfiltfilt(sos, data).astype(np.float64)


def low_pass_filter(data, cutoff, fs, order=5):
    return apply_sos_filter(design_butterworth_sos(cutoff, fs, order, "lowpass"), data)


def high_pass_filter(data, cutoff, fs, order=5):
    return apply_sos_filter(design_butterworth_sos(cutoff, fs, order, "highpass"), data)


def band_pass_filter(data, low_cutoff, high_cutoff, fs, order=5):
    return apply_sos_filter(design_butterworth_sos((low_cutoff, high_cutoff), fs, order, "bandpass"), data)


def band_stop_filter(data, low_cutoff, high_cutoff, fs, order=5):
    return apply_sos_filter(design_butterworth_sos((low_cutoff, high_cutoff), fs, order, "bandstop"), data)


# --------------------------------------------------------------------------
# a14 Welch (dsp.py:495-560 -> scipy.signal.welch)
# --------------------------------------------------------------------------
def compute_psd_welch(x, fs=1.0, window="hann", nperseg=None, noverlap=None, nfft=None,
                      detrend="constant", scaling="density"):
    x = np.asarray(x, dtype=np.float64)
    if x.ndim != 1:
        raise ValueError("Input data must be a 1D array.")
    f, p = scipy.signal.welch(x, fs=fs, window=window, nperseg=nperseg, noverlap=noverlap, nfft=nfft,
                              detrend=detrend, return_onesided=True, scaling=scaling)
    return f.astype(np.float64), p.astype(np.float64)


def welch_explicit(x, fs=1.0, window="hann", nperseg=256, noverlap=None, nfft=None,
                   detrend="constant", scaling="density"):
    """Explicit restatement of scipy.signal.welch for 1-D real x (validates the SciPy call)."""
    x = np.asarray(x, dtype=np.float64)
    if nperseg > x.shape[0]:
        nperseg = x.shape[0]
    if noverlap is None:
        noverlap = nperseg // 2
    if nfft is None:
        nfft = nperseg
    step = nperseg - noverlap
    w = scipy.signal.get_window(window, nperseg)  # periodic (fftbins=True default)
    nseg = (x.shape[0] - noverlap) // step
    idx = np.arange(nperseg)[None, :] + step * np.arange(nseg)[:, None]
    seg = x[idx]
    if detrend == "constant":
        seg = seg - seg.mean(axis=1, keepdims=True)
    elif detrend == "linear":
        seg = scipy.signal.detrend(seg, axis=1, type="linear")
    X = np.fft.rfft(seg * w[None, :], n=nfft, axis=1)
    scale = 1.0 / (fs * (w * w).sum()) if scaling == "density" else 1.0 / w.sum() ** 2
    P = (np.abs(X) ** 2) * scale
    if nfft % 2:
        P[:, 1:] *= 2
    else:
        P[:, 1:-1] *= 2
    return np.fft.rfftfreq(nfft, 1 / fs), P.mean(axis=0)


# --------------------------------------------------------------------------
# f-3  FFT-backed 1-D operations: convolution / correlation (dsp.py:294-433 -> scipy.signal.fftconvolve,
#      scipy.signal.correlate), periodogram (dsp.py:438-498 -> scipy.signal.periodogram), analytic signal
#      (transforms.py:119-151, dsp.py:565-636 -> scipy.signal.hilbert).  Restated from the definitions in float64
#      and pinned on tests/golden/ref_dsp2.npz (outputs of the reference functions).
# --------------------------------------------------------------------------
def conv_slice(n, m, mode):
    """Start and length of scipy's `mode` result inside the full linear convolution of n and m samples
    ('same' is centred with respect to the FIRST input, scipy.signal._signaltools._centered)."""
    if mode == "full":
        return 0, n + m - 1
    if mode == "same":
        return (m - 1) // 2, n
    if mode == "valid":
        return min(n, m) - 1, max(n, m) - min(n, m) + 1
    raise ValueError("acceptable mode flags are 'valid', 'same', or 'full'")


def apply_convolution(data, kernel, mode="same"):             # dsp.py:294-337
    data = np.asarray(data, dtype=np.float64)
    kernel = np.asarray(kernel, dtype=np.float64)
    if data.ndim != 1 or kernel.ndim != 1:
        raise ValueError("Input data and kernel must be 1D arrays.")
    if data.size == 0 or kernel.size == 0:
        return np.array([], dtype=np.float64)
    full = np.convolve(data, kernel, mode="full")              # direct sum: the definition fftconvolve evaluates
    start, count = conv_slice(data.shape[0], kernel.shape[0], mode)
    return full[start:start + count]


def compute_correlation(x, y, mode="full", method="auto"):    # dsp.py:342-394
    """scipy.signal.correlate(x, y) = convolve(x, y[::-1]) for real input, whatever `method`."""
    x = np.asarray(x, dtype=np.float64)
    y = np.asarray(y, dtype=np.float64)
    if x.ndim != 1 or y.ndim != 1:
        raise ValueError("Input sequences for correlation must be 1D arrays.")
    return apply_convolution(x, y[::-1], mode)


def compute_autocorrelation(x, mode="full", method="auto"):   # dsp.py:396-433
    return compute_correlation(x, x, mode, method)


def compute_psd_periodogram(x, fs=1.0, window="hann", nfft=None, detrend="constant", scaling="density"):
    """dsp.py:438-498: scipy.signal.periodogram = one Welch segment spanning the (possibly truncated) signal."""
    x = np.asarray(x, dtype=np.float64)
    if x.ndim != 1:
        raise ValueError("Input data must be a 1D array.")
    if x.size == 0:
        return np.empty(0), np.empty(0)
    if nfft is None:
        nfft = x.shape[0]
    if nfft < x.shape[0]:
        x = x[:nfft]
    nperseg = x.shape[0]
    w = scipy.signal.get_window(window, nperseg) if isinstance(window, (str, tuple)) else np.asarray(window, float)
    seg = x
    if detrend == "constant":
        seg = seg - seg.mean()
    elif detrend == "linear":
        seg = scipy.signal.detrend(seg, type="linear")
    X = np.fft.rfft(seg * w, n=nfft)
    scale = 1.0 / (fs * (w * w).sum()) if scaling == "density" else 1.0 / w.sum() ** 2
    P = (np.abs(X) ** 2) * scale
    if nfft % 2:
        P[1:] *= 2
    else:
        P[1:-1] *= 2
    return np.fft.rfftfreq(nfft, 1 / fs), P


def hilbert_transform(data):                                  # transforms.py:119-151
    data = np.asarray(data, dtype=np.float64)
    if data.ndim != 1:
        raise ValueError("Input data must be a 1D array.")
    n = data.shape[0]
    if n == 0:
        raise ValueError("N must be positive.")
    h = np.zeros(n)
    if n % 2 == 0:
        h[0] = h[n // 2] = 1.0
        h[1:n // 2] = 2.0
    else:
        h[0] = 1.0
        h[1:(n + 1) // 2] = 2.0
    return np.fft.ifft(np.fft.fft(data) * h)


def amplitude_envelope(y, method="hilbert", frame_length=None, hop_length=None):   # dsp.py:565-636
    y = np.asarray(y, dtype=np.float64)
    if y.ndim != 1:
        raise ValueError("Input data must be a 1D array.")
    if method == "hilbert":
        return np.abs(hilbert_transform(y))
    if method == "rms":
        if frame_length is None or hop_length is None:
            raise ValueError("frame_length and hop_length are required for 'rms' envelope method.")
        return rms_energy(y, frame_length=frame_length, hop_length=hop_length, center=True)
    raise ValueError(f"Unsupported envelope method: {method}. Choose 'hilbert' or 'rms'.")


# --------------------------------------------------------------------------
# f-1  time-domain frame features (core/features/time_domain.py:23-227, driven per frame by
#      manager.py:264-286) and RMS / zero-crossing rate (core/audio/features.py:26-131 -> librosa)
# --------------------------------------------------------------------------
TIME_FEATURES = ("mean_amplitude", "std_dev_amplitude", "skewness", "kurtosis", "peak_amplitude",
                 "crest_factor", "signal_entropy")


def mean_amplitude(frame) -> np.float64:                      # time_domain.py:23-44
    frame = np.asarray(frame, dtype=np.float64)
    return np.float64(0.0) if frame.size == 0 else np.mean(np.abs(frame))


def std_dev_amplitude(frame) -> np.float64:                   # time_domain.py:46-67 (population std)
    frame = np.asarray(frame, dtype=np.float64)
    return np.float64(0.0) if frame.size == 0 else np.std(frame)


def _central_moments(frame):
    m = frame.mean()
    d = frame - m
    return (d ** 2).mean(), (d ** 3).mean(), (d ** 4).mean()


def skewness(frame) -> np.float64:
    """time_domain.py:69-97: scipy.stats.skew(bias=False) = sqrt(n(n-1))/(n-2) * m3/m2^1.5; 0 for n < 2 or
    var < eps.  (scipy returns nan for n < 3 with bias=False: n = 2 gives 0/0 -> the reference propagates it;
    restated here as the same formula.)"""
    frame = np.asarray(frame, dtype=np.float64)
    n = frame.size
    if n < 2 or np.var(frame) < EPS64:
        return np.float64(0.0)
    m2, m3, _ = _central_moments(frame)
    g1 = m3 / m2 ** 1.5
    if n < 3:
        return np.float64(g1)          # scipy applies the bias correction only when n > 2
    return np.float64(np.sqrt(n * (n - 1.0)) / (n - 2.0) * g1)


def kurtosis_val(frame) -> np.float64:
    """time_domain.py:99-130: scipy.stats.kurtosis(fisher=True, bias=False) =
    (n-1)/((n-2)(n-3)) * ((n+1) m4/m2^2 - 3(n-1)); 0 for n < 4 or var < eps."""
    frame = np.asarray(frame, dtype=np.float64)
    n = frame.size
    if n < 4 or np.var(frame) < EPS64:
        return np.float64(0.0)
    m2, _, m4 = _central_moments(frame)
    return np.float64((n - 1.0) / ((n - 2.0) * (n - 3.0)) * ((n + 1.0) * m4 / m2 ** 2 - 3.0 * (n - 1.0)))


def peak_amplitude(frame) -> np.float64:                      # time_domain.py:132-153
    frame = np.asarray(frame, dtype=np.float64)
    return np.float64(0.0) if frame.size == 0 else np.max(np.abs(frame))


def crest_factor(frame) -> np.float64:                        # time_domain.py:155-186
    frame = np.asarray(frame, dtype=np.float64)
    if frame.size == 0:
        return np.float64(0.0)
    rms = np.sqrt(np.mean(frame ** 2))
    return np.float64(0.0) if rms < EPS64 else np.float64(peak_amplitude(frame) / rms)


def histogram_counts(frame, num_bins):
    """np.histogram(frame, bins=num_bins) as NumPy computes it for uniform bins: index from the scaled
    offset, then corrected against the linspace edges; the last bin is closed on the right."""
    first, last = frame.min(), frame.max()
    if first == last:
        first, last = first - 0.5, last + 0.5
    edges = np.linspace(first, last, num_bins + 1)
    norm = num_bins / (last - first)
    idx = ((frame - first) * norm).astype(np.intp)
    idx[idx == num_bins] -= 1
    idx[frame < edges[idx]] -= 1
    inc = (frame >= edges[idx + 1]) & (idx != num_bins - 1)
    idx[inc] += 1
    return np.bincount(idx, minlength=num_bins)


def signal_entropy(frame, num_bins: int = 10) -> np.float64:  # time_domain.py:188-227
    frame = np.asarray(frame, dtype=np.float64)
    if frame.size < 2 or num_bins < 1 or np.all(frame == frame[0]):
        return np.float64(0.0)
    counts = histogram_counts(frame, num_bins)
    pk = counts[counts > 0] / frame.size
    pk = pk / pk.sum()                                        # scipy.stats.entropy normalises
    return np.float64(-(pk * np.log(pk)).sum())


_TIME_FUNCS = {"mean_amplitude": mean_amplitude, "std_dev_amplitude": std_dev_amplitude, "skewness": skewness,
               "kurtosis": kurtosis_val, "peak_amplitude": peak_amplitude, "crest_factor": crest_factor,
               "signal_entropy": signal_entropy}


def time_features_frames(y, frame_length=2048, hop_length=512, center=True, num_bins=10):
    """manager.py:264-286: zero-pad frame_length//2 when centred, frame, apply each function per frame."""
    y = np.asarray(y, dtype=np.float64)
    fr = frame_signal(y, frame_length, hop_length, center)   # [T, frame_length]
    out = {}
    for name, fn in _TIME_FUNCS.items():
        if name == "signal_entropy":
            out[name] = np.array([fn(f, num_bins) for f in fr], dtype=np.float64)
        else:
            out[name] = np.array([fn(f) for f in fr], dtype=np.float64)
    return out


def rms_energy(y=None, S=None, frame_length=2048, hop_length=512, center=True):
    """audio/features.py:73-131 -> librosa.feature.rms (UNPINNED, restated): from y: zero-pad frame_length//2,
    frame, sqrt(mean(x^2)); from a magnitude spectrogram S: sqrt(2 sum(|S|^2 with DC (and Nyquist for even
    frame_length) halved) / frame_length^2)."""
    if S is not None:
        x = np.abs(np.asarray(S, dtype=np.float64)) ** 2
        x[0] *= 0.5
        if frame_length % 2 == 0:
            x[-1] *= 0.5
        return np.sqrt(2.0 * x.sum(axis=0) / frame_length ** 2)
    y = np.asarray(y, dtype=np.float64)
    if y.ndim != 1:
        raise ValueError("Input audio data 'y' must be a 1D array.")
    fr = frame_signal(y, frame_length, hop_length, center)   # [T, frame_length]
    return np.sqrt(np.mean(fr ** 2, axis=1))


def zero_crossing_rate(y, frame_length=2048, hop_length=512, center=True, threshold=1e-10):
    """audio/features.py:26-71 -> librosa.feature.zero_crossing_rate (UNPINNED, restated): EDGE-pad
    frame_length//2, frame, zero the samples with |x| <= threshold, count sign-bit changes between neighbours
    inside the frame (the first sample of a frame never counts: pad=False), divide by frame_length."""
    y = np.asarray(y, dtype=np.float64)
    if y.ndim != 1:
        raise ValueError("Input audio data must be a 1D array.")
    if center:
        y = np.pad(y, frame_length // 2, mode="edge")
    T = 1 + (len(y) - frame_length) // hop_length if len(y) >= frame_length else 0
    out = np.zeros(max(T, 0), dtype=np.float64)
    for t in range(T):
        f = y[t * hop_length: t * hop_length + frame_length].copy()
        f[np.abs(f) <= threshold] = 0.0
        sb = np.signbit(f)
        out[t] = np.count_nonzero(sb[1:] != sb[:-1]) / frame_length
    return out


# --------------------------------------------------------------------------
# a16 extract_features orchestration (manager.py:78-445), dict_of_arrays form
# --------------------------------------------------------------------------
SPECTRUM_FEATURES = ("spectral_centroid", "spectral_bandwidth", "spectral_flatness",
                     "spectral_rolloff", "dominant_frequency")
KNOWN_FEATURES = set(SPECTRUM_FEATURES) | {"spectral_contrast", "mfcc"} | set(TIME_FEATURES) | {
    "rms_energy", "zero_crossing_rate"}


def _nan_pad(a, T):
    """manager.py:378-386: a feature array shorter than the frame count is padded with NaN, a longer one cut."""
    a = np.asarray(a, dtype=np.float64)
    if len(a) >= T:
        return a[:T]
    out = np.full(T, np.nan, dtype=np.float64)
    out[: len(a)] = a
    return out


def extract_features(y, sr, features, frame_length=2048, hop_length=512, center=True,
                     window="hann", feature_params=None, per_frame_loop=False):
    """Oracle for manager.py:78-445 (dict_of_arrays), features processed IN THE ORDER GIVEN as the reference does:

    * `time` starts from the frame-count rule (:149-169); the first feature that needs the STFT computes it, and
      when the STFT has another frame count (odd frame_length with hop | len(y)) `time` is re-made from the STFT
      (:186-194);
    * every feature array is NaN-padded / cut to the frame count current WHEN IT IS PROCESSED (:376-387);
    * `spectral_bandwidth` without an earlier `spectral_centroid` stores the centroid it depends on as an output
      column of its own (:296-301);
    * rows whose length differs from the final `time` are dropped (:408-420).

    ``per_frame_loop=True`` drives a6/a7/a9 one frame at a time exactly as manager.py:304-316 does (used for the
    'reference-equivalent' CPU timing).
    """
    feature_params = feature_params or {}
    y = np.asarray(y, dtype=np.float64)
    unknown = [f for f in features if f not in KNOWN_FEATURES]
    if unknown:
        raise ValueError(f"Unknown feature(s) requested: {unknown}.")
    if y.ndim != 1:
        raise ValueError("Input audio signal 'y' must be a 1D array.")
    T = num_frames(len(y), frame_length, hop_length, center)
    if T <= 0:
        return {"time": np.array([], dtype=np.float64)}
    res = {"time": frames_to_time(np.arange(T), sr, hop_length, frame_length if center else None).astype(np.float64)}
    freqs = fft_frequencies(sr, frame_length)
    box = {}

    def s_mag():
        if "S" not in box:
            box["S"] = np.abs(stft(y, frame_length, hop_length, frame_length, window, center))
            if box["S"].shape[1] != len(res["time"]):                     # manager.py:186-194
                res["time"] = frames_to_time(np.arange(box["S"].shape[1]), sr, hop_length, frame_length).astype(np.float64)
        return box["S"]

    stats = None
    tstats = None
    done = set()
    for name in features:
        if name in done:
            continue
        p = feature_params.get(name, {})
        cur = len(res["time"])
        if name in TIME_FEATURES:
            if tstats is None:
                tstats = time_features_frames(y, frame_length, hop_length, center,
                                              feature_params.get("signal_entropy", {}).get("num_bins", 10))
            res[name] = _nan_pad(tstats[name], cur)
        elif name == "rms_energy":
            res[name] = _nan_pad(rms_energy(y, frame_length=frame_length, hop_length=hop_length, center=center), cur)
        elif name == "zero_crossing_rate":
            res[name] = _nan_pad(zero_crossing_rate(y, frame_length, hop_length, center), cur)
        elif name in SPECTRUM_FEATURES:
            S_mag = s_mag()
            if name == "spectral_bandwidth" and "spectral_centroid" not in res:     # manager.py:296-301
                res["spectral_centroid"] = np.array([spectral_centroid(S_mag[:, i], freqs) for i in range(S_mag.shape[1])],
                                                    dtype=np.float64)
                done.add("spectral_centroid")
            if per_frame_loop:
                fn = {"spectral_centroid": spectral_centroid, "spectral_bandwidth": spectral_bandwidth,
                      "spectral_flatness": spectral_flatness, "spectral_rolloff": spectral_rolloff,
                      "dominant_frequency": dominant_frequency}[name]
                if name == "spectral_flatness":
                    vals = [fn(S_mag[:, i]) for i in range(S_mag.shape[1])]
                else:
                    vals = [fn(S_mag[:, i], freqs, **p) for i in range(S_mag.shape[1])]
                res[name] = np.array(vals, dtype=np.float64)
            else:
                if stats is None:
                    stats = spectral_stats_frames(
                        S_mag, freqs,
                        roll_percent=feature_params.get("spectral_rolloff", {}).get("roll_percent", 0.85),
                        p=feature_params.get("spectral_bandwidth", {}).get("p", 2))
                res[name] = stats[name]
        elif name == "spectral_contrast":
            S_mag = s_mag()
            C = spectral_contrast(S_mag, sr, freqs=freqs, **p)
            for i in range(C.shape[0] - 1):
                res[f"contrast_band_{i}"] = C[i]
            res["contrast_delta"] = C[-1]
        elif name == "mfcc":
            mp = feature_params.get("mfcc", {})
            M = melspectrogram(s_mag() ** mp.get("power", 2.0), sr, frame_length, mp.get("n_mels", 128),
                               mp.get("fmin", 0.0), mp.get("fmax", sr / 2.0))
            Ldb = power_to_db(M, ref=np.max)
            C = mfcc(S=Ldb, sr=sr, n_mfcc=mp.get("n_mfcc", 13), dct_type=mp.get("dct_type", 2),
                     norm=mp.get("norm", "ortho"), lifter=mp.get("lifter", 0.0))
            for i in range(C.shape[0]):
                res[f"mfcc_{i}"] = C[i]
        done.add(name)
    final_T = len(res["time"])
    return {k: v for k, v in res.items() if k == "time" or len(v) == final_T}     # manager.py:408-420


# --------------------------------------------------------------------------
# f-4  feature formatting for ML: scalers (core/ml_utils/scaling.py:49-175 -> scikit-learn) and the sequence /
#      image formatters (core/ml_utils/formatters.py:166-334 -> NumPy, scipy.ndimage.zoom).  Where the reference
#      calls scikit-learn / SciPy the same routine is called here; pinned on tests/golden/ref_ml.npz.
# --------------------------------------------------------------------------
def apply_scaling(features, scaler_type="standard", scaler_params=None):
    """(scaled float64 array, dict of the fitted attributes) -- scaling.py:49-143 with fit=True."""
    from sklearn.preprocessing import MinMaxScaler, RobustScaler, StandardScaler
    X = np.asarray(features, dtype=np.float64)
    if X.ndim == 1:
        X = X.reshape(-1, 1)
    cls = {"standard": StandardScaler, "minmax": MinMaxScaler, "robust": RobustScaler}[scaler_type]
    sc = cls(**(scaler_params or {}))
    out = sc.fit_transform(X)
    attrs = {k: getattr(sc, k) for k in ("mean_", "var_", "scale_", "min_", "data_min_", "data_max_", "center_")
             if getattr(sc, k, None) is not None}
    return out.astype(np.float64), attrs


def format_feature_vectors_per_segment(features_dict, segment_indices, aggregation="mean"):      # formatters.py:51-163
    """[n_segments, n_features] float64; invalid segments stay NaN; NaN-aware aggregations (formatters.py:28-45)."""
    names = list(features_dict.keys())
    n = len(features_dict[names[0]])
    fn = {"mean": np.mean, "std": np.std, "median": np.median, "min": np.min, "max": np.max}
    how = {k: (aggregation if isinstance(aggregation, str) else aggregation.get(k, "mean")) for k in names}
    out = np.full((len(segment_indices), len(names)), np.nan)
    for i, (a, b) in enumerate(segment_indices):
        if not (0 <= a < n and a < b and b <= n):
            continue
        for j, k in enumerate(names):
            v = np.asarray(features_dict[k], dtype=np.float64)[a:b]
            v = v[~np.isnan(v)]
            if v.size:
                out[i, j] = fn[how[k]](v)
    return out


def format_feature_sequences(features_dict, max_sequence_length=None, padding_value=0.0, truncation_strategy="post",
                             output_format="list_of_arrays"):                    # formatters.py:166-253
    names = list(features_dict.keys())
    seq = np.stack([np.asarray(features_dict[k], dtype=np.float64) for k in names], axis=1)
    n = seq.shape[0]
    if max_sequence_length is not None and max_sequence_length > 0:
        if n > max_sequence_length:
            seq = seq[:max_sequence_length] if truncation_strategy == "post" else seq[n - max_sequence_length:]
        elif n < max_sequence_length:
            seq = np.pad(seq, ((0, max_sequence_length - n), (0, 0)), mode="constant", constant_values=padding_value)
    return [seq] if output_format == "list_of_arrays" else seq[None]


def format_features_as_image(feature_map, output_shape=None, resize_order=1, normalize=True):   # formatters.py:256-334
    import scipy.ndimage
    img = np.asarray(feature_map, dtype=np.float64)
    if output_shape is not None and img.shape != tuple(output_shape):
        img = scipy.ndimage.zoom(img, (output_shape[0] / img.shape[0], output_shape[1] / img.shape[1]),
                                 order=resize_order, mode="nearest")
        h, w = output_shape
        img = img[:h, :w]
        img = np.pad(img, ((0, h - img.shape[0]), (0, w - img.shape[1])), mode="constant")
    if normalize:
        lo, hi = np.min(img), np.max(img)
        img = np.zeros_like(img) if hi - lo < EPS64 else (img - lo) / (hi - lo)
    return img


# --------------------------------------------------------------------------
# batched helpers + the synthetic workload of SURVEY section 8d
# --------------------------------------------------------------------------
def synth_clips(n_clips, length=48000, sr=48000, seed=20250523, dtype=np.float32):
    """3 random sines + white noise per clip, peak <= 0.9 (SURVEY 8d recipe)."""
    rng = np.random.default_rng(seed)
    t = np.arange(length, dtype=np.float64) / sr
    out = np.empty((n_clips, length), dtype=dtype)
    for i in range(n_clips):
        f = rng.uniform(50.0, 20000.0 if sr >= 44100 else 0.45 * sr, 3)
        a = rng.uniform(0.05, 0.3, 3)
        ph = rng.uniform(0, 2 * np.pi, 3)
        y = (a[:, None] * np.sin(2 * np.pi * f[:, None] * t[None, :] + ph[:, None])).sum(axis=0)
        y += rng.normal(0.0, 0.05, length)
        y *= 0.9 / max(np.abs(y).max(), 1e-12) if np.abs(y).max() > 0.9 else 1.0
        out[i] = y.astype(dtype)
    return out


def mfcc_batch(Y, sr, n_fft=2048, hop_length=512, n_mels=40, n_mfcc=13, center=True, window="hann",
               fmin=0.0, fmax=None, lifter=0.0):
    """[B, n_mfcc, T] float64 -- config C2 for a batch of clips, one clip at a time."""
    return np.stack([mfcc_manager(np.asarray(y, dtype=np.float64), sr, n_fft, hop_length, center, window,
                                  n_mels, n_mfcc, fmin, fmax, 2.0, lifter) for y in Y])


# --------------------------------------------------------------------------
# a15 constant-Q transform (dsp.py:231-289 -> librosa.cqt)       PARITY UNPINNED
# --------------------------------------------------------------------------
# Restatement of librosa.cqt / librosa.vqt (gamma = 0) as published for librosa 0.10: early downsampling,
# per-octave STFT (rectangular window) times a sparsified frequency-domain constant-Q basis, recursive
# decimation by two, octave stacking and length scaling.  ONE DOCUMENTED DEVIATION: librosa resamples with
# libsoxr ('soxr_hq', a C library that is not available here: pass band to 0.913 of the new Nyquist, stop band about
# -125 dB); `cqt_resample2` below filters with a 41-tap half-band FIR, Kaiser beta 10 (stop band <= -99 dB from
# 0.66 pi, within 1.3e-5 up to 0.34 pi; round 2 used resample_poly's beta 5, -56 dB) and keeps librosa's length /
# sqrt(2) scaling rules.  Measured distance to a 301-tap -155 dB half-band: 2e-5 ... 5e-5 of the CQT's peak
# (tools/cqt_decimator_study.py).  Parity target for the device CQT is THIS restatement.
HANN_BANDWIDTH = 1.50018310546875          # librosa.filters.WINDOW_BANDWIDTHS['hann']


def note_c1_hz():
    """librosa.note_to_hz('C1') = 440 * 2**((24 - 69) / 12)."""
    return 440.0 * 2.0 ** ((24 - 69) / 12.0)


def cqt_frequencies(n_bins, fmin, bins_per_octave=12, tuning=0.0):
    corr = 2.0 ** (float(tuning) / bins_per_octave)
    return corr * fmin * 2.0 ** (np.arange(n_bins, dtype=np.float64) / bins_per_octave)


def cqt_decimation_taps():
    """The octave decimator: firwin(41, 0.5, window=('kaiser', 10.0)), a half-band FIR."""
    return scipy.signal.firwin(41, 0.5, window=("kaiser", 10.0))


def cqt_resample2(y):
    """Decimate by two: stand-in for librosa.resample(y, orig_sr=2, target_sr=1, res_type='soxr_hq', scale=True):
    z[n] = sum_j h[j] y[2 n + 20 - j] (zero outside the signal), ceil(len / 2) outputs."""
    y = np.asarray(y, dtype=np.float64)
    h = cqt_decimation_taps()
    n = int(np.ceil(y.shape[-1] * 0.5))
    z = np.convolve(y, h)[(len(h) - 1) // 2:][::2][:n]
    if z.shape[-1] < n:
        z = np.pad(z, (0, n - z.shape[-1]))
    return z * np.sqrt(2.0)                # scale=True: y_hat /= sqrt(ratio), ratio = 1/2


def _wavelet_lengths(freqs, sr, filter_scale, alpha):
    Q = float(filter_scale) / alpha
    cutoff = np.max(freqs * (1 + 0.5 * HANN_BANDWIDTH / Q))
    return Q * sr / freqs, cutoff


def cqt_filter_fft(sr, freqs, filter_scale, alpha, sparsity=0.01):
    """Frequency-domain basis of one octave: (fft_basis complex128 [n_filters, n_fft//2+1], n_fft)."""
    lengths, _ = _wavelet_lengths(freqs, sr, filter_scale, alpha)
    max_len = int(2.0 ** np.ceil(np.log2(lengths.max())))
    basis = np.zeros((len(freqs), max_len), dtype=np.complex128)
    for i, (ilen, f) in enumerate(zip(lengths, freqs)):
        t = np.arange(-ilen // 2, ilen // 2, dtype=np.float64)
        sig = np.exp(1j * 2 * np.pi * f / sr * t)
        sig = sig * scipy.signal.get_window("hann", len(sig), fftbins=True)
        sig = sig / np.sum(np.abs(sig))                      # norm=1
        lp = (max_len - len(sig)) // 2
        basis[i, lp:lp + len(sig)] = sig                     # util.pad_center
    n_fft = max_len
    basis *= lengths[:, None] / float(n_fft)
    fb = np.fft.fft(basis, n=n_fft, axis=1)[:, :n_fft // 2 + 1]
    # util.sparsify_rows(quantile=sparsity)
    mags = np.abs(fb)
    norms = mags.sum(axis=1, keepdims=True)
    srt = np.sort(mags, axis=1)
    cum = np.cumsum(srt / norms, axis=1)
    tidx = np.argmin(cum < sparsity, axis=1)
    out = np.zeros_like(fb)
    for i, j in enumerate(tidx):
        keep = mags[i] >= srt[i, j]
        out[i, keep] = fb[i, keep]
    return out, n_fft


def cqt_plan(sr, hop_length=512, fmin=None, n_bins=84, bins_per_octave=12, filter_scale=1.0, sparsity=0.01):
    """Everything that does not depend on the signal: octave schedule, bases (scalings folded in)."""
    if fmin is None:
        fmin = note_c1_hz()
    n_oct = int(np.ceil(float(n_bins) / bins_per_octave))
    n_filters = min(bins_per_octave, n_bins)
    freqs = cqt_frequencies(n_bins, fmin, bins_per_octave)
    r = 2.0 ** (2.0 / bins_per_octave)
    alpha = (r - 1) / (r + 1)
    lengths_full, cutoff = _wavelet_lengths(freqs, sr, filter_scale, alpha)
    nyq = sr / 2.0
    if cutoff > nyq:
        raise ValueError(f"Wavelet basis with max frequency={freqs.max()} would exceed the Nyquist frequency={nyq}. "
                         "Try reducing the number of frequency bins.")
    # early downsampling
    c1 = max(0, int(np.ceil(np.log2(nyq / cutoff)) - 1) - 1)
    twos = 0
    h = hop_length
    while h > 0 and h % 2 == 0:
        twos += 1; h //= 2
    c2 = max(0, twos - n_oct + 1)
    early = min(c1, c2)
    if twos < n_oct - 1:
        raise ValueError(f"hop_length must be a positive integer multiple of 2^{n_oct - 1} for {n_oct}-octave CQT")
    sr0 = sr / float(2 ** early)
    hop0 = hop_length // (2 ** early)
    octs = []
    my_sr, my_hop = sr0, hop0
    for i in range(n_oct):
        sl = slice(-n_filters, None) if i == 0 else slice(-n_filters * (i + 1), -n_filters * i)
        fo = freqs[sl]
        fb, n_fft = cqt_filter_fft(my_sr, fo, filter_scale, alpha, sparsity)
        fb = fb * np.sqrt(sr0 / my_sr)
        octs.append({"basis": fb, "n_fft": n_fft, "hop": my_hop, "sr": my_sr, "n": len(fo)})
        if my_hop % 2 == 0:
            my_hop //= 2
            my_sr /= 2.0
    # scale=True: lengths are taken at the (early-downsampled) rate
    lengths_s, _ = _wavelet_lengths(freqs, sr0, filter_scale, alpha)
    return {"early": early, "octaves": octs, "n_bins": n_bins, "scale": 1.0 / np.sqrt(lengths_s), "freqs": freqs}


def cqt(y, sr, hop_length=512, fmin=None, n_bins=84, bins_per_octave=12, filter_scale=1.0, sparsity=0.01):
    """complex128 [n_bins, 1 + len(y)//hop_length] (librosa.cqt layout), see the deviation note above."""
    y = np.asarray(y, dtype=np.float64)
    if y.ndim != 1:
        raise ValueError("Input data must be a 1D array.")
    plan = cqt_plan(sr, hop_length, fmin, n_bins, bins_per_octave, filter_scale, sparsity)
    my_y = y
    for _ in range(plan["early"]):
        my_y = cqt_resample2(my_y)
    resp = []
    for i, o in enumerate(plan["octaves"]):
        D = stft(my_y, n_fft=o["n_fft"], hop_length=o["hop"], window="boxcar", center=True)
        resp.append(o["basis"] @ D)
        if i + 1 < len(plan["octaves"]) and plan["octaves"][i + 1]["hop"] != o["hop"]:
            my_y = cqt_resample2(my_y)
    max_col = min(c.shape[1] for c in resp)
    out = np.zeros((n_bins, max_col), dtype=np.complex128)
    end = n_bins
    for c in resp:                                           # __trim_stack: highest octave first
        n_oct = c.shape[0]
        if end < n_oct:
            out[:end] = c[-end:, :max_col]
        else:
            out[end - n_oct:end] = c[:, :max_col]
        end -= n_oct
    return out * plan["scale"][:, None]
