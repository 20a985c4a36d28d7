"""Host-side (float64 NumPy) construction of the constant tables the kernels consume:
analysis windows, twiddles, the Slaney mel filterbank and its block-sparse MFMA packing,
DCT-II rows, lifter, spectral-contrast band plan.  Pure host logic, no GPU needed.

Behaviour follows what the reference obtains from librosa / SciPy at
sygnals/core/features/manager.py:184-227, cepstral.py:106-115 and
frequency_domain.py:200-207 (librosa>=0.10 semantics).
"""
from __future__ import annotations

import functools

import numpy as np
import scipy.signal

WAVES = 16          # default waves per workgroup of the fused kernel (stft_mel.hip): 8 or 16
MAXW = 16
MAX_BANDS = 16      # SYG_MAX_BANDS


def analysis_window(window, win_length: int, n_fft: int) -> np.ndarray:
    """Periodic window (fftbins=True) of win_length, centre-padded to n_fft, float64."""
    if isinstance(window, np.ndarray) or isinstance(window, (list,)):
        w = np.asarray(window, dtype=np.float64)
        if w.ndim != 1 or w.shape[0] != win_length:
            raise ValueError(f"window array must be 1-D of length win_length={win_length}")
    else:
        w = scipy.signal.get_window(window, win_length, fftbins=True).astype(np.float64)
    if win_length > n_fft:
        raise ValueError(f"win_length={win_length} must be <= n_fft={n_fft}")
    if win_length < n_fft:
        lp = (n_fft - win_length) // 2
        w = np.concatenate([np.zeros(lp), w, np.zeros(n_fft - win_length - lp)])
    return w


def twiddles(n: int) -> np.ndarray:
    """W_n^k = exp(-2*pi*i*k/n), k = 0..n-1, as float32 [n, 2] (evaluated in float64)."""
    k = np.arange(n, dtype=np.float64)
    ang = -2.0 * np.pi * k / n
    return np.stack([np.cos(ang), np.sin(ang)], axis=1).astype(np.float32)


def _hz_to_mel(f):
    f = np.asarray(f, dtype=np.float64)
    lin = f * (3.0 / 200.0)
    with np.errstate(divide="ignore"):
        log = 15.0 + np.log(np.maximum(f, 1e-300) / 1000.0) * (27.0 / np.log(6.4))
    return np.where(f >= 1000.0, log, lin)


def _mel_to_hz(m):
    m = np.asarray(m, dtype=np.float64)
    return np.where(m >= 15.0, 1000.0 * np.exp((np.log(6.4) / 27.0) * (m - 15.0)), m * (200.0 / 3.0))


def mel_filterbank(sr: float, n_fft: int, n_mels: int = 128, fmin: float = 0.0, fmax=None) -> np.ndarray:
    """Slaney-scale, Slaney-normalised triangular filterbank, float32 [n_mels, 1+n_fft//2].

    librosa stores the basis in float32 (and applies the area normalisation to the float32
    triangles); both roundings are reproduced because the float64 reference pipeline
    multiplies by exactly these float32 values.
    """
    if fmax is None:
        fmax = sr / 2.0
    if n_mels < 1:
        raise ValueError("n_mels must be >= 1")
    freqs = np.fft.rfftfreq(n_fft, 1.0 / sr)
    edges = _mel_to_hz(np.linspace(_hz_to_mel(fmin), _hz_to_mel(fmax), n_mels + 2))
    d = np.diff(edges)
    ramps = edges[:, None] - freqs[None, :]
    up = -ramps[:-2] / d[:-1, None]
    down = ramps[2:] / d[1:, None]
    tri = np.maximum(0.0, np.minimum(up, down)).astype(np.float32)
    enorm = 2.0 / (edges[2:] - edges[:-2])
    return (tri.astype(np.float64) * enorm[:, None]).astype(np.float32)


def pack_mel_plan(basis: np.ndarray, waves: int = WAVES):
    """Block-sparse packing of a [M, F] filterbank for v_mfma_f32_16x16x4_f32.

    Rows are grouped in tiles of 16; for each tile only its non-zero column range is kept.
    The tiles' 4-bin k-steps are split over `waves` contiguous segments (one per wave,
    balanced greedily).  Returns (wpacked float32 [steps, 64], plan int32 [2 + 4*16] =
    {n_tiles, n_waves, tile[16], k0[16], nsteps[16], woff[16]})
    where A[m = l & 15][k = l >> 4] of step i = basis[16*tile + m][k0 + 4*i + k]; rows of `wpacked` are
    stored four steps at a time as [group][lane][4] so that a lane fetches four steps with one 16-byte load.
    """
    basis = np.asarray(basis, dtype=np.float32)
    M, F = basis.shape
    nt = (M + 15) // 16
    if waves not in (8, 16):
        raise ValueError("waves must be 8 or 16")
    if nt > waves:
        raise ValueError(f"n_mels={M} needs {nt} tiles > {waves} waves (max n_mels {16 * waves})")
    lo = np.zeros(nt, int); hi = np.zeros(nt, int)
    for t in range(nt):
        cols = np.flatnonzero(np.any(basis[16 * t:16 * t + 16] != 0, axis=0))
        if cols.size:
            lo[t], hi[t] = (cols[0] // 16) * 16, cols[-1] + 1    # segment starts are multiples of 16 bins (row skew)
    steps = (hi - lo + 3) // 4
    nw = np.ones(nt, int)
    for _ in range(waves - nt):
        nw[int(np.argmax(steps / nw))] += 1
    tile = -np.ones(MAXW, np.int32); k0 = np.zeros(MAXW, np.int32)
    ns = np.zeros(MAXW, np.int32); woff = np.zeros(MAXW, np.int32)
    blocks = []
    w = 0
    off = 0
    lane = np.arange(64)
    for t in range(nt):
        per = -(-steps[t] // nw[t]) if steps[t] else 0
        per = -(-per // 4) * 4                              # whole groups of 4 steps = 16 bins
        for j in range(nw[t]):
            s0 = min(j * per, steps[t]); s1 = min((j + 1) * per, steps[t])
            n4 = -(-(s1 - s0) // 4) * 4                     # padded with zero-weight steps to a multiple of 4
            tile[w] = t; k0[w] = (lo[t] + 4 * s0) if s1 > s0 else 0; ns[w] = n4; woff[w] = off
            if s1 > s0:
                rows = 16 * t + (lane & 15)[None, :]
                cols = (lo[t] + 4 * np.arange(s0, s0 + n4))[:, None] + (lane >> 4)[None, :]
                ok = (rows < M) & (cols < F) & (np.arange(s0, s0 + n4) < s1)[:, None]
                blk = np.where(ok, basis[np.minimum(rows, M - 1), np.minimum(cols, F - 1)], 0.0).astype(np.float32)
                # device layout: [group of 4 steps][lane][step in group] -> one float4 per lane per group
                blocks.append(blk.reshape(n4 // 4, 4, 64).transpose(0, 2, 1).reshape(n4, 64))
                off += n4
            w += 1
    blocks.append(np.zeros((32, 64), np.float32))      # tail: the kernel pre-loads 5 groups unconditionally
    wpacked = np.concatenate(blocks, axis=0)
    plan = np.concatenate([[nt, waves], tile, k0, ns, woff]).astype(np.int32)
    return np.ascontiguousarray(wpacked), plan


def dct_matrix(n_out: int, n_in: int, dct_type: int = 2, norm="ortho") -> np.ndarray:
    """Rows k < n_out of scipy.fft.dct(type=dct_type, norm=norm) as a float32 matrix [n_out, n_in]."""
    import scipy.fft
    if n_out > n_in:
        raise ValueError(f"n_mfcc={n_out} cannot exceed n_mels={n_in}")
    D = scipy.fft.dct(np.eye(n_in), axis=0, type=dct_type, norm=norm)[:n_out]
    return np.ascontiguousarray(D.astype(np.float32))


def lifter_weights(n_mfcc: int, lifter: float):
    if lifter < 0:
        raise ValueError(f"MFCC lifter={lifter} must be a non-negative number")
    if lifter == 0:
        return None
    return (1.0 + (lifter / 2.0) * np.sin(np.pi * np.arange(1, n_mfcc + 1) / lifter)).astype(np.float32)


def contrast_plan(freqs: np.ndarray, sr: float, n_bands: int = 6, fmin: float = 200.0, quantile: float = 0.02):
    """Band bin ranges [lo, hi) and tail sizes k of librosa.feature.spectral_contrast.

    Octave edges [0, fmin, 2 fmin, ...]; bands after the first include the bin below;
    the last band runs to Nyquist; every other band drops its top bin; k is taken from
    the bin count *before* that drop.
    """
    freqs = np.atleast_1d(np.asarray(freqs, dtype=np.float64))
    if n_bands < 1 or int(n_bands) != n_bands:
        raise ValueError("n_bands must be a positive integer")
    if n_bands + 1 > MAX_BANDS:
        raise ValueError(f"n_bands must be <= {MAX_BANDS - 1}")
    if not 0.0 < quantile < 1.0:
        raise ValueError("quantile must lie in the range (0, 1)")
    if fmin <= 0:
        raise ValueError("fmin must be a positive number")
    octa = np.concatenate([[0.0], fmin * 2.0 ** np.arange(n_bands + 1)])
    if np.any(octa[:-1] >= 0.5 * sr):
        raise ValueError("Frequency band exceeds Nyquist. Reduce either fmin or n_bands.")
    lo = np.zeros(MAX_BANDS, np.int32); hi = np.zeros(MAX_BANDS, np.int32); kk = np.zeros(MAX_BANDS, np.int32)
    for b in range(n_bands + 1):
        idx = np.flatnonzero((freqs >= octa[b]) & (freqs <= octa[b + 1]))
        first, last = int(idx[0]), int(idx[-1])
        if b > 0:
            first -= 1
        if b == n_bands:
            last = len(freqs) - 1
        count = last - first + 1
        if b < n_bands:
            last -= 1
        lo[b], hi[b], kk[b] = first, last + 1, max(int(np.rint(quantile * count)), 1)
    return np.concatenate([[n_bands + 1], lo, hi, kk]).astype(np.int32)


def butter_padlen(sos: np.ndarray) -> int:
    sos = np.asarray(sos, dtype=np.float64)
    ntaps = 2 * sos.shape[0] + 1
    ntaps -= min(int((sos[:, 2] == 0).sum()), int((sos[:, 5] == 0).sum()))
    return 3 * ntaps
