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


MEL_MIN_STEPS = 28      # the kernel issues 7 groups of 4 MFMA steps unconditionally
MEL_MAX_STEPS = 1088    # positions of one skewed power row (P_STRIDE 1090, a multiple of 4 below it)


def row_pos(k):
    """Word position of bin k inside a skewed LDS power row of the fused kernel: one pad word after every 16 bins."""
    return k + (k >> 4)


def pack_mel_plan(basis: np.ndarray, waves: int = WAVES):
    """Block-sparse packing of a [M, F] filterbank for v_mfma_f32_4x4x1_16b_f32 (sixteen 4 x 4 blocks per instruction).

    The rows are taken in groups of four; only the non-zero column range of a group is kept and cut into chunks; chunk c
    is slot c: wave c // 4 runs its four slots side by side, one ROW POSITION per slot and MFMA step.  A slot walks
    consecutive words of the kernel's skewed power rows (row_pos: a pad word after every 16 bins, which gets a zero
    weight), so the kernel addresses its B operands with one base and immediate offsets.  `steps` (positions per slot)
    is the smallest multiple of 4 >= 28 for which the chunks fit the 4 * waves slots.  Returns (wpacked float32, plan
    int32):

      wpacked  [waves][steps / 4][64 lanes][4]  A operands, four steps per 16-byte load: lane l of step i carries
               basis[4 g + (l & 3)][bin at position p0 + i] of its slot s = l >> 4 (the four frame groups (l >> 2) & 3
               see the same weight), zero at pad positions, past the chunk and past the last mel row; then 8 groups of
               zero rows; then the int32 tables (bit patterns) [slot p0 | slot group | group first slot | group slot
               count], 64 entries each
      plan     {2, waves, steps, n_groups, table_off (in floats)}
    """
    basis = np.asarray(basis, dtype=np.float32)
    M, F = basis.shape
    if waves not in (8, 16):
        raise ValueError("waves must be 8 or 16")
    ng = (M + 3) // 4
    if ng > 64:
        raise ValueError(f"n_mels={M} needs {ng} groups of four rows (max 64: n_mels <= 256)")
    nslots = 4 * waves
    rng = []
    for g in range(ng):
        cols = np.flatnonzero(np.any(basis[4 * g:4 * g + 4] != 0, axis=0))
        rng.append((int(cols[0]), int(cols[-1]) + 1) if cols.size else (0, 0))

    def chunks_of(a, b, steps):
        """Chunks of the position range [row_pos(a), row_pos(b - 1)] in pieces of `steps` positions."""
        if b <= a:
            return []
        p, pe = row_pos(a) & ~1, row_pos(b - 1) + 1          # even start: the kernel reads two positions per LDS word pair
        return [(q, min(q + steps, pe)) for q in range(p, pe, steps)]

    # every non-empty group needs a slot of its own whatever the chunk length; positions per slot are bounded by a row
    n_nonempty = sum(1 for a, b in rng if b > a)
    if n_nonempty > nslots:
        raise ValueError(f"filterbank has {n_nonempty} non-empty groups of four mel rows, a {waves}-wave plan holds "
                         f"{nslots} slots (n_mels <= {4 * nslots} at most): use the dense filterbank path")
    steps = MEL_MIN_STEPS
    while sum(len(chunks_of(a, b, steps)) for a, b in rng) > nslots:
        steps += 4
        if steps > MEL_MAX_STEPS:
            raise ValueError(f"filterbank does not fit {nslots} slots of at most {MEL_MAX_STEPS} row positions")
    slot_p0 = np.zeros(64, np.int32); slot_g = -np.ones(64, np.int32)
    g_first = np.zeros(64, np.int32); g_cnt = np.zeros(64, np.int32)
    wts = np.zeros((nslots, steps, 4), np.float32)             # [slot][step][row]
    c = 0
    for g, (a, b) in enumerate(rng):
        g_first[g] = c
        for (p0, p1) in chunks_of(a, b, steps):                 # ascending positions = ascending bins
            slot_p0[c] = p0; slot_g[c] = g
            for i in range(p1 - p0):
                p = p0 + i
                if p % 17 == 16:                                # pad word of the skewed row
                    continue
                k = p - p // 17
                if a <= k < b:                                  # (a chunk may start one position below its range)
                    rows = basis[4 * g:min(4 * g + 4, M), k]
                    wts[c, i, :rows.shape[0]] = rows
            c += 1
        g_cnt[g] = c - g_first[g]
    assert c <= nslots
    lane = np.arange(64)
    # [wave][step][lane] = wts[wave * 4 + (lane >> 4)][step][lane & 3]
    A = wts.reshape(waves, 4, steps, 4)[:, lane >> 4, :, lane & 3]          # -> [64 lanes, waves, steps]
    A = np.transpose(A, (1, 2, 0))                                          # [waves][steps][64]
    # device layout: [wave][group of 4 steps][lane][step in group]
    A = A.reshape(waves, steps // 4, 4, 64).transpose(0, 1, 3, 2).reshape(-1)
    tail = np.zeros(8 * 64 * 4, np.float32)
    table = np.concatenate([slot_p0, slot_g, g_first, g_cnt]).astype(np.int32)
    wpacked = np.concatenate([A.astype(np.float32), tail, table.view(np.float32)])
    plan = np.array([2, waves, steps, ng, A.size + tail.size], dtype=np.int32)
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
