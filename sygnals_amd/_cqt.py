"""Host-side plan of the constant-Q transform (float64 NumPy): octave schedule, frequency-domain bases with
every constant scaling folded in, decimation filter.  Follows the recursive algorithm librosa.cqt / vqt
(gamma = 0) publishes for librosa >= 0.10, as reached from compute_cqt (sygnals/core/dsp.py:231-289).

Resampler: librosa decimates with libsoxr ('soxr_hq': pass band to 0.913 of the new Nyquist, stop band about -125 dB),
which is not reproducible here; the device path decimates with a 41-tap half-band FIR, Kaiser beta 10 (round 3; round
2 used scipy.signal.resample_poly's beta 5): stop band <= -99 dB from 0.66 pi, pass band within 1.3e-5 up to 0.34 pi --
the band the next octave's wavelets occupy after librosa's early-downsampling rule -- at the same 21 multiplies per
output.  Measured against a 301-tap, -155 dB half-band on the oracle (tools/cqt_decimator_study.py): 2.2e-5 of the
CQT's peak on the C5 recipe, 5e-5 on white noise (beta 5: 1.2e-3 and 2.8e-3); longer filters (81 ... 161 taps) stop at
1e-5: the rest is how each filter treats its transition band.  librosa's output-length and sqrt(2) amplitude
conventions are kept.
"""
from __future__ import annotations

import numpy as np
import scipy.signal

_HANN_BW = 1.50018310546875      # equivalent noise bandwidth librosa tabulates for 'hann'


def c1_hz() -> float:
    return 440.0 * 2.0 ** ((24 - 69) / 12.0)          # note_to_hz('C1')


DECIMATOR_TAPS, DECIMATOR_BETA = 41, 10.0


def decimation_taps() -> np.ndarray:
    """The octave decimator: a 41-tap half-band FIR, Kaiser window beta 10 (module docstring).  A half-band filter is
    zero at every even offset from its centre; firwin leaves ~1e-18 there (sin(pi k) in floating point).  Those 20
    taps are set to exactly zero so that the device kernel skips them (syg_decimate2_f32 skips zero taps): 21 multiplies
    per output instead of 41, a change of 1e-18 relative in the filter."""
    taps = scipy.signal.firwin(DECIMATOR_TAPS, 0.5, window=("kaiser", DECIMATOR_BETA))
    taps[np.abs(taps) < 1e-15 * np.abs(taps).max()] = 0.0
    return taps


class CqtPlan:
    def __init__(self, sr, hop_length=512, fmin=None, n_bins=84, bins_per_octave=12, tuning=0.0, filter_scale=1.0,
                 sparsity=0.01):
        if fmin is None:
            fmin = c1_hz()
        if n_bins < 1 or bins_per_octave < 1:
            raise ValueError("n_bins and bins_per_octave must be positive")
        bpo = int(bins_per_octave)
        self.n_bins = int(n_bins)
        n_oct = -(-self.n_bins // bpo)
        n_filt = min(bpo, self.n_bins)
        freqs = fmin * 2.0 ** (float(tuning) / bpo) * 2.0 ** (np.arange(self.n_bins) / bpo)
        r = 2.0 ** (2.0 / bpo)
        alpha = (r - 1.0) / (r + 1.0)
        Q = float(filter_scale) / alpha
        cutoff = float(np.max(freqs * (1 + 0.5 * _HANN_BW / Q)))
        nyq = sr / 2.0
        if cutoff > nyq:
            raise ValueError(f"Wavelet basis with max frequency={freqs.max()} would exceed the Nyquist "
                             f"frequency={nyq}. Try reducing the number of frequency bins.")
        twos = (hop_length & -hop_length).bit_length() - 1 if hop_length > 0 else 0
        if hop_length <= 0 or twos < n_oct - 1:
            raise ValueError(f"hop_length must be a positive integer multiple of 2^{n_oct - 1} for {n_oct}-octave CQT")
        early = min(max(0, int(np.ceil(np.log2(nyq / cutoff)) - 1) - 1), max(0, twos - n_oct + 1))
        self.early = early
        sr0 = sr / float(2 ** early)
        hop = hop_length >> early
        self.octaves = []
        my_sr = sr0
        row_hi = self.n_bins
        for i in range(n_oct):
            lo = max(0, self.n_bins - n_filt * (i + 1)) if i > 0 else self.n_bins - n_filt
            hi = self.n_bins - n_filt * i
            fo = freqs[lo:hi] if i > 0 else freqs[-n_filt:]
            lengths = Q * my_sr / fo
            n_fft = 1 << int(np.ceil(np.log2(lengths.max())))
            basis = np.zeros((len(fo), n_fft), dtype=np.complex128)
            for j, (ln, f) in enumerate(zip(lengths, fo)):
                t = np.arange(-ln // 2, ln // 2, dtype=np.float64)
                sig = np.exp(2j * np.pi * f / my_sr * t) * scipy.signal.get_window("hann", len(t), fftbins=True)
                sig /= np.abs(sig).sum()
                lp = (n_fft - len(sig)) // 2
                basis[j, lp:lp + len(sig)] = sig
            fb = np.fft.fft(basis * (lengths[:, None] / n_fft), axis=1)[:, : n_fft // 2 + 1]
            mag = np.abs(fb)
            srt = np.sort(mag, axis=1)
            cum = np.cumsum(srt / mag.sum(axis=1, keepdims=True), axis=1)
            thr = srt[np.arange(len(fo)), np.argmin(cum < sparsity, axis=1)]
            fb = np.where(mag >= thr[:, None], fb, 0.0)
            fb *= np.sqrt(sr0 / my_sr)
            # rows of the stacked output this octave fills (top octave first); the last octave may be clipped
            n_rows = len(fo)
            row0 = row_hi - n_rows
            self.octaves.append({"basis": fb, "n_fft": n_fft, "hop": hop, "row0": max(row0, 0),
                                 "skip": max(0, -row0), "n": n_rows})
            row_hi -= n_rows
            if hop % 2 == 0 and i + 1 < n_oct:
                hop //= 2
                my_sr /= 2.0
                self.octaves[-1]["decimate_after"] = True
            else:
                self.octaves[-1]["decimate_after"] = False
        # scale=True: divide by sqrt(filter length at the early-downsampled rate); folded into the bases
        scale = 1.0 / np.sqrt(Q * sr0 / freqs)
        for o in self.octaves:
            rows = np.arange(o["row0"], o["row0"] + o["n"] - o["skip"])
            o["basis"] = o["basis"][o["skip"]:] * scale[rows][:, None]
            o["n"] = len(rows)
        self.freqs = freqs
        self.hop_length = int(hop_length)

    def one_launch_shape(self) -> bool:
        """True when syg_cqt_fused_f32 can take the whole transform: hop_length 512, one early decimation, at most seven
        octaves at frame length 256 and hop 256 >> o with the same number (<= 16) of unclipped filters, and ONE operand
        table -- the octaves' bases are the same matrix up to rounding (the decimator's sqrt(2) and the scalings above
        cancel).  48 / 44.1 kHz with the default 84 bins qualify; 22.05 kHz (no early decimation) and hop 1024 do not."""
        oc = self.octaves
        if self.hop_length != 512 or self.early != 1 or not 1 <= len(oc) <= 7 or not 1 <= oc[0]["n"] <= 16:
            return False
        b0 = oc[0]["basis"]
        tol = 1e-12 * float(np.abs(b0).max())
        return all(o["n_fft"] == 256 and o["hop"] == (256 >> i) and o["skip"] == 0 and o["n"] == oc[0]["n"]
                   and o["basis"].shape == b0.shape and float(np.abs(o["basis"] - b0).max()) <= tol
                   for i, o in enumerate(oc))
