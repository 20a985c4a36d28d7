"""CPU study of the CQT's stated deviation: the octave decimator is a Kaiser half-band FIR, librosa's is
libsoxr 'soxr_hq' (pass band to 0.913 of the new Nyquist, stop band about -125 dB).  Prints (a) the analytic response
figures of candidate half-band filters and (b) the end-to-end difference of the oracle's CQT run with each of them
against a long reference-grade half-band (301 taps, -150 dB), on the C5 recipe and on a wide-band worst case.
    python3 tools/cqt_decimator_study.py [seconds]"""
import sys
import numpy as np
import scipy.signal
sys.path.insert(0, ".")
from oracle import cpu_ref as O
from sygnals_amd.synth import synth_stream

def halfband(ntaps, beta):
    h = scipy.signal.firwin(ntaps, 0.5, window=("kaiser", beta))
    h[np.abs(h) < 1e-15 * np.abs(h).max()] = 0.0
    return h

def response(h, lo, hi, n=8192):
    w = np.linspace(lo * np.pi, hi * np.pi, n)
    return np.abs(scipy.signal.freqz(h, worN=w)[1])

CAND = {"kaiser41_b5 (round 2)": halfband(41, 5.0), "kaiser41_b12.8 (round 3)": halfband(41, 12.8), "kaiser27_b12.8": halfband(27, 12.8), "kaiser81_b9": halfband(81, 9.0), "kaiser121_b12.3": halfband(121, 12.27),
        "kaiser161_b13": halfband(161, 13.0)}
REF = halfband(301, 15.0)
print("filter                    nonzero  |H|-1 on [0,.17pi] [0,.4pi]   max|H| on [.6pi,pi]  [.83pi,pi]")
for name, h in list(CAND.items()) + [("reference 301 taps", REF)]:
    pb1 = np.max(np.abs(response(h, 0, 0.17) - 1)); pb2 = np.max(np.abs(response(h, 0, 0.4) - 1))
    sb1 = np.max(response(h, 0.6, 1.0)); sb2 = np.max(response(h, 0.83, 1.0))
    print(f"{name:26s} {int((h != 0).sum()):4d}   {pb1:9.2e}      {pb2:9.2e}   {sb1:9.2e} ({20*np.log10(sb1):6.1f} dB)  {sb2:9.2e} ({20*np.log10(sb2):6.1f} dB)")

def cqt_with(y, sr, taps):
    def res2(x):
        n = int(np.ceil(x.shape[-1] * 0.5))
        z = scipy.signal.upfirdn(taps, np.asarray(x, np.float64), 1, 2)[(len(taps) - 1) // 2 // 1:][::1]
        # upfirdn(h, x, 1, 2) = (h * x)[::2]; centre the filter: drop (ntaps-1)/2 input samples of delay
        full = np.convolve(np.asarray(x, np.float64), taps)[(len(taps) - 1) // 2:]
        z = full[::2][:n]
        if z.shape[-1] < n:
            z = np.pad(z, (0, n - z.shape[-1]))
        return z * np.sqrt(2.0)
    old = O.cqt_resample2
    O.cqt_resample2 = res2
    try:
        return O.cqt(y, sr)
    finally:
        O.cqt_resample2 = old

secs = float(sys.argv[1]) if len(sys.argv) > 1 else 20.0
sr = 48000
rng = np.random.default_rng(1)
signals = {"C5 recipe (pink-ish noise + slow chirp)": synth_stream(int(secs * sr), sr).astype(np.float64),
           "white noise (wide-band worst case)": rng.normal(0, 0.2, int(secs * sr))}
for sname, y in signals.items():
    Cref = np.abs(cqt_with(y, sr, REF))
    # the scipy.resample_poly form the oracle ships must equal the 41-tap candidate
    chk = np.abs(O.cqt(y, sr))
    print(f"\n{sname}: {secs:.0f} s, |CQT| peak {Cref.max():.4g}")
    for name, h in CAND.items():
        C = np.abs(cqt_with(y, sr, h))
        d = np.abs(C - Cref)
        edge = int(1.5 * sr / 512)                          # 1.5 s at either end: the filters' own start / end transients
        d_in = d[:, edge:-edge]
        print(f"  {name:24s} interior (edges of 1.5 s left out): max|dC|/peak {d_in.max() / Cref.max():.2e}   per octave "
              + " ".join(f"{d_in[12 * o:12 * o + 12].max() / Cref.max():.1e}" for o in range(7)))
        per_oct = [d[12 * o:12 * o + 12].max() / Cref.max() for o in range(7)]
        rel_cell = np.max(d / np.maximum(Cref, 1e-3 * Cref.max()))
        print(f"  {name:24s} max|dC|/peak {d.max() / Cref.max():.2e}   cell-relative (cells > 1e-3 peak) {rel_cell:.2e}   per octave (low->high) "
              + " ".join(f"{v:.1e}" for v in per_oct))
    print(f"  (oracle as shipped vs kaiser41 candidate: {np.abs(chk - np.abs(cqt_with(y, sr, CAND['kaiser41_b5 (round 2)']))).max() / Cref.max():.1e})")
