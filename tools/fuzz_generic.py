"""Randomised fuzz of the non-fused rows (FFT/IFFT any length, STFT any power-of-two n_fft, Welch, sosfiltfilt,
time-domain frame features) against the float64 oracle (development aid)."""
import sys
import numpy as np, torch
sys.path.insert(0, ".")
from sygnals_amd import ops
from sygnals_amd.core import dsp as D, filters as Fm
from oracle import cpu_ref as O

def rel(a, b):
    return float(np.max(np.abs(np.asarray(a) - np.asarray(b))) / max(float(np.max(np.abs(b))), 1e-30))

rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
N = int(sys.argv[2]) if len(sys.argv) > 2 else 60
worst = {}
def check(name, e, tol, info):
    worst[name] = max(worst.get(name, 0.0), e)
    assert e <= tol, (name, e, info)

for it in range(N):
    n = int(rng.choice([rng.integers(2, 300), rng.integers(300, 5000), 2 ** rng.integers(1, 15), rng.integers(5000, 40000)]))
    x = rng.normal(0, 1, n)
    win = rng.choice(["hann", "hamming", None])
    nn = None if rng.random() < 0.6 else int(rng.integers(max(2, n // 2), 2 * n))
    f, X = D.compute_fft(x, 1000.0, nn, win)
    fr, Xr = O.compute_fft(x.astype(np.float32).astype(np.float64), 1000.0, nn, win)
    check("fft", rel(X, Xr), 2e-5 if n > 8192 else 1e-5, (n, nn, win))
    xi = D.compute_ifft(Xr.astype(np.complex64).astype(np.complex128))
    check("ifft", rel(xi, O.compute_ifft(Xr.astype(np.complex64).astype(np.complex128))), 2e-5 if n > 8192 else 1e-5, (n, nn))
    # STFT with a generic power-of-two frame
    nf = int(2 ** rng.integers(5, 13))
    hop = int(rng.integers(1, nf + 1))
    L = int(rng.integers(nf, 6 * nf + 50))
    y = rng.normal(0, 0.3, L).astype(np.float32).astype(np.float64)
    center = bool(rng.integers(0, 2))
    S = D.compute_stft(y, nf, hop, None, "hann", center)
    Sr = O.stft(y, nf, hop, nf, "hann", center)
    check("stft", rel(S, Sr), 1e-5, (nf, hop, L, center))
    # Welch
    nps = int(2 ** rng.integers(4, 12)); Lw = int(rng.integers(nps, 20 * nps))
    xw = rng.normal(0, 1, Lw).astype(np.float32).astype(np.float64) + 0.3
    nov = int(rng.integers(0, nps))
    fw, P = D.compute_psd_welch(xw, 100.0, "hann", nps, nov)
    fo, Po = O.compute_psd_welch(xw, 100.0, "hann", nps, nov)
    check("welch", rel(P, Po), 1e-5, (nps, nov, Lw))
    # sosfiltfilt
    order = int(rng.integers(1, 9)); fs = 8000.0
    kind = rng.choice(["lowpass", "highpass", "bandpass", "bandstop"])
    cut = float(rng.uniform(200, 3000)) if kind in ("lowpass", "highpass") else tuple(sorted(rng.uniform(150, 3500, 2) + np.array([0.0, 100.0])))
    if kind in ("bandpass", "bandstop") and order > 4:
        order = 4
    sos = O.design_butterworth_sos(cut, fs, order, kind)
    Ls = int(rng.integers(O.sosfiltfilt_padlen(sos) + 2, 6000))
    xs = rng.normal(0, 1, Ls).astype(np.float32).astype(np.float64)
    ys = Fm.apply_sos_filter(sos, xs)
    check("sosfiltfilt", rel(ys, O.apply_sos_filter(sos, xs)), 1e-5, (order, kind, cut, Ls))
    # time-domain frame features
    fl = int(rng.integers(2, 3000)); hp = int(rng.integers(1, fl + 1)); Lt = int(rng.integers(fl, 4 * fl + 10))
    yt = (rng.normal(0, 0.2, Lt) + rng.uniform(-0.3, 0.3)).astype(np.float32)
    ct = bool(rng.integers(0, 2)); nb = int(rng.integers(1, 40))
    st = ops.frame_stats(ops.to_device_f32(yt[None, :]), fl, hp, ct, nb).cpu().numpy()[0].astype(np.float64)
    ref = O.time_features_frames(yt.astype(np.float64), fl, hp, ct, nb)
    tr = len(ref["mean_amplitude"])          # odd frame_length: the reference may form one frame fewer (NaN-padded)
    assert tr in (st.shape[1], st.shape[1] - 1)
    for r, nm in enumerate(O.TIME_FEATURES):
        check(nm, rel(st[r, :tr], ref[nm]), 1e-5, (fl, hp, Lt, ct, nb))
    z = O.zero_crossing_rate(yt.astype(np.float64), fl, hp, ct)
    assert np.array_equal(np.round(st[8, :len(z)] * fl), np.round(z * fl)), ("zcr", fl, hp, Lt, ct)
print("fuzz ok:", {k: f"{v:.1e}" for k, v in worst.items()})
