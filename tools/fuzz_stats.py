"""Randomised fuzz of the per-frame spectral statistics / contrast rows and the CQT against the oracle."""
import sys
import numpy as np, torch
sys.path.insert(0, ".")
from sygnals_amd import ops
from sygnals_amd.core.features.manager import extract_features_batch
from sygnals_amd.core.dsp import compute_cqt
from oracle import cpu_ref as O
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
N = int(sys.argv[2]) if len(sys.argv) > 2 else 60
worst = {}
def check(name, a, b, tol, info):
    e = float(np.max(np.abs(a - b)) / max(float(np.max(np.abs(b))), 1e-30))
    worst[name] = max(worst.get(name, 0.0), e)
    assert e <= tol, (name, e, info)
for it in range(N):
    sr = int(rng.choice([16000, 22050, 44100, 48000]))
    hop = int(rng.choice([128, 256, 512, 700]))
    L = int(rng.integers(3000, 20000))
    center = bool(rng.integers(0, 2)) or L < 2048
    Y = O.synth_clips(2, L, sr, seed=int(rng.integers(0, 1 << 30)))
    if it % 5 == 0:
        Y[1] *= 1e-4
    fmin = float(rng.choice([100.0, 200.0, 300.0]))
    nb = int(rng.integers(1, 7))
    while fmin * 2 ** nb >= sr / 2:
        nb -= 1
    q = float(rng.choice([0.01, 0.02, 0.05, 0.1]))
    roll = float(rng.choice([0.5, 0.85, 0.95]))
    p = float(rng.choice([1, 2, 3]))
    fp = {"spectral_contrast": {"n_bands": nb, "fmin": fmin, "quantile": q}, "spectral_rolloff": {"roll_percent": roll},
          "spectral_bandwidth": {"p": p}}
    feats = ["spectral_centroid", "spectral_bandwidth", "spectral_flatness", "spectral_rolloff", "dominant_frequency",
             "spectral_contrast"]
    out = extract_features_batch(ops.to_device_f32(Y), sr, feats, 2048, hop, center, feature_params=fp)
    for b in range(2):
        ref = O.extract_features(Y[b].astype(np.float64), sr, feats, 2048, hop, center, feature_params=fp)
        S = np.abs(O.stft(Y[b].astype(np.float64), 2048, hop, 2048, "hann", center))
        st = O.spectral_stats_frames(S, O.fft_frequencies(sr, 2048), roll, p)
        info = (sr, hop, L, center, nb, fmin, q, roll, p, b)
        check("centroid", out["spectral_centroid"][b], ref["spectral_centroid"], 1e-5, info)
        check("bandwidth", out["spectral_bandwidth"][b], ref["spectral_bandwidth"], 2e-5, info)
        check("flatness", out["spectral_flatness"][b], ref["spectral_flatness"], 5e-5, info)
        ok = st["rolloff_margin"] > 1e-6
        assert np.array_equal(out["spectral_rolloff"][b][ok], ref["spectral_rolloff"][ok]), ("rolloff", info)
        # contrast: the gate is on the raw tail means (1e-5 of the spectrogram peak, like the STFT itself); the dB
        # values take log10 of band valleys that can sit at the fp32 FFT noise floor and are
        # only tracked here
        for k in ref:
            if k.startswith("contrast"):
                e = float(np.max(np.abs(out[k][b] - ref[k])) / max(float(np.max(np.abs(ref[k]))), 1e-30))
                worst["contrast dB (tracked)"] = max(worst.get("contrast dB (tracked)", 0.0), e)
    from sygnals_amd import _tables as T
    fr = O.fft_frequencies(sr, 2048)
    plan = T.contrast_plan(fr, sr, nb, fmin, q)
    _, _, pvd = ops.stft2048_mel(ops.to_device_f32(Y), sr, hop, center, n_mels=16, contrast=plan)
    pv = pvd.cpu().numpy()
    for b in range(2):
        S = np.abs(O.stft(Y[b].astype(np.float64), 2048, hop, 2048, "hann", center))
        atol = 1e-5 * S.max()
        for k, (bins, kk) in enumerate(O.contrast_bands(fr, sr, nb, fmin, q)):
            srt = np.sort(S[bins], axis=0)
            ev = np.abs(pv[b, 1, k] - srt[:kk].mean(axis=0)).max() / S.max()
            ep = np.abs(pv[b, 0, k] - srt[-kk:].mean(axis=0)).max() / S.max()
            worst["contrast means / peak"] = max(worst.get("contrast means / peak", 0.0), ev, ep)
            assert max(ev, ep) <= 1e-5, ("contrast means", ev, ep, (sr, hop, L, center, nb, fmin, q, b, k, kk))
    if it % 6 == 0:
        Lc = int(rng.integers(20000, 60000)); src = 48000
        yc = O.synth_clips(1, Lc, src, seed=it)[0].astype(np.float64)
        nbins = int(rng.choice([36, 48, 60, 84])); hopc = int(rng.choice([256, 512, 1024]))
        try:
            C = compute_cqt(yc, src, hop_length=hopc, n_bins=nbins)
        except ValueError:
            continue
        Cr = O.cqt(yc, src, hop_length=hopc, n_bins=nbins)
        check("cqt", C, Cr, 1e-5, (Lc, nbins, hopc))
print("fuzz ok:", {k: f"{v:.1e}" for k, v in worst.items()})
