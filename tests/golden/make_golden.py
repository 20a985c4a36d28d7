#!/usr/bin/env python3
"""Generate golden vectors by EXECUTING the reference's own functions.

Run once in the build container (needs /root/reference; never runs on the GPU
box):   python tests/golden/make_golden.py

What is executed: the SciPy/NumPy-backed functions of
  sygnals/core/filters.py                      (loads as-is)
  sygnals/core/dsp.py                          (compute_fft/ifft, apply_window, compute_psd_welch; for
                                               ref_dsp2.npz: apply_convolution, compute_correlation,
                                               compute_autocorrelation, compute_psd_periodogram,
                                               amplitude_envelope(method='hilbert'))
  sygnals/core/ml_utils/{scaling,formatters}.py (apply_scaling and the three *_scale helpers, format_feature_sequences,
                                               format_features_as_image; load as-is: scikit-learn, SciPy and pandas are
                                               installed)
  sygnals/core/transforms.py                   (hilbert_transform; its top-level ``import pywt`` gets the same
                                               empty placeholder as librosa -- PyWavelets is not installed and no
                                               wavelet function is called)
  sygnals/core/features/frequency_domain.py    (the five per-frame functions)
  sygnals/core/features/time_domain.py         (the seven per-frame functions; loads as-is)
loaded BY FILE PATH (``sygnals/core/__init__.py`` pulls pandasql/soundfile,
which are not installed).  dsp.py and frequency_domain.py have a top-level
``import librosa``; librosa is not installed, so an EMPTY placeholder module is
registered under that name only to let the import statement pass.  The
placeholder defines nothing: any reference function that would touch librosa
raises AttributeError, and none of those is used below -- every number written
here comes out of the reference's code running on the real SciPy/NumPy.

Only inputs and outputs (data) are stored; no reference source is copied.
"""
import importlib.util
import os
import sys
import types

import numpy as np

REF = "/root/reference/sygnals/core"
OUT = os.path.dirname(os.path.abspath(__file__))


def load(name, rel):
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, rel))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_signal(rng, n, fs):
    t = np.arange(n) / fs
    f = rng.uniform(0.01 * fs, 0.4 * fs, 3)
    a = rng.uniform(0.1, 0.5, 3)
    ph = rng.uniform(0, 2 * np.pi, 3)
    y = (a[:, None] * np.sin(2 * np.pi * f[:, None] * t[None, :] + ph[:, None])).sum(0)
    return y + rng.normal(0, 0.05, n)


def main():
    if "librosa" not in sys.modules:
        sys.modules["librosa"] = types.ModuleType("librosa")  # empty placeholder, see docstring
    filters = load("ref_filters", "filters.py")
    dsp = load("ref_dsp", "dsp.py")
    fd = load("ref_freq", "features/frequency_domain.py")
    rng = np.random.default_rng(20250523)

    # ---- filters.py -----------------------------------------------------
    g = {}
    designs = [
        ("bp4_48k", (300.0, 3400.0), 48000.0, 4, "bandpass"),
        ("lp5_1k", 100.0, 1000.0, 5, "lowpass"),
        ("hp5_1k", 100.0, 1000.0, 5, "highpass"),
        ("bs5_1k", (100.0, 200.0), 1000.0, 5, "bandstop"),
        ("lp8_1k", 100.0, 1000.0, 8, "lowpass"),
        ("bp2_16k", (500.0, 2000.0), 16000.0, 2, "bandpass"),
    ]
    for name, cutoff, fs, order, kind in designs:
        sos = filters.design_butterworth_sos(cutoff, fs, order, kind)
        x = test_signal(rng, 3000 if fs > 2000 else 2000, fs)
        g[f"{name}_sos"] = sos
        g[f"{name}_x"] = x
        g[f"{name}_y"] = filters.apply_sos_filter(sos, x)
    x = test_signal(rng, 6000, 48000.0)
    g["conv_x"] = x
    g["conv_bp"] = filters.band_pass_filter(x, 300.0, 3400.0, 48000.0, order=4)
    g["conv_lp"] = filters.low_pass_filter(x, 4000.0, 48000.0)
    g["conv_hp"] = filters.high_pass_filter(x, 4000.0, 48000.0)
    g["conv_bs"] = filters.band_stop_filter(x, 1000.0, 5000.0, 48000.0)
    np.savez_compressed(os.path.join(OUT, "ref_filters.npz"), **g)

    # ---- dsp.py ---------------------------------------------------------
    g = {}
    x = test_signal(rng, 1000, 1000.0)
    g["x1000"] = x
    for win in ("hann", "hamming", "blackman", None):
        for n in (None, 1024, 512, 1500):
            fr, sp = dsp.compute_fft(x, fs=1000.0, n=n, window=win)
            key = f"fft_{win}_{n}"
            g[key + "_f"] = fr
            g[key + "_s"] = sp
    fr, sp = dsp.compute_fft(x, fs=1000.0, window=None)
    g["ifft_none"] = dsp.compute_ifft(sp)
    g["ifft_n768"] = dsp.compute_ifft(sp, n=768)
    g["ifft_n1200"] = dsp.compute_ifft(sp, n=1200)
    x2 = test_signal(rng, 4096, 48000.0)
    g["x4096"] = x2
    fr, sp = dsp.compute_fft(x2, fs=48000.0)
    g["fft4096_f"], g["fft4096_s"] = fr, sp
    for w in ("hann", "hamming", "blackman", "bartlett", "boxcar"):
        g[f"win_{w}"] = dsp.apply_window(x, w)
    x3 = test_signal(rng, 20000, 48000.0) + 0.3  # DC offset exercises detrend
    g["x20000"] = x3
    for tag, kw in (("w4096", dict(nperseg=4096)), ("w256", dict(nperseg=256)),
                    ("w1024o768", dict(nperseg=1024, noverlap=768)),
                    ("w512nfft1024", dict(nperseg=512, nfft=1024)),
                    ("w1024spec", dict(nperseg=1024, scaling="spectrum")),
                    ("w1024nodet", dict(nperseg=1024, detrend=False)),
                    ("w1024hamming", dict(nperseg=1024, window="hamming"))):
        f, p = dsp.compute_psd_welch(x3, fs=48000.0, **kw)
        g[f"welch_{tag}_f"], g[f"welch_{tag}_p"] = f, p
    np.savez_compressed(os.path.join(OUT, "ref_dsp.npz"), **g)

    # ---- frequency_domain.py -------------------------------------------
    g = {}
    F = 1025
    freqs = np.fft.rfftfreq(2048, 1 / 48000.0)
    spectra = np.abs(rng.normal(0, 1, (24, F))) * np.exp(-np.linspace(0, rng.uniform(1, 6), F))[None, :]
    spectra[3] = 0.0                       # all-zero frame
    spectra[4] = 0.0; spectra[4, 100] = 1.0  # single peak
    spectra[5] = 1.0                       # flat
    spectra[6] *= 1e-9                     # tiny but non-zero
    g["freqs"] = freqs
    g["spectra"] = spectra
    g["centroid"] = np.array([fd.spectral_centroid(s, freqs) for s in spectra])
    g["bandwidth"] = np.array([fd.spectral_bandwidth(s, freqs) for s in spectra])
    g["bandwidth_p1"] = np.array([fd.spectral_bandwidth(s, freqs, p=1) for s in spectra])
    g["bandwidth_c"] = np.array([fd.spectral_bandwidth(s, freqs, centroid=np.float64(5000.0)) for s in spectra])
    g["flatness"] = np.array([fd.spectral_flatness(s) for s in spectra])
    g["rolloff85"] = np.array([fd.spectral_rolloff(s, freqs) for s in spectra])
    g["rolloff50"] = np.array([fd.spectral_rolloff(s, freqs, roll_percent=0.5) for s in spectra])
    g["rolloff100"] = np.array([fd.spectral_rolloff(s, freqs, roll_percent=1.0) for s in spectra])
    g["rolloff0"] = np.array([fd.spectral_rolloff(s, freqs, roll_percent=0.0) for s in spectra])
    g["dominant"] = np.array([fd.dominant_frequency(s, freqs) for s in spectra])
    e = np.array([], dtype=np.float64)
    g["empty"] = np.array([fd.spectral_centroid(e, e), fd.spectral_bandwidth(e, e), fd.spectral_flatness(e),
                           fd.spectral_rolloff(e, e), fd.dominant_frequency(e, e)])
    np.savez_compressed(os.path.join(OUT, "ref_freq.npz"), **g)

    # ---- time_domain.py -------------------------------------------------
    td = load("ref_time", "features/time_domain.py")
    g = {}
    N = 512
    frames = rng.normal(0, 0.3, (20, N))
    frames[1] = 0.0                                   # all-zero frame
    frames[2] = 0.25                                  # constant
    frames[3] = np.sin(2 * np.pi * 5 * np.arange(N) / N)
    frames[4] = np.sign(frames[3])                    # square wave
    frames[5] = np.abs(frames[5]) ** 3                # strongly skewed
    frames[6] = frames[6] + 2.0                       # large offset
    frames[7] = 0.0; frames[7, 17] = 1.0              # impulse
    frames[8] = np.round(frames[8] * 4) / 4           # few distinct values (histogram edges hit exactly)
    frames[9] = np.linspace(-1.0, 1.0, N)             # uniform ramp: samples on histogram bin edges
    frames[10] *= 1e-7                                # tiny but non-zero
    frames = frames.astype(np.float32).astype(np.float64)   # fp32-representable: the device sees the same values
    g["frames"] = frames
    names = ["mean_amplitude", "std_dev_amplitude", "skewness", "kurtosis", "peak_amplitude", "crest_factor",
             "signal_entropy"]
    for nm in names:
        g[nm] = np.array([td.TIME_DOMAIN_FEATURES[nm](f) for f in frames])
    g["signal_entropy_b4"] = np.array([td.signal_entropy(f, num_bins=4) for f in frames])
    g["signal_entropy_b32"] = np.array([td.signal_entropy(f, num_bins=32) for f in frames])
    short = [np.array([], dtype=np.float64), np.array([0.5]), np.array([0.5, -0.25]), np.array([0.1, 0.2, 0.4])]
    for i, f in enumerate(short):
        g[f"short{i}"] = f
        g[f"short{i}_out"] = np.array([td.TIME_DOMAIN_FEATURES[nm](f) for nm in names])
    np.savez_compressed(os.path.join(OUT, "ref_time.npz"), **g)

    # ---- dsp.py / transforms.py: FFT-backed 1-D operations (own generator: earlier files stay bit-identical) ----
    if "pywt" not in sys.modules:
        sys.modules["pywt"] = types.ModuleType("pywt")        # empty placeholder, see docstring
    tr = load("ref_transforms", "transforms.py")
    rng2 = np.random.default_rng(20250524)
    g = {}
    sig = {"a1000": test_signal(rng2, 1000, 1000.0), "b50": rng2.normal(0, 1, 50), "c4096": test_signal(rng2, 4096, 48000.0),
           "d999": test_signal(rng2, 999, 1000.0) + 0.2, "k31": np.hanning(31) / np.hanning(31).sum(),
           "k200": rng2.normal(0, 0.1, 200), "k512": rng2.normal(0, 0.05, 512) * np.exp(-np.arange(512) / 90.0),
           "e7": np.array([0, 0, 1, 1, 1, 0, 0], dtype=float), "k2": np.array([1, -1], dtype=float),
           "one": np.array([0.75])}
    for k, v in sig.items():
        g["sig_" + k] = v
    for a, b in (("a1000", "k31"), ("b50", "k200"), ("c4096", "k512"), ("e7", "k2"), ("a1000", "one"), ("one", "k2")):
        for mode in ("full", "same", "valid"):
            g[f"conv_{a}_{b}_{mode}"] = dsp.apply_convolution(sig[a], sig[b], mode=mode)
    for a, b in (("a1000", "d999"), ("b50", "k200"), ("k200", "b50"), ("e7", "k2"), ("c4096", "k512")):
        for mode in ("full", "same", "valid"):
            g[f"corr_{a}_{b}_{mode}"] = dsp.compute_correlation(sig[a], sig[b], mode=mode)
    g["corr_a1000_d999_full_fft"] = dsp.compute_correlation(sig["a1000"], sig["d999"], method="fft")
    g["corr_a1000_d999_full_direct"] = dsp.compute_correlation(sig["a1000"], sig["d999"], method="direct")
    for a in ("a1000", "b50"):
        for mode in ("full", "same", "valid"):
            g[f"acorr_{a}_{mode}"] = dsp.compute_autocorrelation(sig[a], mode=mode)
    pcases = {"default": {}, "nfft1024": dict(nfft=1024), "nfft2048": dict(nfft=2048), "nfft512": dict(nfft=512),
              "nfft777": dict(nfft=777), "spectrum": dict(scaling="spectrum"), "nodetrend": dict(detrend=False),
              "boxcar": dict(window="boxcar"), "hamming": dict(window="hamming")}
    for a in ("a1000", "d999", "c4096"):
        for tag, kw in pcases.items():
            f, p = dsp.compute_psd_periodogram(sig[a], fs=1000.0, **kw)
            g[f"pgram_{a}_{tag}_f"], g[f"pgram_{a}_{tag}_p"] = f, p
    for a in ("a1000", "d999", "c4096", "b50", "e7", "one"):
        g[f"hilbert_{a}"] = tr.hilbert_transform(sig[a])
        g[f"envelope_{a}"] = dsp.amplitude_envelope(sig[a], method="hilbert")
    # linear detrending (appended last: no random draws, earlier entries unchanged)
    ramp = {k: sig[k] + np.linspace(-0.5, 1.5, len(sig[k])) for k in ("a1000", "d999", "c4096")}
    for a, x in ramp.items():
        g["ramp_" + a] = x
        f, p = dsp.compute_psd_periodogram(x, fs=1000.0, detrend="linear")
        g[f"pgram_{a}_linear_f"], g[f"pgram_{a}_linear_p"] = f, p
    for tag, kw in (("w256", dict(nperseg=256)), ("w1024o768", dict(nperseg=1024, noverlap=768)),
                    ("w512nfft1024", dict(nperseg=512, nfft=1024))):
        f, p = dsp.compute_psd_welch(ramp["c4096"], fs=48000.0, detrend="linear", **kw)
        g[f"welch_linear_{tag}_f"], g[f"welch_linear_{tag}_p"] = f, p
    # Welch with segment lengths that are not powers of two
    for tag, kw in (("n1000", dict(nperseg=1000)), ("n300nfft500", dict(nperseg=300, nfft=500)),
                    ("n777o100", dict(nperseg=777, noverlap=100)), ("n1000lin", dict(nperseg=1000, detrend="linear")),
                    ("n20000", dict(nperseg=20000))):                      # longer than the signal: scipy shortens it
        f, p = dsp.compute_psd_welch(ramp["c4096"], fs=48000.0, **kw)
        g[f"welch_any_{tag}_f"], g[f"welch_any_{tag}_p"] = f, p
    np.savez_compressed(os.path.join(OUT, "ref_dsp2.npz"), **g)
    # ---- ml_utils: scalers and formatters (scikit-learn / scipy.ndimage behind the reference's functions) ----
    REFML = "ml_utils"
    sc = load("ref_scaling", os.path.join(REFML, "scaling.py"))
    fm = load("ref_formatters", os.path.join(REFML, "formatters.py"))
    rng3 = np.random.default_rng(20250525)
    g = {}
    X = rng3.normal(0, 1, (300, 6)) * np.array([1.0, 10.0, 0.01, 5.0, 1.0, 3.0]) + np.array([0.0, 100.0, -3.0, 0.0, 7.0, 1e3])
    X[:, 4] = 7.0                                             # constant feature
    X = X.astype(np.float32).astype(np.float64)               # float32-representable: the device sees the same values
    g["X"] = X
    for tag, kind, kw in (("std", "standard", {}), ("std_nomean", "standard", {"with_mean": False}),
                          ("std_nostd", "standard", {"with_std": False}), ("mm", "minmax", {}),
                          ("mm_m11", "minmax", {"feature_range": (-1, 1)}), ("rob", "robust", {}),
                          ("rob_1090", "robust", {"quantile_range": (10.0, 90.0)}),
                          ("rob_nocenter", "robust", {"with_centering": False})):
        out, scaler = sc.apply_scaling(X, scaler_type=kind, scaler_params=kw)
        g[f"scale_{tag}"] = out
        for attr in ("mean_", "var_", "scale_", "min_", "center_"):
            if getattr(scaler, attr, None) is not None:
                g[f"scale_{tag}_{attr}"] = np.asarray(getattr(scaler, attr))
    out, scaler = sc.standard_scale(X[:, 1])                  # 1-D input is reshaped to one column
    g["scale_1d"] = out
    feats = {f"f{i}": X[:40, i] for i in range(4)}
    g["seq_list"] = fm.format_feature_sequences(feats)[0]
    g["seq_pad64"] = fm.format_feature_sequences(feats, max_sequence_length=64, padding_value=-1.0, output_format="padded_array")
    g["seq_cut16_post"] = fm.format_feature_sequences(feats, max_sequence_length=16, output_format="padded_array")
    g["seq_cut16_pre"] = fm.format_feature_sequences(feats, max_sequence_length=16, truncation_strategy="pre",
                                                     output_format="padded_array")
    segs = [(0, 15), (15, 30), (30, 40), (5, 6)]
    for agg in ("mean", "std", "median", "min", "max"):
        g[f"vec_{agg}"] = fm.format_feature_vectors_per_segment(feats, segs, aggregation=agg, output_format="numpy")
    g["vec_mixed"] = fm.format_feature_vectors_per_segment(feats, segs, aggregation={"f0": "max", "f1": "min", "f2": "std"},
                                                           output_format="numpy")
    M = np.abs(rng3.normal(0, 1, (40, 94))).astype(np.float32).astype(np.float64) * np.linspace(3, 0.1, 40)[:, None]
    g["img_in"] = M
    g["img_norm"] = fm.format_features_as_image(M)
    g["img_64x64"] = fm.format_features_as_image(M, output_shape=(64, 64))
    g["img_20x200_nonorm"] = fm.format_features_as_image(M, output_shape=(20, 200), normalize=False)
    g["img_128x32_nearest"] = fm.format_features_as_image(M, output_shape=(128, 32), resize_order=0)
    g["img_const"] = fm.format_features_as_image(np.full((5, 7), 2.5))
    np.savez_compressed(os.path.join(OUT, "ref_ml.npz"), **g)
    print("golden vectors written to", OUT)


if __name__ == "__main__":
    main()
