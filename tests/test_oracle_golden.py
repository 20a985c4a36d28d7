"""Pin the oracle against golden vectors produced by the reference's own code
(tests/golden/make_golden.py).  CPU only."""
import os

import numpy as np
import pytest
from numpy.testing import assert_allclose, assert_array_equal

from oracle import cpu_ref as O

G = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def gf():
    return np.load(os.path.join(G, "ref_filters.npz"))


@pytest.fixture(scope="module")
def gd():
    return np.load(os.path.join(G, "ref_dsp.npz"))


@pytest.fixture(scope="module")
def gq():
    return np.load(os.path.join(G, "ref_freq.npz"))


DESIGNS = [("bp4_48k", (300.0, 3400.0), 48000.0, 4, "bandpass"), ("lp5_1k", 100.0, 1000.0, 5, "lowpass"),
           ("hp5_1k", 100.0, 1000.0, 5, "highpass"), ("bs5_1k", (100.0, 200.0), 1000.0, 5, "bandstop"),
           ("lp8_1k", 100.0, 1000.0, 8, "lowpass"), ("bp2_16k", (500.0, 2000.0), 16000.0, 2, "bandpass")]


@pytest.mark.parametrize("name,cutoff,fs,order,kind", DESIGNS)
def test_butterworth_design_and_filtfilt(gf, name, cutoff, fs, order, kind):
    sos = O.design_butterworth_sos(cutoff, fs, order, kind)
    assert_array_equal(sos, gf[f"{name}_sos"])
    y = O.apply_sos_filter(sos, gf[f"{name}_x"])
    assert_array_equal(y, gf[f"{name}_y"])


@pytest.mark.parametrize("name", ["bp4_48k", "lp5_1k", "bs5_1k"])
def test_filtfilt_explicit_restatement(gf, name):
    """The explicit odd-extension / zi / fwd-bwd restatement reproduces the reference output."""
    y = O.sosfiltfilt_explicit(gf[f"{name}_sos"], gf[f"{name}_x"])
    ref = gf[f"{name}_y"]
    assert np.max(np.abs(y - ref)) <= 1e-11 * np.max(np.abs(ref))


def test_padlen_c3_filter(gf):
    assert O.sosfiltfilt_padlen(gf["bp4_48k_sos"]) == 27
    assert gf["bp4_48k_sos"].shape == (4, 6)


def test_convenience_filters(gf):
    x = gf["conv_x"]
    assert_array_equal(O.band_pass_filter(x, 300.0, 3400.0, 48000.0, order=4), gf["conv_bp"])
    assert_array_equal(O.low_pass_filter(x, 4000.0, 48000.0), gf["conv_lp"])
    assert_array_equal(O.high_pass_filter(x, 4000.0, 48000.0), gf["conv_hp"])
    assert_array_equal(O.band_stop_filter(x, 1000.0, 5000.0, 48000.0), gf["conv_bs"])


def test_filter_error_strings():
    # strings pinned by reference tests/test_filters.py:73-87
    with pytest.raises(ValueError, match="strictly between 0 and Nyquist"):
        O.design_butterworth_sos(500.0, 1000.0, 4, "lowpass")
    with pytest.raises(ValueError, match="Low cutoff .* must be less than high cutoff"):
        O.design_butterworth_sos((200.0, 100.0), 1000.0, 4, "bandpass")
    with pytest.raises(TypeError, match="cutoff must be a float .* or a tuple"):
        O.design_butterworth_sos([100.0], 1000.0, 4, "lowpass")


@pytest.mark.parametrize("win", ["hann", "hamming", "blackman", None])
@pytest.mark.parametrize("n", [None, 1024, 512, 1500])
def test_compute_fft(gd, win, n):
    f, s = O.compute_fft(gd["x1000"], fs=1000.0, n=n, window=win)
    assert_array_equal(f, gd[f"fft_{win}_{n}_f"])
    assert_array_equal(s, gd[f"fft_{win}_{n}_s"])
    assert s.dtype == np.complex128 and f.dtype == np.float64


def test_compute_ifft(gd):
    _, sp = O.compute_fft(gd["x1000"], fs=1000.0, window=None)
    assert_array_equal(O.compute_ifft(sp), gd["ifft_none"])
    assert_array_equal(O.compute_ifft(sp, n=768), gd["ifft_n768"])
    assert_array_equal(O.compute_ifft(sp, n=1200), gd["ifft_n1200"])
    assert_allclose(gd["ifft_none"], gd["x1000"], atol=1e-9, rtol=1e-7)  # reference tests/test_dsp.py:69-76


@pytest.mark.parametrize("w", ["hann", "hamming", "blackman", "bartlett", "boxcar"])
def test_apply_window(gd, w):
    assert_array_equal(O.apply_window(gd["x1000"], w), gd[f"win_{w}"])


def test_apply_window_bad_name():
    with pytest.raises(ValueError, match="Invalid window type 'nope'"):
        O.apply_window(np.ones(8), "nope")


WELCH = [("w4096", dict(nperseg=4096)), ("w256", dict(nperseg=256)), ("w1024o768", dict(nperseg=1024, noverlap=768)),
         ("w512nfft1024", dict(nperseg=512, nfft=1024)), ("w1024spec", dict(nperseg=1024, scaling="spectrum")),
         ("w1024nodet", dict(nperseg=1024, detrend=False)), ("w1024hamming", dict(nperseg=1024, window="hamming"))]


@pytest.mark.parametrize("tag,kw", WELCH)
def test_welch(gd, tag, kw):
    f, p = O.compute_psd_welch(gd["x20000"], fs=48000.0, **kw)
    assert_array_equal(f, gd[f"welch_{tag}_f"])
    assert_array_equal(p, gd[f"welch_{tag}_p"])
    f2, p2 = O.welch_explicit(gd["x20000"], fs=48000.0, **{"nperseg": 256, **kw})
    assert_allclose(f2, f, rtol=0, atol=1e-9)
    assert np.max(np.abs(p2 - p)) <= 1e-12 * np.max(np.abs(p))


def test_per_frame_spectral_functions(gq):
    S, fr = gq["spectra"], gq["freqs"]
    assert_array_equal(np.array([O.spectral_centroid(s, fr) for s in S]), gq["centroid"])
    assert_array_equal(np.array([O.spectral_bandwidth(s, fr) for s in S]), gq["bandwidth"])
    assert_array_equal(np.array([O.spectral_bandwidth(s, fr, p=1) for s in S]), gq["bandwidth_p1"])
    assert_array_equal(np.array([O.spectral_bandwidth(s, fr, centroid=5000.0) for s in S]), gq["bandwidth_c"])
    assert_array_equal(np.array([O.spectral_flatness(s) for s in S]), gq["flatness"])
    assert_array_equal(np.array([O.spectral_rolloff(s, fr) for s in S]), gq["rolloff85"])
    assert_array_equal(np.array([O.spectral_rolloff(s, fr, roll_percent=0.5) for s in S]), gq["rolloff50"])
    assert_array_equal(np.array([O.spectral_rolloff(s, fr, roll_percent=1.0) for s in S]), gq["rolloff100"])
    assert_array_equal(np.array([O.spectral_rolloff(s, fr, roll_percent=0.0) for s in S]), gq["rolloff0"])
    assert_array_equal(np.array([O.dominant_frequency(s, fr) for s in S]), gq["dominant"])


def test_per_frame_edge_cases(gq):
    e = np.array([], dtype=np.float64)
    got = np.array([O.spectral_centroid(e, e), O.spectral_bandwidth(e, e), O.spectral_flatness(e),
                    O.spectral_rolloff(e, e), O.dominant_frequency(e, e)])
    assert_array_equal(got, gq["empty"])
    # all-zero frame (reference tests/test_features_freq.py:95-101 etc.)
    z, fr = gq["spectra"][3], gq["freqs"]
    assert O.spectral_centroid(z, fr) == 0.0 and O.spectral_bandwidth(z, fr) == 0.0
    assert O.spectral_flatness(z) == 0.0 and O.spectral_rolloff(z, fr) == fr[-1]
    assert O.dominant_frequency(z, fr) == fr[0]


def test_vectorised_stats_match_per_frame(gq):
    S, fr = gq["spectra"], gq["freqs"]
    st = O.spectral_stats_frames(S.T, fr)
    assert_allclose(st["spectral_centroid"], gq["centroid"], rtol=1e-13, atol=1e-9)
    assert_allclose(st["spectral_bandwidth"], gq["bandwidth"], rtol=1e-12, atol=1e-9)
    assert_allclose(st["spectral_flatness"], gq["flatness"], rtol=1e-12, atol=1e-15)
    assert_array_equal(st["spectral_rolloff"], gq["rolloff85"])
    assert_array_equal(st["dominant_frequency"], gq["dominant"])


# ---------------------------------------------------------------- time-domain frame features
def _time_golden():
    return np.load(os.path.join(G, "ref_time.npz"))


@pytest.mark.parametrize("name", ["mean_amplitude", "std_dev_amplitude", "skewness", "kurtosis", "peak_amplitude",
                                  "crest_factor", "signal_entropy"])
def test_time_domain_functions_match_reference(name):
    g = _time_golden()
    out = np.array([O._TIME_FUNCS[name](f) for f in g["frames"]])
    np.testing.assert_allclose(out, g[name], rtol=1e-12, atol=1e-13)


def test_time_domain_entropy_bins_and_short_frames():
    g = _time_golden()
    for nb in (4, 32):
        out = np.array([O.signal_entropy(f, nb) for f in g["frames"]])
        np.testing.assert_allclose(out, g[f"signal_entropy_b{nb}"], rtol=1e-12, atol=1e-13)
    for i in range(4):
        f = g[f"short{i}"]
        out = np.array([O._TIME_FUNCS[nm](f) for nm in O.TIME_FEATURES])
        np.testing.assert_allclose(out, g[f"short{i}_out"], rtol=1e-12, atol=1e-13)


def test_time_domain_restatement_agrees_with_scipy_stats():
    """The explicit moment / histogram formulas against the SciPy / NumPy routines the reference calls."""
    import scipy.stats
    rng = np.random.default_rng(5)
    for n in (5, 64, 2048):
        f = rng.gamma(2.0, 0.1, n) - 0.1
        assert abs(O.skewness(f) - scipy.stats.skew(f, bias=False)) < 1e-12
        assert abs(O.kurtosis_val(f) - scipy.stats.kurtosis(f, fisher=True, bias=False)) < 1e-11
        c, _ = np.histogram(f, bins=10)
        assert np.array_equal(O.histogram_counts(f, 10), c)
        assert abs(O.signal_entropy(f) - scipy.stats.entropy(c[c > 0] / n)) < 1e-13


# ---------------------------------------------------------------- f-3: FFT-backed 1-D operations (ref_dsp2.npz)
CONV_PAIRS = [("a1000", "k31"), ("b50", "k200"), ("c4096", "k512"), ("e7", "k2"), ("a1000", "one"), ("one", "k2")]
CORR_PAIRS = [("a1000", "d999"), ("b50", "k200"), ("k200", "b50"), ("e7", "k2"), ("c4096", "k512")]
PGRAM_CASES = {"default": {}, "nfft1024": dict(nfft=1024), "nfft2048": dict(nfft=2048), "nfft512": dict(nfft=512),
               "nfft777": dict(nfft=777), "spectrum": dict(scaling="spectrum"), "nodetrend": dict(detrend=False),
               "boxcar": dict(window="boxcar"), "hamming": dict(window="hamming")}


@pytest.fixture(scope="module")
def g2():
    return np.load(os.path.join(G, "ref_dsp2.npz"))


def _close(got, want, rel=1e-10):
    assert got.shape == want.shape
    scale = max(float(np.max(np.abs(want))), 1e-300) if want.size else 1.0
    assert float(np.max(np.abs(got - want))) <= rel * scale if want.size else True


@pytest.mark.parametrize("mode", ["full", "same", "valid"])
def test_convolution_matches_reference(g2, mode):
    for a, b in CONV_PAIRS:
        _close(O.apply_convolution(g2["sig_" + a], g2["sig_" + b], mode), g2[f"conv_{a}_{b}_{mode}"])


def test_convolution_readme_example(g2):
    # the inputs of the docstring example dsp.py:318-322; the values are what the reference RETURNS for them (the
    # docstring prints the result shifted by one sample -- scipy centres 'same' at (m - 1) // 2 = 0 for m = 2)
    np.testing.assert_allclose(g2["conv_e7_k2_same"], [0, 0, 1, 0, 0, -1, 0], atol=1e-12)
    np.testing.assert_allclose(O.apply_convolution(g2["sig_e7"], g2["sig_k2"], "same"), [0, 0, 1, 0, 0, -1, 0],
                               atol=1e-12)


@pytest.mark.parametrize("mode", ["full", "same", "valid"])
def test_correlation_matches_reference(g2, mode):
    for a, b in CORR_PAIRS:
        _close(O.compute_correlation(g2["sig_" + a], g2["sig_" + b], mode), g2[f"corr_{a}_{b}_{mode}"])
    for a in ("a1000", "b50"):
        _close(O.compute_autocorrelation(g2["sig_" + a], mode), g2[f"acorr_{a}_{mode}"])


def test_correlation_methods_agree(g2):
    want = O.compute_correlation(g2["sig_a1000"], g2["sig_d999"])
    _close(want, g2["corr_a1000_d999_full_fft"])
    _close(want, g2["corr_a1000_d999_full_direct"])


@pytest.mark.parametrize("tag", sorted(PGRAM_CASES))
def test_periodogram_matches_reference(g2, tag):
    for a in ("a1000", "d999", "c4096"):
        f, p = O.compute_psd_periodogram(g2["sig_" + a], fs=1000.0, **PGRAM_CASES[tag])
        np.testing.assert_allclose(f, g2[f"pgram_{a}_{tag}_f"], rtol=0, atol=1e-9)
        _close(p, g2[f"pgram_{a}_{tag}_p"])


def test_hilbert_and_envelope_match_reference(g2):
    for a in ("a1000", "d999", "c4096", "b50", "e7", "one"):
        _close(O.hilbert_transform(g2["sig_" + a]), g2[f"hilbert_{a}"])
        _close(O.amplitude_envelope(g2["sig_" + a]), g2[f"envelope_{a}"])


def test_f3_error_strings():
    with pytest.raises(ValueError, match="must be 1D"):
        O.apply_convolution(np.zeros((2, 2)), np.zeros(2))
    with pytest.raises(ValueError, match="must be 1D"):
        O.compute_correlation(np.zeros((2, 2)), np.zeros(2))
    with pytest.raises(ValueError, match="must be a 1D"):
        O.compute_psd_periodogram(np.zeros((2, 2)))
    with pytest.raises(ValueError, match="Unsupported envelope method"):
        O.amplitude_envelope(np.zeros(8), method="peak")
    with pytest.raises(ValueError, match="required for 'rms'"):
        O.amplitude_envelope(np.zeros(8), method="rms")


def test_linear_detrend_matches_reference(g2):
    for a in ("a1000", "d999", "c4096"):
        f, p = O.compute_psd_periodogram(g2["ramp_" + a], fs=1000.0, detrend="linear")
        np.testing.assert_allclose(f, g2[f"pgram_{a}_linear_f"], rtol=0, atol=1e-9)
        _close(p, g2[f"pgram_{a}_linear_p"], 1e-9)
    for tag, kw in (("w256", dict(nperseg=256)), ("w1024o768", dict(nperseg=1024, noverlap=768)),
                    ("w512nfft1024", dict(nperseg=512, nfft=1024))):
        f, p = O.welch_explicit(g2["ramp_c4096"], fs=48000.0, detrend="linear", **kw)
        np.testing.assert_allclose(f, g2[f"welch_linear_{tag}_f"], rtol=0, atol=1e-6)
        _close(p, g2[f"welch_linear_{tag}_p"], 1e-9)


WELCH_ANY = [("n1000", dict(nperseg=1000)), ("n300nfft500", dict(nperseg=300, nfft=500)),
             ("n777o100", dict(nperseg=777, noverlap=100)), ("n1000lin", dict(nperseg=1000, detrend="linear")),
             ("n20000", dict(nperseg=20000))]


@pytest.mark.parametrize("tag,kw", WELCH_ANY)
def test_welch_any_segment_length_matches_reference(g2, tag, kw):
    f, p = O.welch_explicit(g2["ramp_c4096"], fs=48000.0, **kw)
    np.testing.assert_allclose(f, g2[f"welch_any_{tag}_f"], rtol=0, atol=1e-6)
    _close(p, g2[f"welch_any_{tag}_p"], 1e-9)


# ---------------------------------------------------------------- f-4: scalers and formatters (ref_ml.npz)
SCALER_CASES = [("std", "standard", {}), ("std_nomean", "standard", {"with_mean": False}),
                ("std_nostd", "standard", {"with_std": False}), ("mm", "minmax", {}),
                ("mm_m11", "minmax", {"feature_range": (-1, 1)}), ("rob", "robust", {}),
                ("rob_1090", "robust", {"quantile_range": (10.0, 90.0)}), ("rob_nocenter", "robust", {"with_centering": False})]


@pytest.fixture(scope="module")
def gm():
    return np.load(os.path.join(G, "ref_ml.npz"))


@pytest.mark.parametrize("tag,kind,kw", SCALER_CASES)
def test_scalers_match_reference(gm, tag, kind, kw):
    out, attrs = O.apply_scaling(gm["X"], kind, kw)
    assert_allclose(out, gm[f"scale_{tag}"], rtol=1e-12, atol=1e-12)
    for a in ("mean_", "var_", "scale_", "min_", "center_"):
        if f"scale_{tag}_{a}" in gm.files:
            assert_allclose(attrs[a], gm[f"scale_{tag}_{a}"], rtol=1e-12, atol=1e-12)


def test_formatters_match_reference(gm):
    feats = {f"f{i}": gm["X"][:40, i] for i in range(4)}
    assert_array_equal(O.format_feature_sequences(feats)[0], gm["seq_list"])
    assert_array_equal(O.format_feature_sequences(feats, 64, -1.0, output_format="padded_array"), gm["seq_pad64"])
    assert_array_equal(O.format_feature_sequences(feats, 16, output_format="padded_array"), gm["seq_cut16_post"])
    assert_array_equal(O.format_feature_sequences(feats, 16, truncation_strategy="pre", output_format="padded_array"),
                       gm["seq_cut16_pre"])
    M = gm["img_in"]
    assert_allclose(O.format_features_as_image(M), gm["img_norm"], rtol=0, atol=1e-15)
    assert_allclose(O.format_features_as_image(M, (64, 64)), gm["img_64x64"], rtol=0, atol=1e-15)
    assert_allclose(O.format_features_as_image(M, (20, 200), normalize=False), gm["img_20x200_nonorm"], rtol=0, atol=1e-15)
    assert_allclose(O.format_features_as_image(M, (128, 32), resize_order=0), gm["img_128x32_nearest"], rtol=0, atol=1e-15)
    assert_array_equal(O.format_features_as_image(np.full((5, 7), 2.5)), gm["img_const"])


def test_segment_vectors_match_reference(gm):
    feats = {f"f{i}": gm["X"][:40, i] for i in range(4)}
    segs = [(0, 15), (15, 30), (30, 40), (5, 6)]
    for agg in ("mean", "std", "median", "min", "max"):
        assert_allclose(O.format_feature_vectors_per_segment(feats, segs, agg), gm[f"vec_{agg}"], rtol=1e-13, atol=0)
    assert_allclose(O.format_feature_vectors_per_segment(feats, segs, {"f0": "max", "f1": "min", "f2": "std"}),
                    gm["vec_mixed"], rtol=1e-13, atol=0)
    nanrow = O.format_feature_vectors_per_segment(feats, [(0, 10), (10, 60)])
    assert np.isnan(nanrow[1]).all() and not np.isnan(nanrow[0]).any()
