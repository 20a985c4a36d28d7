"""GPU parity of the FFT-backed 1-D operations: convolution, cross/auto-correlation, periodogram,
analytic signal and Hilbert envelope -- sygnals_amd.core.dsp / core.transforms -> ops -> libsygnals_hip.so
(syg_pack_rows_f32, syg_rconv_spectrum_c64, syg_analytic_mask_c64, syg_psd_onesided_f32 + the FFT kernels).

Pinned on tests/golden/ref_dsp2.npz (outputs of the reference's own functions), then against the float64 oracle on
larger and batched inputs, then through size-independent properties at sizes the direct-sum oracle cannot reach.
Tolerance: 1e-5 of the output's peak (fp32 arithmetic on the device, float64 in the reference)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")
from oracle import cpu_ref as O
from tests.gpu_util import assert_parity, peak_rel
from tests.test_oracle_golden import CONV_PAIRS, CORR_PAIRS, PGRAM_CASES, WELCH_ANY

TOL = 1e-5
G = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def g2():
    return np.load(os.path.join(G, "ref_dsp2.npz"))


@pytest.fixture(scope="module", autouse=True)
def _gpu():
    from sygnals_amd import ops
    ops.require_gpu()


@pytest.mark.parametrize("mode", ["full", "same", "valid"])
def test_convolution_vs_reference_golden(g2, mode):
    from sygnals_amd.core.dsp import apply_convolution
    for a, b in CONV_PAIRS:
        got = apply_convolution(g2["sig_" + a], g2["sig_" + b], mode=mode)
        assert got.dtype == np.float64
        assert_parity(got, g2[f"conv_{a}_{b}_{mode}"], TOL, f"conv {a}*{b} {mode}")


@pytest.mark.parametrize("mode", ["full", "same", "valid"])
def test_correlation_vs_reference_golden(g2, mode):
    from sygnals_amd.core.dsp import compute_autocorrelation, compute_correlation
    for a, b in CORR_PAIRS:
        got = compute_correlation(g2["sig_" + a], g2["sig_" + b], mode=mode)
        assert_parity(got, g2[f"corr_{a}_{b}_{mode}"], TOL, f"corr {a},{b} {mode}")
    for a in ("a1000", "b50"):
        assert_parity(compute_autocorrelation(g2["sig_" + a], mode=mode), g2[f"acorr_{a}_{mode}"], TOL, "acorr")
    for method in ("fft", "direct"):
        got = compute_correlation(g2["sig_a1000"], g2["sig_d999"], method=method)
        assert_parity(got, g2[f"corr_a1000_d999_full_{method}"], TOL, method)


def test_autocorrelation_docstring_properties():
    # dsp.py:414-427: peak at zero lag, secondary peak one period (10 samples) later
    from sygnals_amd.core.dsp import compute_autocorrelation
    fs = 100
    x = np.sin(2 * np.pi * 10 * np.arange(fs * 2) / fs)
    ac = compute_autocorrelation(x, mode="full")
    zero = len(x) - 1
    assert np.argmax(ac) == zero
    assert ac[zero + 10] > 0.8 * ac[zero]
    # dsp.py:378-383: y is x delayed by one sample -> peak at lag +1
    from sygnals_amd.core.dsp import compute_correlation
    a = np.array([0, 1, 2, 1, 0], dtype=float)
    b = np.array([0, 0, 1, 2, 1], dtype=float)
    c = compute_correlation(a, b, mode="full")
    assert_parity(c, O.compute_correlation(a, b), TOL, "corr small")


@pytest.mark.parametrize("tag", sorted(PGRAM_CASES))
def test_periodogram_vs_reference_golden(g2, tag):
    from sygnals_amd.core.dsp import compute_psd_periodogram
    for a in ("a1000", "d999", "c4096"):
        f, p = compute_psd_periodogram(g2["sig_" + a], fs=1000.0, **PGRAM_CASES[tag])
        assert f.dtype == np.float64 and p.dtype == np.float64
        np.testing.assert_allclose(f, g2[f"pgram_{a}_{tag}_f"], rtol=0, atol=1e-9)
        assert_parity(p, g2[f"pgram_{a}_{tag}_p"], TOL, f"periodogram {a} {tag}")


def test_periodogram_peak_and_window_array(g2):
    # dsp.py:470-476: 100 Hz sine at fs = 1000 peaks at 100 Hz
    from sygnals_amd.core.dsp import compute_psd_periodogram
    fs = 1000
    x = np.sin(2 * np.pi * 100 * np.arange(fs) / fs)
    f, p = compute_psd_periodogram(x, fs=fs)
    assert abs(f[np.argmax(p)] - 100.0) < 1e-9
    w = np.kaiser(1000, 5.0)
    f, p = compute_psd_periodogram(g2["sig_a1000"], fs=1000.0, window=w)
    assert_parity(p, O.compute_psd_periodogram(g2["sig_a1000"], fs=1000.0, window=w)[1], TOL, "array window")
    with pytest.raises(ValueError, match="length of nperseg"):
        compute_psd_periodogram(g2["sig_a1000"], window=np.ones(10))


def test_hilbert_and_envelope_vs_reference_golden(g2):
    from sygnals_amd.core.dsp import amplitude_envelope
    from sygnals_amd.core.transforms import hilbert_transform
    for a in ("a1000", "d999", "c4096", "b50", "e7", "one"):
        h = hilbert_transform(g2["sig_" + a])
        assert h.dtype == np.complex128
        assert peak_rel(h, g2[f"hilbert_{a}"]) <= TOL, a
        assert_parity(amplitude_envelope(g2["sig_" + a]), g2[f"envelope_{a}"], TOL, f"envelope {a}")


def test_envelope_rms_method_and_errors():
    from sygnals_amd.core.dsp import amplitude_envelope
    rng = np.random.default_rng(3)
    y = rng.normal(0, 0.3, 8000)
    got = amplitude_envelope(y, method="rms", frame_length=128, hop_length=64)
    assert_parity(got, O.amplitude_envelope(y, "rms", 128, 64), TOL, "rms envelope")
    with pytest.raises(ValueError, match="required for 'rms'"):
        amplitude_envelope(y, method="rms")
    with pytest.raises(ValueError, match="Unsupported envelope method"):
        amplitude_envelope(y, method="peak")
    with pytest.raises(ValueError, match="must be a 1D"):
        amplitude_envelope(np.zeros((2, 8)))


def test_f3_argument_errors():
    from sygnals_amd.core.dsp import apply_convolution, compute_correlation, compute_psd_periodogram
    from sygnals_amd.core.transforms import hilbert_transform
    with pytest.raises(ValueError, match="must be 1D"):
        apply_convolution(np.zeros((2, 2)), np.zeros(2))
    with pytest.raises(ValueError, match="must be 1D"):
        compute_correlation(np.zeros((2, 2)), np.zeros(2))
    with pytest.raises(ValueError, match="mode"):
        apply_convolution(np.zeros(4), np.zeros(2), mode="circular")
    with pytest.raises(ValueError, match="method"):
        compute_correlation(np.zeros(4), np.zeros(2), method="magic")
    with pytest.raises(ValueError, match="must be a 1D"):
        compute_psd_periodogram(np.zeros((2, 2)))
    with pytest.raises(ValueError, match="must be a 1D"):
        hilbert_transform(np.zeros((2, 2)))
    assert apply_convolution(np.zeros(0), np.zeros(3)).shape == (0,)


# ---------------------------------------------------------------- batched forms against the oracle
@pytest.mark.parametrize("n,m,shared", [(3000, 257, True), (3000, 257, False), (48000, 1023, True), (100, 4000, False),
                                       (17, 1, True), (1, 1, True), (65536, 2, True)])
def test_convolve_batch_vs_oracle(n, m, shared):
    from sygnals_amd import ops
    from sygnals_amd.core.dsp import convolve_batch
    rng = np.random.default_rng(n * 7 + m)
    B = 5
    X = rng.normal(0, 0.4, (B, n)).astype(np.float32)
    K = (rng.normal(0, 1.0, (1 if shared else B, m)) / np.sqrt(m)).astype(np.float32)
    X[1] *= 1e-3                                             # rows of very different scale stay independent
    xd, kd = ops.to_device_f32(X), ops.to_device_f32(K)
    for mode in ("full", "same", "valid"):
        for corr in (False, True):
            got = convolve_batch(xd, kd[0] if shared else kd, mode, correlate=corr).cpu().numpy()
            for b in range(B):
                kb = K[0 if shared else b].astype(np.float64)
                want = (O.compute_correlation if corr else O.apply_convolution)(X[b].astype(np.float64), kb, mode)
                assert_parity(got[b], want, TOL, f"n={n} m={m} {mode} corr={corr} row {b}")


def test_small_kernel_next_to_large_signal_keeps_its_precision():
    """The kernel is 1e-4 of the signal's scale; the error stays relative to the product's peak (each sequence has
    its own transform -- see dsp_extra.hip header)."""
    from sygnals_amd.core.dsp import apply_convolution
    rng = np.random.default_rng(11)
    x = rng.normal(0, 50.0, 5000)
    k = rng.normal(0, 1e-3, 301)
    assert_parity(apply_convolution(x, k, "full"), O.apply_convolution(x, k, "full"), TOL, "scale split")


def test_analytic_and_periodogram_batch_vs_oracle():
    from sygnals_amd import ops
    from sygnals_amd.core.dsp import analytic_batch, periodogram_batch
    rng = np.random.default_rng(5)
    for n in (1, 2, 3, 64, 777, 2048, 12000):
        X = (rng.normal(0, 0.5, (4, n)) + 0.3).astype(np.float32)
        a = analytic_batch(ops.to_device_f32(X)).cpu().numpy().astype(np.float64)
        for b in range(4):
            want = O.hilbert_transform(X[b].astype(np.float64))
            assert peak_rel(a[b, :, 0] + 1j * a[b, :, 1], want) <= TOL, n
        if n >= 2:
            f, p = periodogram_batch(ops.to_device_f32(X), fs=8000.0)
            for b in range(4):
                wf, wp = O.compute_psd_periodogram(X[b].astype(np.float64), fs=8000.0)
                assert_parity(p[b].cpu().numpy(), wp, TOL, f"periodogram n={n}")
                np.testing.assert_allclose(f, wf, rtol=0, atol=1e-9)


@pytest.mark.parametrize("n", [2, 16, 250, 4096, 6000, 8192, 16384, 48000, 65536, 44100])
def test_fused_analytic_signal_and_envelope(n):
    """syg_fft_*_strided_ex_f32: real input, analytic weights at the inverse's load, |.| at its store -- the analytic signal
    and the envelope of the three-pass chain (pack -> mask -> |.|), against the oracle and against the unfused chain; odd
    and even lengths, one-launch and four-step sizes, powers of two and 7-smooth lengths."""
    from sygnals_amd import ops
    from sygnals_amd.core.dsp import envelope_batch
    rng = np.random.default_rng(n)
    X = (rng.normal(0, 0.5, (5, n)) + 0.2).astype(np.float32)
    xd = ops.to_device_f32(X)
    a = ops.analytic_fused(xd, False)
    e = ops.analytic_fused(xd, True)
    assert a is not None and e is not None and a.shape == (5, n, 2) and e.shape == (5, n)
    chain = ops.fft_any(ops.pack_rows(xd, n, cplx=True))
    from sygnals_amd._lib import lib, check
    import ctypes as C
    check(lib().syg_analytic_mask_c64(ops._ptr(chain), 5, n, C.c_void_p(ops._stream_ptr())), "mask")
    chain = ops.fft_any(chain, True).cpu().numpy().astype(np.float64)
    a = a.cpu().numpy().astype(np.float64); e = e.cpu().numpy().astype(np.float64)
    for b in range(5):
        want = O.hilbert_transform(X[b].astype(np.float64))
        assert peak_rel(a[b, :, 0] + 1j * a[b, :, 1], want) <= TOL, n
        assert peak_rel(e[b], np.abs(want)) <= TOL, n
        assert peak_rel(a[b, :, 0] + 1j * a[b, :, 1], chain[b, :, 0] + 1j * chain[b, :, 1]) <= 2e-6, n
    assert np.array_equal(envelope_batch(xd).cpu().numpy(), e.astype(np.float32))


# ---------------------------------------------------------------- properties at sizes beyond the direct-sum oracle
def test_long_convolution_properties():
    """2^20-sample rows against a 4097-tap kernel (four-step FFT of 2^20 complex points):
    delta kernel = delayed copy, linearity in the signal, and the sum rule sum(x * k) = sum(x) * sum(k)."""
    from sygnals_amd import ops
    from sygnals_amd.core.dsp import convolve_batch
    rng = np.random.default_rng(21)
    n, m = 1 << 20, 4097
    X = rng.normal(0, 0.3, (3, n)).astype(np.float32)
    X[2] = X[0] + 2.0 * X[1]
    xd = ops.to_device_f32(X)
    delta = np.zeros(m, dtype=np.float32); delta[100] = 1.0
    full = convolve_batch(xd, ops.to_device_f32(delta), "full").cpu().numpy()
    assert full.shape == (3, n + m - 1)
    assert np.max(np.abs(full[:, 100:100 + n] - X)) <= TOL * np.max(np.abs(X))
    assert np.max(np.abs(full[:, :100])) <= TOL and np.max(np.abs(full[:, 100 + n:])) <= TOL
    k = (rng.normal(0, 1.0, m) / np.sqrt(m)).astype(np.float32)
    y = convolve_batch(xd, ops.to_device_f32(k), "full").cpu().numpy().astype(np.float64)
    peak = np.max(np.abs(y))
    assert np.max(np.abs(y[2] - (y[0] + 2.0 * y[1]))) <= 4 * TOL * peak
    for b in range(3):
        want = X[b].astype(np.float64).sum() * k.astype(np.float64).sum()
        assert abs(y[b].sum() - want) <= 1e-3 * max(abs(want), np.abs(y[b]).sum() * 1e-3)
    # a window of the long result against the direct sum
    seg = slice(500000, 500512)
    want = np.array([np.dot(X[0, i - m + 1:i + 1][::-1].astype(np.float64), k.astype(np.float64))
                     for i in range(seg.start, seg.stop)])
    assert np.max(np.abs(y[0, seg] - want)) <= TOL * peak


def test_long_hilbert_and_periodogram_properties():
    """n = 2^22 + 17 (Bluestein over the four-step FFT): Re(analytic) = x, a pure tone's envelope is its amplitude,
    and Parseval: sum(Pxx) * df = mean(x^2) for the boxcar window without detrending."""
    from sygnals_amd import ops
    from sygnals_amd.core.dsp import analytic_batch, periodogram_batch
    n = (1 << 22) + 17
    t = np.arange(n, dtype=np.float64)
    tone = 0.7 * np.cos(2 * np.pi * (40000.0 / n) * t)         # integer number of cycles
    rng = np.random.default_rng(8)
    X = np.stack([tone, rng.normal(0, 0.3, n)]).astype(np.float32)
    a = analytic_batch(ops.to_device_f32(X)).cpu().numpy()
    assert np.max(np.abs(a[:, :, 0] - X)) <= 2 * TOL * np.max(np.abs(X))
    env = np.hypot(a[0, :, 0], a[0, :, 1])
    assert np.max(np.abs(env - 0.7)) <= 1e-4
    f, p = periodogram_batch(ops.to_device_f32(X), fs=1.0, window="boxcar", detrend=False)
    p = p.cpu().numpy().astype(np.float64)
    for b in range(2):
        power = np.mean(X[b].astype(np.float64) ** 2)
        assert abs(p[b].sum() * (1.0 / n) - power) <= 1e-5 * power


def test_cli_dsp_commands(tmp_path):
    """`sygnals dsp convolution / correlation / psd-periodogram / hilbert` (README.md:748-815, 884-899) end to end."""
    import pandas as pd
    from click.testing import CliRunner
    from sygnals_amd.cli.main import cli
    rng = np.random.default_rng(2)
    x = rng.normal(0, 0.3, 3000); k = np.hanning(33) / np.hanning(33).sum()
    pd.DataFrame({"value": x}).to_csv(tmp_path / "x.csv", index=False)
    np.savez(tmp_path / "k.npz", data=k)
    run = lambda *a: CliRunner().invoke(cli, ["dsp", *map(str, a)])
    r = run("convolution", tmp_path / "x.csv", tmp_path / "k.npz", "-o", tmp_path / "c.npz", "--mode", "same")
    assert r.exit_code == 0, r.output
    assert_parity(np.load(tmp_path / "c.npz")["data"], O.apply_convolution(x, k, "same"), TOL, "cli conv")
    r = run("correlation", tmp_path / "x.csv", "-o", tmp_path / "a.npz")
    assert r.exit_code == 0, r.output
    assert_parity(np.load(tmp_path / "a.npz")["data"], O.compute_autocorrelation(x), TOL, "cli acorr")
    r = run("correlation", tmp_path / "x.csv", tmp_path / "k.npz", "-o", tmp_path / "cc.npz", "--mode", "valid")
    assert r.exit_code == 0, r.output
    assert_parity(np.load(tmp_path / "cc.npz")["data"], O.compute_correlation(x, k, "valid"), TOL, "cli corr")
    r = run("psd-periodogram", tmp_path / "x.csv", "--fs", 1000, "-o", tmp_path / "p.csv", "--scaling", "spectrum")
    assert r.exit_code == 0, r.output
    df = pd.read_csv(tmp_path / "p.csv")
    wf, wp = O.compute_psd_periodogram(x, fs=1000.0, scaling="spectrum")
    assert_parity(df["PSD"].to_numpy(), wp, TOL, "cli periodogram")
    np.testing.assert_allclose(df["Frequency"].to_numpy(), wf, atol=1e-9)
    r = run("psd-periodogram", tmp_path / "x.csv", "--fs", 1000, "-o", tmp_path / "p2.csv", "--detrend", "linear")
    assert r.exit_code == 0, r.output
    assert_parity(pd.read_csv(tmp_path / "p2.csv")["PSD"].to_numpy(),
                  O.compute_psd_periodogram(x, fs=1000.0, detrend="linear")[1], TOL, "cli periodogram linear")
    r = run("hilbert", tmp_path / "x.csv", "-o", tmp_path / "h.npz")
    assert r.exit_code == 0, r.output
    z = np.load(tmp_path / "h.npz")
    assert peak_rel(z["analytic_signal"], O.hilbert_transform(x)) <= TOL
    assert_parity(z["envelope"], O.amplitude_envelope(x), TOL, "cli envelope")


def test_linear_detrend_vs_reference_golden(g2):
    """detrend='linear' (scipy.signal.detrend: least-squares line per row / per Welch segment) on ramped signals."""
    from sygnals_amd.core.dsp import compute_psd_periodogram, compute_psd_welch
    for a in ("a1000", "d999", "c4096"):
        f, p = compute_psd_periodogram(g2["ramp_" + a], fs=1000.0, detrend="linear")
        assert_parity(p, g2[f"pgram_{a}_linear_p"], TOL, f"periodogram linear {a}")
    for tag, kw in (("w256", dict(nperseg=256)), ("w1024o768", dict(nperseg=1024, noverlap=768)),
                    ("w512nfft1024", dict(nperseg=512, nfft=1024))):
        f, p = compute_psd_welch(g2["ramp_c4096"], fs=48000.0, detrend="linear", **kw)
        np.testing.assert_allclose(f, g2[f"welch_linear_{tag}_f"], rtol=0, atol=1e-6)
        assert_parity(p, g2[f"welch_linear_{tag}_p"], TOL, f"welch linear {tag}")
    with pytest.raises(ValueError, match="Trend type"):
        compute_psd_welch(g2["ramp_c4096"], detrend="quadratic")


def test_linear_detrend_large_offset_and_slope():
    """A steep ramp on a large offset under small noise: the line has to be removed to ~1e-7 of its size before the
    spectrum of the noise is visible at all."""
    from sygnals_amd.core.dsp import compute_psd_periodogram, compute_psd_welch
    rng = np.random.default_rng(4)
    n = 20000
    x = (100.0 + 0.01 * np.arange(n) + rng.normal(0, 0.05, n)).astype(np.float32).astype(np.float64)
    f, p = compute_psd_welch(x, fs=1.0, nperseg=1024, detrend="linear")
    wf, wp = O.welch_explicit(x, fs=1.0, nperseg=1024, detrend="linear")
    assert_parity(p[1:], wp[1:], 1e-4, "welch linear, steep ramp")      # stated: cancellation of a 300:0.05 ramp in fp32
    f, p = compute_psd_periodogram(x, fs=1.0, detrend="linear")
    wf, wp = O.compute_psd_periodogram(x, fs=1.0, detrend="linear")
    assert_parity(p[1:], wp[1:], 1e-4, "periodogram linear, steep ramp")


@pytest.mark.parametrize("tag,kw", WELCH_ANY)
def test_welch_any_segment_length_vs_reference_golden(g2, tag, kw):
    """nperseg / nfft that are not powers of two (scipy takes any): overlapping-row path, arbitrary-length FFT."""
    from sygnals_amd.core.dsp import compute_psd_welch
    f, p = compute_psd_welch(g2["ramp_c4096"], fs=48000.0, **kw)
    np.testing.assert_allclose(f, g2[f"welch_any_{tag}_f"], rtol=0, atol=1e-6)
    assert_parity(p, g2[f"welch_any_{tag}_p"], TOL, f"welch {tag}")


def test_welch_batch_any_length_many_segments():
    """More segments than one launch takes rows (65535): the float64 average runs chunk after chunk."""
    from sygnals_amd import ops
    from sygnals_amd.core.dsp import welch_batch
    rng = np.random.default_rng(12)
    x = (rng.normal(0, 0.3, (2, 70000 * 6 + 12)) + 0.1).astype(np.float32)
    f, p = welch_batch(ops.to_device_f32(x), fs=100.0, nperseg=12, noverlap=6)          # 70001 segments
    for b in range(2):
        wf, wp = O.welch_explicit(x[b].astype(np.float64), fs=100.0, nperseg=12, noverlap=6)
        assert_parity(p[b].cpu().numpy(), wp, TOL, "welch nperseg=12")


@pytest.mark.parametrize("n_fft,hop,win_length,center", [(1000, 250, None, True), (600, 150, 400, True), (1023, 100, None, False),
                                                         (3, 1, None, True), (20000, 5000, None, True)])
def test_stft_any_frame_length_vs_oracle(n_fft, hop, win_length, center):
    """compute_stft with frame lengths that are not powers of two (librosa.stft takes any n_fft)."""
    from sygnals_amd.core.dsp import compute_stft
    y = O.synth_clips(1, 30000, 16000, seed=7)[0].astype(np.float64)
    X = compute_stft(y, n_fft=n_fft, hop_length=hop, win_length=win_length, center=center)
    want = O.stft(y, n_fft=n_fft, hop_length=hop, win_length=win_length, center=center)
    assert X.shape == want.shape and X.dtype == np.complex128
    assert peak_rel(X, want) <= TOL


def test_extract_features_with_odd_frame_length():
    """`features extract --frame-length 1000`: the manager falls to the generic STFT for any frame length."""
    from sygnals_amd.core.features.manager import extract_features
    sr = 16000
    y = O.synth_clips(1, 24000, sr, seed=9)[0].astype(np.float64)
    feats = ["mfcc", "spectral_centroid", "spectral_bandwidth", "rms_energy"]
    got = extract_features(y, sr, feats, frame_length=1000, hop_length=250, output_format="dict_of_arrays")
    want = O.extract_features(y, sr, feats, frame_length=1000, hop_length=250)
    for k in ("mfcc_0", "mfcc_5", "mfcc_12", "spectral_centroid", "spectral_bandwidth", "rms_energy"):
        assert_parity(got[k], want[k], TOL, k)
    # odd frame_length, hop dividing len(y): the centred STFT is one frame short of the manager's frame-count rule;
    # the reference then re-makes `time` from the STFT's own frame count (manager.py:186-194) -- no NaN padding
    got = extract_features(y, sr, ["spectral_centroid", "mfcc"], frame_length=999, hop_length=250, output_format="dict_of_arrays")
    want = O.extract_features(y, sr, ["spectral_centroid", "mfcc"], frame_length=999, hop_length=250)
    assert len(got["time"]) == len(want["time"]) == 24000 // 250 and np.array_equal(got["time"], want["time"])
    assert list(got) == list(want)
    assert_parity(got["spectral_centroid"], want["spectral_centroid"], TOL, "odd centroid")
    assert_parity(got["mfcc_3"], want["mfcc_3"], TOL, "odd mfcc")


@pytest.mark.parametrize("n", [2, 3, 5, 6, 7, 10, 12, 15, 21, 30, 35, 49, 100, 105, 240, 375, 441, 1000, 1500, 2401, 3000,
                               4410, 5000, 7000, 7680, 8000])
def test_mixed_radix_fft_vs_numpy(n):
    """Lengths 2^a 3^b 5^c 7^d up to 8192: one mixed-radix launch (syg_fft_mixed_strided_c2c_f32), forward and inverse."""
    from sygnals_amd import ops
    rng = np.random.default_rng(n)
    z = (rng.normal(0, 1, (3, n)) + 1j * rng.normal(0, 1, (3, n))).astype(np.complex64)
    x = torch.from_numpy(np.stack([z.real, z.imag], axis=-1)).cuda()
    assert ops.smooth_split(n) == (n, 1)
    for inverse, ref in ((False, np.fft.fft(z.astype(np.complex128), axis=1)), (True, np.fft.ifft(z.astype(np.complex128), axis=1))):
        got = ops.fft_any(x, inverse).cpu().numpy().astype(np.float64)
        assert peak_rel(got[..., 0] + 1j * got[..., 1], ref) <= TOL, (n, inverse)


@pytest.mark.parametrize("n", [16000, 22050, 44100, 48000, 96000, 1000000, 8192 * 8192 // 64 * 3])
def test_long_smooth_fft_vs_numpy(n):
    """One second of audio at the usual rates, 10^6 and 3 * 2^20 points: four-step over two mixed-radix passes (no
    Bluestein); compute_fft's default n = len(data) lands here."""
    from sygnals_amd import ops
    from sygnals_amd.core.dsp import compute_fft, compute_ifft
    assert ops.smooth_split(n) is not None and not ops.is_pow2(n)
    rng = np.random.default_rng(n % 1000)
    y = rng.normal(0, 0.3, n)
    f, s = compute_fft(y, fs=float(n), window=None)
    ref = np.fft.fft(y.astype(np.float32).astype(np.float64))
    assert peak_rel(s, ref) <= TOL
    back = compute_ifft(s)
    assert_parity(back, y, 2 * TOL, "round trip")


def test_smooth_split_rules():
    from sygnals_amd import ops
    assert ops.smooth_split(48000) == (200, 240) and ops.smooth_split(44100) == (210, 210)
    assert ops.smooth_split(8192) == (8192, 1) and ops.smooth_split(2 * 8192) == (128, 128)
    assert ops.smooth_split(11) is None and ops.smooth_split(48000 * 11) is None
    assert ops.smooth_split(7 ** 9) is None                      # smooth, but no split into two factors <= 8192


@pytest.mark.parametrize("n", [11, 13, 17, 97, 1009, 4099, 9973, 47999, 65537])
def test_fft_prime_lengths_bluestein(n):
    """Lengths with a prime factor above 7: chirp-z over the next 7-smooth transform length."""
    from sygnals_amd import ops
    rng = np.random.default_rng(n)
    z = (rng.normal(0, 1, (2, n)) + 1j * rng.normal(0, 1, (2, n))).astype(np.complex64)
    x = torch.from_numpy(np.stack([z.real, z.imag], axis=-1)).cuda()
    assert ops.smooth_split(n) is None
    got = ops.fft_any(x).cpu().numpy().astype(np.float64)
    assert peak_rel(got[..., 0] + 1j * got[..., 1], np.fft.fft(z.astype(np.complex128), axis=1)) <= TOL
    got = ops.fft_any(x, True).cpu().numpy().astype(np.float64)
    assert peak_rel(got[..., 0] + 1j * got[..., 1], np.fft.ifft(z.astype(np.complex128), axis=1)) <= TOL


def test_generic_stft_batched_over_clips():
    """Frame lengths that are not powers of two: the frames of many clips go through one packing launch
    (syg_pack_frames_f32); every clip equals its own single-clip result and the oracle."""
    from sygnals_amd import ops
    Y = O.synth_clips(37, 9000, 16000, seed=5)
    yd = ops.to_device_f32(Y)
    for n_fft, hop, center in ((1000, 250, True), (600, 200, False), (90, 7, True)):
        X = ops.stft_any(yd, n_fft, hop, center).cpu().numpy().astype(np.float64)
        for b in (0, 17, 36):
            want = O.stft(Y[b].astype(np.float64), n_fft=n_fft, hop_length=hop, center=center).T   # [T, F]
            assert peak_rel(X[b, :, :, 0] + 1j * X[b, :, :, 1], want) <= TOL, (n_fft, b)
    # a non-contiguous view of the clips
    wide = ops.to_device_f32(np.pad(Y, ((0, 0), (3, 5))))[:, 3:3 + 9000]
    X2 = ops.stft_any(wide, 1000, 250, False).cpu().numpy()
    assert np.array_equal(X2, ops.stft_any(yd, 1000, 250, False).cpu().numpy())


def test_reference_dsp_cases():
    """The reference's own cases for these functions (tests/test_dsp.py:109-200), same inputs and assertions."""
    from sygnals_amd.core.dsp import (amplitude_envelope, apply_convolution, compute_autocorrelation, compute_correlation,
                                      compute_psd_periodogram, compute_psd_welch)
    x = np.array([1, 2, 3, 4, 5], dtype=float)
    k = np.array([1, 0, -1], dtype=float)
    np.testing.assert_allclose(apply_convolution(x, k, mode="same"), np.convolve(x, k, mode="same"), atol=1e-5)
    a = np.array([1, 2, 3, 2, 1], dtype=float)
    b = np.array([0, 1, 2, 1, 0], dtype=float)
    assert np.argmax(compute_correlation(a, b, mode="full")) == 4
    fs, freq = 1000, 50.0                                       # the sine_wave fixture: 1 s, amplitude 1
    t = np.linspace(0, 1, fs, endpoint=False)
    s = np.sin(2 * np.pi * freq * t)
    ac = compute_autocorrelation(s, mode="full")
    c = len(s) - 1
    assert np.argmax(ac) == c
    lag = int(round(fs / freq))
    assert ac[c + lag] > 0.8 * ac[c] and ac[c - lag] > 0.8 * ac[c]
    for f, p in (compute_psd_periodogram(s, fs=fs, window="hann"), compute_psd_welch(s, fs=fs, nperseg=256)):
        assert f.shape == p.shape and f.dtype == np.float64 and p.dtype == np.float64
        assert abs(f[np.argmax(p)] - freq) < 1.0 + fs / 256
    env = amplitude_envelope(s, method="hilbert")
    assert env.shape == s.shape and env.dtype == np.float64 and np.all(env >= 0)
    np.testing.assert_allclose(np.mean(env), 1.0, atol=0.05)
    env = amplitude_envelope(s, method="rms", frame_length=256, hop_length=128)
    assert env.dtype == np.float64 and len(env) == 1 + len(s) // 128 and np.all(env >= 0)
    np.testing.assert_allclose(np.mean(env), 1.0 / np.sqrt(2), atol=0.05)


@pytest.mark.parametrize("detrend", [False, "constant", "linear"])
@pytest.mark.parametrize("noverlap", [None, 1000, 1001])
def test_welch_4096_wave_per_segment_kernel(detrend, noverlap):
    """nperseg = nfft = 4096 (config C5's parameters) takes the wave-per-segment kernel when the segments are 16-byte
    aligned (noverlap 2048 / 1000) and the workgroup kernel otherwise (noverlap 1001): both against scipy.signal.welch
    (what compute_psd_welch calls, dsp.py:545-555)."""
    from sygnals_amd.core.dsp import compute_psd_welch
    import scipy.signal
    rng = np.random.default_rng(77)
    sr = 48000
    n = sr * 5 + 123
    t = np.arange(n) / sr
    x = (rng.normal(0, 0.05, n) + 0.3 * np.sin(2 * np.pi * 440.0 * t) + 0.1 * np.sin(2 * np.pi * 9000.3 * t) + 0.2
         + 0.05 * t).astype(np.float32).astype(np.float64)
    f, p = compute_psd_welch(x, fs=sr, nperseg=4096, noverlap=noverlap, detrend=detrend)
    wf, wp = scipy.signal.welch(x, fs=sr, window="hann", nperseg=4096, noverlap=noverlap, detrend=detrend)
    assert np.allclose(f, wf, rtol=0, atol=1e-9)
    assert_parity(p, wp, TOL, f"welch 4096 detrend={detrend} noverlap={noverlap}")
