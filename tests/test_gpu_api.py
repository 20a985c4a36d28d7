"""GPU parity of the reference-shaped Python API (sygnals_amd.core.*) -- the tests read like the
reference's own (tests/test_dsp.py, test_filters.py, test_features_*.py) but pin numbers: against golden
vectors produced by the reference's functions where those run, against the oracle elsewhere."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")
from oracle import cpu_ref as O
from tests.gpu_util import assert_contrast_parity, assert_parity, fft_floor, peak_rel

TOL = 1e-5
G = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def gd():
    return np.load(os.path.join(G, "ref_dsp.npz"))


@pytest.fixture(scope="module", autouse=True)
def _gpu():
    from sygnals_amd import ops
    ops.require_gpu()


@pytest.mark.parametrize("win", ["hann", "hamming", "blackman", None])
@pytest.mark.parametrize("n", [None, 1024, 512, 1500])
def test_compute_fft_vs_reference_golden(gd, win, n):
    from sygnals_amd.core.dsp import compute_fft
    f, s = compute_fft(gd["x1000"], fs=1000.0, n=n, window=win)          # n=None -> 1000 (Bluestein), 1500 too
    assert f.dtype == np.float64 and s.dtype == np.complex128
    assert np.array_equal(f, gd[f"fft_{win}_{n}_f"])
    assert peak_rel(s, gd[f"fft_{win}_{n}_s"]) <= TOL


def test_compute_ifft_vs_reference_golden(gd):
    from sygnals_amd.core.dsp import compute_fft, compute_ifft
    _, sp = O.compute_fft(gd["x1000"], fs=1000.0, window=None)
    assert_parity(compute_ifft(sp), gd["ifft_none"], TOL, "ifft")
    assert_parity(compute_ifft(sp, n=768), gd["ifft_n768"], TOL, "ifft n=768")
    assert_parity(compute_ifft(sp, n=1200), gd["ifft_n1200"], TOL, "ifft n=1200")
    # reference tests/test_dsp.py:69-76: ifft(fft(x)) == x (at fp32 precision here)
    _, s2 = compute_fft(gd["x4096"], fs=48000.0, window=None)
    assert_parity(compute_ifft(s2), gd["x4096"], TOL, "fft/ifft round trip")
    x = compute_ifft(sp)
    assert x.dtype == np.float64


@pytest.mark.parametrize("n", [4096, 16384, 65536, 48000, 12345])
def test_fft_long_and_odd_lengths(n):
    from sygnals_amd.core.dsp import compute_fft, compute_ifft
    rng = np.random.default_rng(n)
    x = rng.normal(size=n).astype(np.float32).astype(np.float64)
    f, s = compute_fft(x, fs=48000.0, window=None)
    ref = np.fft.fft(x)
    assert peak_rel(s, ref) <= TOL
    assert_parity(compute_ifft(ref), x, TOL, f"ifft n={n}")
    # linearity / Parseval: size-independent properties
    assert abs(np.sum(np.abs(s) ** 2) / n - np.sum(x ** 2)) <= 1e-4 * np.sum(x ** 2)


@pytest.mark.parametrize("w", ["hann", "hamming", "blackman", "bartlett", "boxcar"])
def test_apply_window_vs_reference_golden(gd, w):
    from sygnals_amd.core.dsp import apply_window
    assert_parity(apply_window(gd["x1000"], w), gd[f"win_{w}"], TOL, w)


def test_compute_stft_like_reference_tests():
    from sygnals_amd.core.dsp import compute_stft
    sr = 22050
    t = np.arange(2 * sr) / sr
    y = np.sin(2 * np.pi * (100 + 2450 * t / 2) * t)                      # chirp, reference tests/test_dsp.py:79-91
    X = compute_stft(y, n_fft=1024, hop_length=256)
    assert X.shape == (513, 1 + len(y) // 256) and X.dtype == np.complex128
    assert peak_rel(X, O.stft(y.astype(np.float32).astype(np.float64), 1024, 256)) <= TOL
    X2 = compute_stft(y)                                                  # defaults: 2048, hop 512
    assert X2.shape == (1025, 1 + len(y) // 512)
    assert peak_rel(X2, O.stft(y.astype(np.float32).astype(np.float64))) <= TOL
    X3 = compute_stft(y, n_fft=512, pad_mode="reflect")
    assert peak_rel(X3, O.stft(y.astype(np.float32).astype(np.float64), 512, pad_mode="reflect")) <= TOL
    with pytest.raises(ValueError, match="1D"):
        compute_stft(np.zeros((2, 10)))


WELCH = [("w4096", dict(nperseg=4096)), ("w256", dict(nperseg=256)), ("w1024o768", dict(nperseg=1024, noverlap=768)),
         ("w512nfft1024", dict(nperseg=512, nfft=1024)), ("w1024spec", dict(nperseg=1024, scaling="spectrum")),
         ("w1024nodet", dict(nperseg=1024, detrend=False)), ("w1024hamming", dict(nperseg=1024, window="hamming"))]


@pytest.mark.parametrize("tag,kw", WELCH)
def test_compute_psd_welch_vs_reference_golden(gd, tag, kw):
    from sygnals_amd.core.dsp import compute_psd_welch
    f, p = compute_psd_welch(gd["x20000"], fs=48000.0, **kw)
    assert f.dtype == np.float64 and p.dtype == np.float64
    assert np.allclose(f, gd[f"welch_{tag}_f"], rtol=0, atol=1e-9)
    assert_parity(p, gd[f"welch_{tag}_p"], TOL, tag)


def test_filters_vs_reference_golden():
    from sygnals_amd.core import filters as F
    g = np.load(os.path.join(G, "ref_filters.npz"))
    x = g["conv_x"]
    assert_parity(F.band_pass_filter(x, 300.0, 3400.0, 48000.0, order=4), g["conv_bp"], TOL, "band_pass")
    assert_parity(F.low_pass_filter(x, 4000.0, 48000.0), g["conv_lp"], TOL, "low_pass")
    assert_parity(F.high_pass_filter(x, 4000.0, 48000.0), g["conv_hp"], TOL, "high_pass")
    assert_parity(F.band_stop_filter(x, 1000.0, 5000.0, 48000.0), g["conv_bs"], TOL, "band_stop")
    y = F.apply_sos_filter(g["lp8_1k_sos"], g["lp8_1k_x"])
    assert y.dtype == np.float64 and y.shape == g["lp8_1k_x"].shape
    assert_parity(y, g["lp8_1k_y"], TOL, "apply_sos_filter")
    with pytest.raises(ValueError, match="greater than padlen"):
        F.apply_sos_filter(g["lp8_1k_sos"], np.zeros(10))


def test_per_frame_feature_functions_vs_reference_golden():
    from sygnals_amd.core.features import frequency_domain as fd
    g = np.load(os.path.join(G, "ref_freq.npz"))
    S, fr = g["spectra"], g["freqs"]
    for i in (0, 3, 4, 5, 6, 11):
        s = S[i]
        c = fd.spectral_centroid(s, fr)
        assert isinstance(c, np.float64)
        assert abs(c - g["centroid"][i]) <= TOL * g["centroid"].max()
        assert abs(fd.spectral_bandwidth(s, fr) - g["bandwidth"][i]) <= TOL * g["bandwidth"].max()
        assert abs(fd.spectral_bandwidth(s, fr, p=1) - g["bandwidth_p1"][i]) <= TOL * g["bandwidth_p1"].max()
        assert abs(fd.spectral_bandwidth(s, fr, centroid=np.float64(5000.0)) - g["bandwidth_c"][i]) <= 2 * TOL * g["bandwidth_c"].max()
        assert abs(fd.spectral_flatness(s) - g["flatness"][i]) <= TOL * g["flatness"].max()
        assert fd.spectral_rolloff(s, fr) == g["rolloff85"][i]
        assert fd.spectral_rolloff(s, fr, roll_percent=0.5) == g["rolloff50"][i]
        assert fd.dominant_frequency(s, fr) == g["dominant"][i]
    z = S[3]                                                             # all-zero frame constants
    assert fd.spectral_centroid(z, fr) == 0.0 and fd.spectral_bandwidth(z, fr) == 0.0
    assert fd.spectral_flatness(z) == 0.0 and fd.spectral_rolloff(z, fr) == fr[-1] and fd.dominant_frequency(z, fr) == fr[0]


def test_spectral_contrast_api():
    from sygnals_amd.core.features.frequency_domain import spectral_contrast
    y = O.synth_clips(1, 22050, 22050, seed=8)[0].astype(np.float64)
    S = np.abs(O.stft(y, 2048, 512))
    C = spectral_contrast(S, 22050, n_bands=6)
    assert C.shape == (7, S.shape[1]) and C.dtype == np.float64
    assert_parity(C, O.spectral_contrast(S.astype(np.float32).astype(np.float64), 22050), TOL, "contrast")
    Cl = spectral_contrast(S, 22050, linear=True)
    assert_parity(Cl, O.spectral_contrast(S.astype(np.float32).astype(np.float64), 22050, linear=True), TOL, "linear")
    with pytest.raises(ValueError, match="Nyquist"):
        spectral_contrast(np.ones((257, 4)), 8000)


def test_mfcc_function_both_entry_points():
    from sygnals_amd.core.features.cepstral import mfcc
    y = O.synth_clips(1, 22050, 22050, seed=9)[0].astype(np.float64)
    S = O.power_to_db(O.melspectrogram(np.abs(O.stft(y)) ** 2, 22050))
    m = mfcc(S=S, sr=22050, n_mfcc=20)
    assert m.shape == (20, 1 + len(y) // 512) and m.dtype == np.float64      # reference tests/test_features_cepstral.py:47-75
    assert_parity(m, O.mfcc(S=S.astype(np.float32).astype(np.float64), n_mfcc=20), TOL, "mfcc(S)")
    assert_parity(mfcc(S=S, n_mfcc=13), m[:13], 1e-6, "first 13 of 20")     # :94-117
    assert_parity(mfcc(S=S, n_mfcc=13, lifter=22.0), O.mfcc(S=S.astype(np.float32).astype(np.float64), lifter=22.0), TOL, "lifter")
    my = mfcc(y=y, sr=22050, n_mfcc=13, n_fft=1024, hop_length=512)
    assert_parity(my, O.mfcc(y=y.astype(np.float32).astype(np.float64), sr=22050, n_fft=1024, hop_length=512), TOL, "mfcc(y)")


def test_extract_features_like_reference_manager_tests():
    from sygnals_amd.core.features.manager import extract_features
    import pandas as pd
    sr = 22050
    y = O.synth_clips(1, sr, sr, seed=10)[0].astype(np.float64)
    feats = ["spectral_centroid", "spectral_bandwidth", "spectral_flatness", "spectral_rolloff", "dominant_frequency",
             "spectral_contrast", "mfcc"]
    d = extract_features(y, sr, feats, output_format="dict_of_arrays", feature_params={"mfcc": {"n_mfcc": 20}})
    ref = O.extract_features(y.astype(np.float32).astype(np.float64), sr, feats, feature_params={"mfcc": {"n_mfcc": 20}})
    T = 1 + len(y) // 512
    assert set(d) == set(ref) and all(v.shape == (T,) and v.dtype == np.float64 for v in d.values())
    assert np.allclose(d["time"], ref["time"], rtol=0, atol=1e-12)
    st = O.spectral_stats_frames(np.abs(O.stft(y.astype(np.float32).astype(np.float64))), O.fft_frequencies(sr, 2048))
    for k in d:
        if k == "time":
            continue
        if k in ("spectral_rolloff", "dominant_frequency"):
            m = st["rolloff_margin" if k == "spectral_rolloff" else "dominant_margin"] > 1e-6
            assert np.array_equal(d[k][m], ref[k][m]), k
        else:
            assert_parity(d[k], ref[k], TOL, k)
    df = extract_features(y, sr, ["mfcc", "spectral_centroid"])
    assert isinstance(df, pd.DataFrame) and df.index.name == "time" and isinstance(df.index, pd.TimedeltaIndex)
    assert list(df.columns) == [f"mfcc_{i}" for i in range(13)] + ["spectral_centroid"] and len(df) == T
    # other frame sizes take the generic kernels (reference tests use frame 1024 / hop 256)
    d2 = extract_features(y, sr, ["mfcc", "spectral_centroid", "spectral_contrast"], frame_length=1024, hop_length=256,
                          output_format="dict_of_arrays")
    r2 = O.extract_features(y.astype(np.float32).astype(np.float64), sr, ["mfcc", "spectral_centroid", "spectral_contrast"],
                            frame_length=1024, hop_length=256)
    for k in r2:
        assert_parity(d2[k], r2[k], TOL if k != "time" else 1e-12, f"frame 1024: {k}")
    # frame 4096: MFCC alone takes the fused samples -> mel kernel, with a statistic beside it the generic chain
    for feats in (["mfcc"], ["mfcc", "spectral_centroid"]):
        d4 = extract_features(y, sr, feats, frame_length=4096, hop_length=1024, output_format="dict_of_arrays",
                              feature_params={"mfcc": {"n_mels": 40}})
        r4 = O.extract_features(y.astype(np.float32).astype(np.float64), sr, feats, frame_length=4096, hop_length=1024,
                                feature_params={"mfcc": {"n_mels": 40}})
        for k in r4:
            assert_parity(d4[k], r4[k], TOL if k != "time" else 1e-12, f"frame 4096 {feats}: {k}")
    # short signals (reference tests/test_features_manager.py:183-220)
    short = extract_features(y[:512], sr, ["spectral_centroid"], frame_length=1024, hop_length=256, output_format="dict_of_arrays")
    assert short["spectral_centroid"].shape == (3,)
    one = extract_features(y[:100], sr, ["mfcc"], output_format="dict_of_arrays")
    assert one["mfcc_0"].shape == (1,)


def test_cli_features_extract_c1(tmp_path):
    """Config C1: 10 s mono 16 kHz PCM16 WAV through `features extract -f mfcc` (n_mels 128, 13 MFCC)."""
    from click.testing import CliRunner
    from scipy.io import wavfile
    from sygnals_amd.cli.main import cli
    y = O.synth_clips(1, 160000, 16000, seed=12)[0]
    pcm = np.round(y * 32767).astype(np.int16)
    wavfile.write(str(tmp_path / "c1.wav"), 16000, pcm)
    r = CliRunner().invoke(cli, ["features", "extract", str(tmp_path / "c1.wav"), "-o", str(tmp_path / "o.npz"), "-f", "mfcc"])
    assert r.exit_code == 0, r.output
    z = np.load(tmp_path / "o.npz")
    assert set(z.files) == {"time", *[f"mfcc_{i}" for i in range(13)]} and z["mfcc_0"].shape == (313,)
    ref = O.extract_features(pcm.astype(np.float64) / 32768.0, 16000, ["mfcc"])
    got = np.stack([z[f"mfcc_{i}"] for i in range(13)]); exp = np.stack([ref[f"mfcc_{i}"] for i in range(13)])
    assert_parity(got, exp, TOL, "C1 CLI mfcc")
    r = CliRunner().invoke(cli, ["features", "extract", str(tmp_path / "c1.wav"), "-o", str(tmp_path / "o.csv"), "-f", "mfcc",
                                 "-f", "spectral_centroid"])
    assert r.exit_code == 0, r.output
    import pandas as pd
    assert "time" not in pd.read_csv(tmp_path / "o.csv").columns          # reference drops the index (data_handler.py:248)
    r = CliRunner().invoke(cli, ["features", "extract", str(tmp_path / "c1.wav"), "-o", str(tmp_path / "x.npz"), "-f", "bogus"])
    assert r.exit_code == 2 and "Unknown feature" in r.output


def test_compute_cqt_vs_oracle():
    """a15: device CQT against the oracle restatement (same decimation FIR) -- see DESIGN.md for the resampler note."""
    from sygnals_amd.core.dsp import compute_cqt
    sr = 22050
    t = np.arange(2 * sr) / sr
    y = np.sin(2 * np.pi * (60 * t + (900 - 60) / 4 * t * t)).astype(np.float32).astype(np.float64)   # 60 -> 900 Hz
    C = compute_cqt(y, sr, n_bins=60, bins_per_octave=12)                    # reference tests/test_dsp.py:94-106
    assert C.shape == (60, 1 + len(y) // 512) and C.dtype == np.complex128
    assert peak_rel(C, O.cqt(y, sr, n_bins=60)) <= TOL
    pk = np.abs(C).argmax(axis=0)[8:-8]
    assert (np.diff(pk) >= 0).mean() > 0.97
    y2 = O.synth_clips(1, 48000, 48000, seed=4)[0].astype(np.float64)
    C2 = compute_cqt(y2, 48000)                                              # config C5 parameters: 84 bins from C1
    assert C2.shape == (84, 94)
    assert peak_rel(C2, O.cqt(y2, 48000)) <= TOL
    C3 = compute_cqt(y2[:20001], 48000, hop_length=256, n_bins=36, bins_per_octave=12, fmin=110.0)
    assert peak_rel(C3, O.cqt(y2[:20001], 48000, hop_length=256, n_bins=36, fmin=110.0)) <= TOL
    with pytest.raises(ValueError, match="1D"):
        compute_cqt(np.zeros((2, 100)), sr)
    with pytest.raises(ValueError, match="Nyquist"):
        compute_cqt(y2, 8000)
    # librosa's default spelling of the resampler is accepted (with the logged deviation), another one is refused
    assert np.array_equal(compute_cqt(y2, 48000, res_type="soxr_hq"), C2)
    assert np.array_equal(compute_cqt(y2, 48000, res_type="kaiser_halfband"), C2)
    from sygnals_amd._lib import SygnalsHipError
    with pytest.raises(SygnalsHipError, match="res_type"):
        compute_cqt(y2, 48000, res_type="kaiser_best")


@pytest.mark.parametrize("path", ["bf16x3", "gemm", "fft"])
def test_cqt_both_octave_kernels_vs_oracle(path, monkeypatch):
    """The octave response as one framed matrix product on the matrix cores -- with bfloat16-split operands (default,
    fp32-equivalent) or as single fp32 MFMA instructions -- and as rfft x sparse basis rows (the fallback for other
    frame lengths) are the same linear map: all three against the oracle at 1e-5."""
    from sygnals_amd import ops
    monkeypatch.setattr(ops.settings, "cqt_mode", path)
    rng = np.random.default_rng(11)
    sr = 48000
    n = sr * 3 + 77                                          # odd length: ragged decimated lengths, masked edge tiles
    t = np.arange(n) / sr
    x = (rng.normal(0, 0.05, n) + 0.4 * np.sin(2 * np.pi * 261.63 * t) + 0.2 * np.sin(2 * np.pi * 3135.96 * t)).astype(np.float32)
    ref = O.cqt(x.astype(np.float64), sr)
    got = ops.cqt(ops.to_device_f32(np.stack([x, -2 * x])), sr).cpu().numpy()
    got = got[..., 0] + 1j * got[..., 1]
    assert got.shape == (2,) + ref.shape
    assert peak_rel(got[0], ref) <= TOL and peak_rel(got[1], -2 * ref) <= TOL


@pytest.mark.parametrize("sr,hop,n_bins,L,B", [(48000, 512, 84, 48000 * 2 + 77, 1), (48000, 512, 84, 48000 * 2 + 76, 2),
                                              (48000, 1024, 84, 48000 * 3 + 5, 1), (48000, 2048, 84, 48000 * 4, 1),
                                              (22050, 512, 84, 22050 * 3 + 1, 1), (48000, 384, 72, 48000 + 3, 1),
                                              (48000, 512, 84, 3000, 1)])
def test_cqt_staged_frames_identical(sr, hop, n_bins, L, B, monkeypatch):
    """Octaves whose frames overlap split the sample run of a 16-frame tile once in LDS (cqt_bf16x3_staged_kernel); the
    operands and the order of the matrix instructions per accumulator are those of the per-frame kernel: identical
    bits, for every slot skew (hop / 8 = 1 ... 16), hops of 4 mod 8 (two shifted copies), ragged tails, tiles that start
    before and end past the signal, and a batch."""
    from sygnals_amd import ops
    import torch
    rng = np.random.default_rng(L)
    x = ops.to_device_f32(rng.normal(0, 0.3, (B, L)).astype(np.float32))
    monkeypatch.setattr(ops.settings, "cqt_mode", "bf16x3")
    monkeypatch.setattr(ops.settings, "cqt_fused", False)      # (the level-by-level kernels are the subject here)
    with ops.override(cqt_staged=2):                            # 2: also where hop = n_fft / 2 (not the default there)
        a = ops.cqt(x, sr, hop_length=hop, n_bins=n_bins)
    with ops.override(cqt_staged=1):
        a1 = ops.cqt(x, sr, hop_length=hop, n_bins=n_bins)
    with ops.override(cqt_staged=0):
        b = ops.cqt(x, sr, hop_length=hop, n_bins=n_bins)
    assert ops.get_option("cqt_staged") == -1
    assert a.shape == b.shape and torch.equal(a, b) and torch.equal(a1, b)
    if L <= 48000 * 2 + 77 and B == 1:
        ref = O.cqt(x[0].cpu().numpy().astype(np.float64), sr, hop_length=hop, n_bins=n_bins)
        got = a[0].cpu().numpy()
        assert peak_rel(got[..., 0] + 1j * got[..., 1], ref) <= TOL


@pytest.mark.parametrize("B,L", [(1, 1), (1, 5), (2, 101), (1, 3823), (3, 3824 * 2 + 1), (1, 478 * 8 * 9), (2, 1 << 20), (1, (1 << 21) + 12345),
                                 (2, 22050), (3, (1 << 20) + 2)])   # rows of L % 4 == 2: no 16-byte row starts
def test_decimate2_chain_identical_to_level_by_level(B, L):
    """Two or three decimations carried through LDS in one pass: the same bits as one launch per level (zero padding of
    every level at both ends, tiles whose halo crosses the signal's ends, ragged lengths, batches with odd row strides),
    levels that are not wanted left out, and a tap count without the fused form falling back to level-by-level."""
    from sygnals_amd import ops
    from sygnals_amd._cqt import decimation_taps
    import torch
    rng = np.random.default_rng(L)
    x = ops.to_device_f32(rng.normal(0, 1, (B, L)).astype(np.float32))
    taps = ops.to_device_f32(decimation_taps().astype(np.float32))
    s2 = float(np.sqrt(2.0))
    ref, cur = [], x
    for _ in range(7):
        cur = ops.decimate2(cur, taps, s2)
        ref.append(cur)
    for levels in (1, 2, 3, 4, 5, 7):
        got = ops.decimate2_chain(x, taps, s2, levels)
        assert len(got) == levels
        for g, r in zip(got, ref):
            assert g.shape == r.shape and torch.equal(g, r)
    got = ops.decimate2_chain(x, taps, s2, 6, keep=[False, True, False, False, True, False])
    assert got[0] is None and got[2] is None and got[3] is None
    assert torch.equal(got[1], ref[1]) and torch.equal(got[4], ref[4]) and torch.equal(got[5], ref[5])
    t9 = ops.to_device_f32(rng.normal(0, 0.3, 9).astype(np.float32))
    a = ops.decimate2_chain(x, t9, 1.0, 2)
    assert torch.equal(a[1], ops.decimate2(ops.decimate2(x, t9, 1.0), t9, 1.0))


def test_cqt_chain_and_level_by_level_identical(monkeypatch):
    from sygnals_amd import ops
    import torch
    x = ops.to_device_f32(np.random.default_rng(3).normal(0, 0.3, (2, 48000 * 5 + 3)).astype(np.float32))
    with ops.override(cqt_fused=False):
        a = ops.cqt(x, 48000)
        with ops.override(cqt_chain=False):
            assert torch.equal(a, ops.cqt(x, 48000))


@pytest.mark.parametrize("sr,n_bins,fmin,L,B", [
    (48000, 84, None, 48000 * 2 + 77, 1),     # one segment, ragged level lengths
    (48000, 84, None, 48000 * 9 + 5, 2),      # several segments per signal, a batch with an odd row stride
    (48000, 84, None, 3000, 1), (48000, 84, None, 1, 1), (48000, 84, None, 511, 3), (48000, 84, None, 512, 1),
    (48000, 84, None, 65536 * 3, 1),          # frames exactly on segment borders
    (48000, 36, 523.25, 48000 * 3 + 1, 1),    # three octaves
    (48000, 48, 261.63, 100003, 2),           # four octaves
    (44100, 84, None, 44100 * 4, 1),
    (48000, 60, None, 48000 + 1, 1)])         # (three early decimations: not this form's shape -- the level-by-level kernels)
def test_cqt_one_launch_form(sr, n_bins, fmin, L, B):
    """syg_cqt_fused_f32 (every decimation level in LDS only, all octaves' products in the same launch) against the
    level-by-level kernels -- the same operands; the float32 sums are grouped differently (symmetric tap pairs, the
    products' k range in four parts): 2e-6 of the peak -- and against the oracle at 1e-5: segment borders, lead-in at the
    signal's start, level lengths that round up, signals shorter than a frame, fewer than seven octaves."""
    from sygnals_amd import ops
    rng = np.random.default_rng(L + B)
    t = np.arange(L) / sr
    x = (rng.normal(0, 0.2, (B, L)) + 0.4 * np.sin(2 * np.pi * 110.0 * t) + 0.3 * np.sin(2 * np.pi * 2500.0 * t)).astype(np.float32)
    xd = ops.to_device_f32(x)
    a = ops.cqt(xd, sr, n_bins=n_bins, fmin=fmin).cpu().numpy()
    with ops.override(cqt_fused=False):
        b = ops.cqt(xd, sr, n_bins=n_bins, fmin=fmin).cpu().numpy()
    assert a.shape == b.shape
    pk = np.abs(b).max()
    assert np.abs(a - b).max() <= 2e-6 * pk + 1e-30
    if L <= 48000 * 3 + 1:
        ref = O.cqt(x[0].astype(np.float64), sr, n_bins=n_bins, fmin=fmin)
        assert peak_rel(a[0, ..., 0] + 1j * a[0, ..., 1], ref) <= TOL


def test_cqt_one_launch_form_is_taken(monkeypatch):
    """The default call of compute_cqt's shape (48 kHz, hop 512, 84 bins) runs ONE launch; other shapes (hop 1024: two
    early decimations) keep the level-by-level kernels."""
    from sygnals_amd import ops
    calls = []
    lib = ops.lib()
    real = lib.syg_cqt_fused_f32
    x = ops.to_device_f32(np.random.default_rng(1).normal(0, 0.3, (1, 48000)).astype(np.float32))

    class Spy:
        def __getattr__(self, name):
            if name == "syg_cqt_fused_f32":
                return lambda *a: (calls.append(name), real(*a))[1]
            if name.startswith("syg_cqt_octave") or name.startswith("syg_decimate2"):
                calls.append(name)
            return getattr(lib, name)
    monkeypatch.setattr(ops, "lib", lambda: Spy())
    ops.cqt(x, 48000)
    assert calls == ["syg_cqt_fused_f32"]
    calls.clear()
    ops.cqt(x, 48000, hop_length=1024)
    assert calls and "syg_cqt_fused_f32" not in calls


def test_cqt_batch_long_stream_consistency():
    """C5-shaped use: a batch of long streams; size-independent property: linearity and time-shift by whole hops."""
    from sygnals_amd import ops
    rng = np.random.default_rng(5)
    L = 48000 * 8
    x = rng.normal(0, 0.1, (2, L)).astype(np.float32)
    X = ops.cqt(ops.to_device_f32(x), 48000).cpu().numpy()
    X = X[..., 0] + 1j * X[..., 1]
    S = ops.cqt(ops.to_device_f32((x[0] + 2 * x[1])[None]), 48000).cpu().numpy()
    S = S[0, ..., 0] + 1j * S[0, ..., 1]
    assert peak_rel(S, X[0] + 2 * X[1]) <= TOL
    ref = O.cqt(x[0, :48000 * 2].astype(np.float64), 48000)
    got = X[0][:, :150]
    assert peak_rel(got[:, :130], ref[:, :130]) <= TOL      # (frames clear of the 2 s excerpt's end: the lowest octave's
                                                            #  filters are 0.34 s long on either side)


def test_c4_share_full_size_batch_consistency():
    """Config C4's per-GPU share (2048 clips x 1 s @ 48 kHz: MFCC + centroid + rolloff + contrast).  Size-independent
    property: every clip of the big batch equals the same clip run in a 16-clip batch; the oracle checks a sample."""
    from sygnals_amd.core.features.manager import extract_features_batch
    Y = O.synth_clips(16, 48000, 48000, seed=21)
    feats = ["mfcc", "spectral_centroid", "spectral_rolloff", "spectral_contrast"]
    fp = {"mfcc": {"n_mels": 40}}
    big = extract_features_batch(np.tile(Y, (128, 1)), 48000, feats, feature_params=fp)
    small = extract_features_batch(Y, 48000, feats, feature_params=fp)
    assert set(big) == set(small) and big["mfcc_0"].shape == (2048, 94)
    for k in small:
        if k == "time":
            continue
        assert np.array_equal(big[k].reshape(128, 16, -1), np.broadcast_to(small[k], (128, 16, small[k].shape[1]))), k
    # the oracle on a sample of clips, at the north-star tolerance: bin-valued rolloff identical wherever the float64
    # decision margin exceeds 1e-6 (<= 1 bin elsewhere), contrast dB margin-qualified against the fp32 FFT floor of
    # the frame (tests/gpu_util.assert_contrast_parity), everything else 1e-5 peak-relative
    fr = O.fft_frequencies(48000, 2048)
    bands = O.contrast_bands(fr, 48000)
    unsure = cells = flips = 0
    for i in (3, 8, 14):
        ref = O.extract_features(Y[i].astype(np.float64), 48000, feats, feature_params=fp)
        S = np.abs(O.stft(Y[i].astype(np.float64), 2048, 512))
        st = O.spectral_stats_frames(S, fr)
        for k in ref:
            if k == "time" or k.startswith("contrast"):
                continue
            if k == "spectral_rolloff":
                sure = st["rolloff_margin"] > 1e-6
                assert np.array_equal(small[k][i][sure], ref[k][sure]), "rolloff outside the decision margin"
                assert (np.abs(small[k][i] - ref[k]) <= 48000 / 2048 + 1e-9).all()
                flips += int((small[k][i] != ref[k]).sum())
            else:
                assert_parity(small[k][i], ref[k], TOL, k)
        names = [f"contrast_band_{j}" for j in range(6)] + ["contrast_delta"]
        valley = np.stack([np.sort(S[bins], axis=0)[:kk].mean(axis=0) for bins, kk in bands])
        u, n = assert_contrast_parity(np.stack([small[k][i] for k in names]), np.stack([ref[k] for k in names]), valley,
                                      fft_floor(S), TOL, f"C4 contrast clip {i}")
        unsure += u; cells += n
    print(f"C4 sample: rolloff bins inside the 1e-6 margin that differ: {flips}; contrast cells below the decision "
          f"margin: {unsure} of {cells}")
    assert unsure <= 0.10 * cells


def test_c5_full_size_stream_welch_and_cqt():
    """Config C5 on one GPU: a 1-hour 48 kHz stream (172.8 M samples), Welch nperseg 4096 / 50 % and the 84-bin CQT.
    The oracle covers a 2-minute excerpt; at full size: exact power scaling (x2 -> PSD x4), and the CQT of the
    stream delayed by a whole number of hops equals the shifted CQT away from the edges."""
    from sygnals_amd import ops
    from sygnals_amd.core import dsp as D
    sr, L = 48000, 48000 * 3600
    g = torch.Generator(device="cuda").manual_seed(5)
    x = torch.randn(L, device="cuda", generator=g, dtype=torch.float32) * 0.05
    for c0 in range(0, L, 1 << 24):                    # tone phases in float64: float32 loses the sample index past 2^24
        t = torch.arange(c0, min(c0 + (1 << 24), L), device="cuda", dtype=torch.float64)
        x[c0:c0 + t.numel()] += (0.3 * torch.sin(2 * np.pi * 440.0 / sr * t)
                                 + 0.2 * torch.sin(2 * np.pi * 3000.5 / sr * t)).float()
    del t
    x = x.reshape(1, L)
    f, P = D.welch_batch(x, fs=sr, nperseg=4096)
    P = P.cpu().numpy()[0].astype(np.float64)
    assert P.shape == (2049,) and np.isfinite(P).all()
    assert abs(f[np.argmax(P)] - 440.0) < sr / 4096
    _, P2 = D.welch_batch(x * 2.0, fs=sr, nperseg=4096)
    assert np.array_equal(P2.cpu().numpy()[0], (4.0 * P).astype(np.float32))          # exact in floating point
    ex = x[:, : sr * 120]
    _, Pe = D.welch_batch(ex, fs=sr, nperseg=4096)
    fo, Po = O.compute_psd_welch(ex.cpu().numpy()[0].astype(np.float64), sr, "hann", 4096)
    assert peak_rel(Pe.cpu().numpy()[0], Po) <= TOL
    # Parseval: the one-sided density integrates to the signal power
    power = float((x.double() ** 2).mean())
    assert abs(P.sum() * (sr / 4096) - power) <= 2e-3 * power
    # CQT: frames of the delayed stream are the shifted frames (interior)
    C = ops.cqt(x, sr)
    assert tuple(C.shape) == (1, 84, 1 + L // 512, 2)
    k = 1024
    Cs = ops.cqt(x[:, 512 * k:], sr)
    a = C[0, :, k + 200: k + 5200].cpu().numpy().astype(np.float64)
    b = Cs[0, :, 200:5200].cpu().numpy().astype(np.float64)
    assert np.abs(a - b).max() <= TOL * np.abs(a).max()
    exc = x[0, : sr * 4].cpu().numpy().astype(np.float64)
    ref = O.cqt(exc, sr)
    got = C[0, :, :300].cpu().numpy()
    got = got[..., 0] + 1j * got[..., 1]
    assert peak_rel(got[:, :300], ref[:, :300]) <= TOL       # frames clear of the excerpt's end


def test_c3_full_size_filter_then_mfcc():
    """Config C3: the 1024-clip batch through the order-4 Butterworth band-pass filtfilt, then STFT -> MFCC.
    Size-independent property: batch consistency; the oracle (SciPy sosfiltfilt + float64 MFCC chain) checks a sample."""
    from sygnals_amd import ops
    from sygnals_amd.core.filters import apply_sos_filter_batch
    Y = O.synth_clips(16, 48000, 48000, seed=31)
    sos = O.design_butterworth_sos((300.0, 3400.0), 48000, 4, "bandpass")
    big = ops.to_device_f32(np.tile(Y, (64, 1)))
    out = ops.mfcc_batch(apply_sos_filter_batch(sos, big), 48000, n_mels=40).cpu().numpy()
    small = ops.mfcc_batch(apply_sos_filter_batch(sos, ops.to_device_f32(Y)), 48000, n_mels=40, fused=True).cpu().numpy()
    assert out.shape == (1024, 13, 94)
    assert np.array_equal(out.reshape(64, 16, 13, 94), np.broadcast_to(small, (64, 16, 13, 94)))
    for i in (0, 5, 9, 13):
        yf = O.apply_sos_filter(sos, Y[i].astype(np.float64))
        ref = O.mfcc_manager(yf, 48000, n_mels=40)
        assert_parity(small[i], ref, TOL, f"C3 clip {i}")


# ---- a16: the manager's per-feature semantics (manager.py:242-420), mirrored and checked against the oracle's
# in-order restatement
def test_bandwidth_alone_also_emits_the_centroid_it_depends_on():
    """manager.py:296-301: 'spectral_bandwidth' without 'spectral_centroid' stores the centroid as a column too."""
    from sygnals_amd.core.features.manager import extract_features
    y = O.synth_clips(1, 8192, 16000, seed=5)[0].astype(np.float64)
    out = extract_features(y, 16000, ["spectral_bandwidth", "mean_amplitude"], output_format="dict_of_arrays")
    ref = O.extract_features(y, 16000, ["spectral_bandwidth", "mean_amplitude"])
    assert list(out) == list(ref) == ["time", "spectral_centroid", "spectral_bandwidth", "mean_amplitude"]
    for k in ref:
        assert_parity(out[k], ref[k], TOL, k)
    df = extract_features(y, 16000, ["spectral_bandwidth"])
    assert list(df.columns) == ["spectral_centroid", "spectral_bandwidth"]


def test_a_failing_feature_leaves_the_others_in_the_output(monkeypatch, caplog):
    """manager.py:394-397: a feature that raises is logged and skipped; the rest of the call still comes out."""
    import logging
    from sygnals_amd import ops
    from sygnals_amd.core.features import manager as M
    y = O.synth_clips(1, 8192, 16000, seed=6)[0].astype(np.float64)

    def boom(*a, **k):
        raise RuntimeError("injected failure")

    monkeypatch.setattr(ops, "contrast_db", boom)
    with caplog.at_level(logging.ERROR):
        out = M.extract_features(y, 16000, ["spectral_centroid", "spectral_contrast", "mfcc", "hnr", "rms_energy"],
                                 output_format="dict_of_arrays", feature_params={"mfcc": {"n_mels": 40}})
    assert "contrast_band_0" not in out and "hnr" not in out
    assert {"spectral_centroid", "mfcc_0", "mfcc_12", "rms_energy"} <= set(out)
    assert "Error extracting feature 'spectral_contrast': injected failure" in caplog.text
    assert "Error extracting feature 'hnr'" in caplog.text
    ref = O.extract_features(y, 16000, ["spectral_centroid", "mfcc", "rms_energy"], feature_params={"mfcc": {"n_mels": 40}})
    for k in ref:
        assert_parity(out[k], ref[k], TOL, k)
    # every STFT-based feature failing (here: an invalid parameter) still returns the time-domain ones
    out2 = M.extract_features(y, 16000, ["spectral_rolloff", "rms_energy"], output_format="dict_of_arrays",
                              feature_params={"spectral_rolloff": {"roll_percent": 1.5}})
    assert list(out2) == ["time", "rms_energy"]
    # ... and costs ONLY the offending feature: the other STFT-based ones still come out (the reference validates inside
    # each feature's own try, frequency_domain.py:116, 314)
    with caplog.at_level(logging.ERROR):
        out3 = M.extract_features(y, 16000, ["spectral_centroid", "spectral_rolloff", "spectral_bandwidth", "mfcc"],
                                  output_format="dict_of_arrays",
                                  feature_params={"spectral_rolloff": {"roll_percent": 1.5}, "spectral_bandwidth": {"p": 0},
                                                  "mfcc": {"n_mels": 40}})
    assert "spectral_rolloff" not in out3 and "spectral_bandwidth" not in out3
    assert "roll_percent must be between 0.0 and 1.0." in caplog.text and "must be positive" in caplog.text
    ref3 = O.extract_features(y, 16000, ["spectral_centroid", "mfcc"], feature_params={"mfcc": {"n_mels": 40}})
    assert set(out3) == set(ref3)
    for k in ref3:
        assert_parity(out3[k], ref3[k], TOL, k)


def test_odd_frame_length_retimes_from_the_stft_like_the_reference():
    """manager.py:186-194 + 408-420: with an odd frame_length and hop | len(y) the STFT has one frame less than the
    frame-count rule; `time` is re-made from it, rows computed BEFORE with the longer count are dropped by the final
    length check, rows computed AFTER are cut to the new count."""
    from sygnals_amd.core.features.manager import extract_features
    y = O.synth_clips(1, 4096, 16000, seed=2)[0].astype(np.float64)
    feats = ["mean_amplitude", "spectral_centroid", "rms_energy", "mfcc"]
    kw = dict(frame_length=1023, hop_length=256, feature_params={"mfcc": {"n_mels": 32}})
    out = extract_features(y, 16000, feats, output_format="dict_of_arrays", **kw)
    ref = O.extract_features(y, 16000, feats, **kw)
    assert len(ref["time"]) == 16 and "mean_amplitude" not in ref and "rms_energy" in ref
    assert list(out) == list(ref)
    for k in ref:
        assert out[k].shape == ref[k].shape and not np.isnan(out[k]).any()
        assert_parity(out[k], ref[k], TOL, k)


def test_c4_feature_block_equals_the_manager_columns():
    """feature_block (the [B, 22, 94] block config C4 gathers) holds exactly the manager's columns, in order."""
    from sygnals_amd import ops
    from sygnals_amd.core.features.manager import extract_features_batch, feature_block
    Y = O.synth_clips(6, 48000, 48000, seed=41)
    blk = feature_block(ops.to_device_f32(Y), 48000).cpu().numpy()
    assert blk.shape == (6, 22, 94)
    feats = ["mfcc", "spectral_centroid", "spectral_rolloff", "spectral_contrast"]
    d = extract_features_batch(Y, 48000, feats, feature_params={"mfcc": {"n_mels": 40}})
    names = [k for k in d if k != "time"]
    assert names == [f"mfcc_{i}" for i in range(13)] + ["spectral_centroid", "spectral_rolloff"] + \
        [f"contrast_band_{i}" for i in range(6)] + ["contrast_delta"]
    blk2 = feature_block(ops.to_device_f32(Y), 48000, one_launch=False).cpu().numpy()      # mel -> feature_block form
    for r, k in enumerate(names):
        # the manager takes the same one-launch kernel as the block (syg_stft2048_features_tri_f32): the same bits
        assert np.array_equal(blk[:, r].astype(np.float64), d[k]), k
        if r < 13:
            # the one-launch form converts to dB with the hardware log2 (like syg_stft2048_mfcc_f32), the two-launch form
            # with log10f: same values within a fifth of the parity gate, not the same bits
            assert peak_rel(blk2[:, r].astype(np.float64), d[k]) <= 2e-6, k
        else:
            assert np.array_equal(blk2[:, r].astype(np.float64), d[k]), k
    # ... and with the one-launch forms switched off the manager's columns are the two-launch block's, bit for bit
    with ops.override(one_launch_features=False):
        d2 = extract_features_batch(Y, 48000, feats, feature_params={"mfcc": {"n_mels": 40}})
    for r, k in enumerate(names):
        assert np.array_equal(blk2[:, r].astype(np.float64), d2[k]), k


@pytest.mark.parametrize("hop,center,window,n_mels,feats", [
    (512, True, "hann", 40, ["mfcc", "spectral_centroid", "spectral_bandwidth", "spectral_flatness", "dominant_frequency"]),
    (256, False, "hamming", 32, ["spectral_contrast", "mfcc"]),
    (512, True, "hann", 128, ["mfcc", "spectral_rolloff"]),          # no segment-sum form at 128 bands: the mel launch
    (1024, True, "hann", 40, ["mfcc", "spectral_centroid"]),         # hop > 512: the mel launch
    (512, True, "hann", 40, ["mfcc"]),
    (441, True, "blackman", 26, ["spectral_centroid", "spectral_rolloff", "spectral_contrast"]),   # no mel-based feature
])
def test_manager_one_launch_routes_match_the_mel_route(hop, center, window, n_mels, feats):
    """extract_features_batch through the one-launch kernels (the default) and with them switched off
    (ops.settings.one_launch_features = False: mel launch + logmel_dct): same columns, statistics / contrast bit for bit, MFCC
    within a fifth of the parity gate (hardware log2 vs log10f); ragged clip length, a silent clip."""
    from sygnals_amd import ops
    from sygnals_amd.core.features.manager import extract_features_batch
    rng = np.random.default_rng(19)
    Y = (rng.normal(0, 0.2, (9, 30011)) * rng.random((9, 1))).astype(np.float32)
    Y[4] = 0.0
    kw = dict(hop_length=hop, center=center, window=window, feature_params={"mfcc": {"n_mels": n_mels}})
    a = extract_features_batch(Y, 22050 if n_mels == 26 else 48000, feats, **kw)
    with ops.override(one_launch_features=False):
        b = extract_features_batch(Y, 22050 if n_mels == 26 else 48000, feats, **kw)
    assert list(a) == list(b) and len(a) > 1
    for k in a:
        if k.startswith("mfcc_"):
            assert a[k].shape == b[k].shape
        else:
            assert np.array_equal(a[k], b[k], equal_nan=True), k
    if "mfcc" in feats:
        A = np.stack([a[f"mfcc_{i}"] for i in range(13)], 1); Bm = np.stack([b[f"mfcc_{i}"] for i in range(13)], 1)
        for c in range(Y.shape[0]):
            assert peak_rel(A[c], Bm[c]) <= 2e-6 or np.abs(Bm[c]).max() == 0, c

