"""GPU parity of the time-domain frame features against the oracle and the reference's golden vectors.
Calls go through the C ABI (sygnals_amd.ops -> libsygnals_hip.so: syg_frame_stats_f32, syg_rms_from_spec_f32)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")
from oracle import cpu_ref as O
from tests.gpu_util import assert_parity

TOL = 1e-5
G = os.path.join(os.path.dirname(__file__), "golden")
NAMES = ["mean_amplitude", "std_dev_amplitude", "skewness", "kurtosis", "peak_amplitude", "crest_factor",
         "signal_entropy"]


@pytest.fixture(scope="module")
def ops():
    from sygnals_amd import ops
    ops.require_gpu()
    return ops


def test_frames_match_reference_golden(ops):
    """Every frame of the reference's golden set as one 512-sample clip (center=False -> exactly one frame)."""
    g = np.load(os.path.join(G, "ref_time.npz"))
    fr = g["frames"]
    st = ops.frame_stats(ops.to_device_f32(fr), 512, 512, False).cpu().numpy().astype(np.float64)
    assert st.shape == (fr.shape[0], 9, 1)
    for r, nm in enumerate(NAMES):
        assert_parity(st[:, r, 0], g[nm], TOL, nm)
    for nb in (4, 32):
        s2 = ops.frame_stats(ops.to_device_f32(fr), 512, 512, False, num_bins=nb, mask=64).cpu().numpy()
        assert_parity(s2[:, 6, 0].astype(np.float64), g[f"signal_entropy_b{nb}"], TOL, f"entropy bins={nb}")


@pytest.mark.parametrize("L,flen,hop,center", [(48000, 2048, 512, True), (5000, 1024, 256, True),
                                               (4999, 400, 160, False), (700, 2048, 512, True), (3000, 63, 17, True)])
def test_framed_clips_match_oracle(ops, L, flen, hop, center):
    Y = O.synth_clips(3, L, 16000, seed=L + flen)
    Y[1] += 0.2                                   # an offset clip: mean != 0 in every frame
    st = ops.frame_stats(ops.to_device_f32(Y), flen, hop, center).cpu().numpy().astype(np.float64)
    for b in range(Y.shape[0]):
        y = Y[b].astype(np.float64)
        ref = O.time_features_frames(y, flen, hop, center)
        for r, nm in enumerate(NAMES):
            assert_parity(st[b, r], ref[nm], TOL, f"{nm} clip {b}")
        assert_parity(st[b, 7], O.rms_energy(y, frame_length=flen, hop_length=hop, center=center), TOL, "rms")
        z = O.zero_crossing_rate(y, flen, hop, center)
        assert np.array_equal(np.round(st[b, 8] * flen), np.round(z * flen)), "zero-crossing counts must be identical"


def test_mask_selects_rows_and_single_frame_mirror(ops):
    from sygnals_amd.core.features import time_domain as td
    from sygnals_amd.core.audio import features as af
    y = O.synth_clips(1, 8000, 16000, seed=3)[0].astype(np.float64)
    st = ops.frame_stats(ops.to_device_f32(y[None, :]), 1024, 256, True, mask=(1 << 2) | (1 << 7)).cpu().numpy()
    assert np.all(st[0, [0, 1, 3, 4, 5, 6, 8]] == 0.0) and np.any(st[0, 2] != 0.0) and np.all(st[0, 7] > 0.0)
    f = y[1000:1512]
    for nm, fn in td.TIME_DOMAIN_FEATURES.items():
        assert abs(fn(f) - O._TIME_FUNCS[nm](f)) <= TOL * max(1.0, abs(O._TIME_FUNCS[nm](f))), nm
    e = np.array([], dtype=np.float64)
    assert all(fn(e) == 0.0 for fn in td.TIME_DOMAIN_FEATURES.values())
    assert td.skewness(np.array([1.0])) == 0.0 and td.kurtosis_val(np.array([1.0, 2.0, 4.0])) == 0.0
    assert td.signal_entropy(np.ones(64)) == 0.0 and td.crest_factor(np.zeros(64)) == 0.0
    assert_parity(af.rms_energy(y, frame_length=1024, hop_length=256), O.rms_energy(y, frame_length=1024, hop_length=256),
                  TOL, "rms mirror")
    assert_parity(af.zero_crossing_rate(y, 1024, 256), O.zero_crossing_rate(y, 1024, 256), TOL, "zcr mirror")
    S = np.abs(O.stft(y, 1024, 256, window="hann"))
    assert_parity(af.rms_energy(S=S, frame_length=1024), O.rms_energy(S=S, frame_length=1024), TOL, "rms from S")
    with pytest.raises(ValueError):
        af.rms_energy()
    with pytest.raises(ValueError):
        af.zero_crossing_rate(np.zeros((2, 8)))


def test_manager_extracts_time_and_spectral_rows_together(ops):
    from sygnals_amd.core.features.manager import extract_features
    y = O.synth_clips(1, 16000, 16000, seed=9)[0].astype(np.float64)
    feats = ["rms_energy", "zero_crossing_rate", "skewness", "signal_entropy", "spectral_centroid", "mfcc"]
    out = extract_features(y, 16000, feats, output_format="dict_of_arrays", feature_params={"mfcc": {"n_mels": 40}})
    ref = O.extract_features(y, 16000, feats, feature_params={"mfcc": {"n_mels": 40}})
    assert set(out) == set(ref)
    for k in ref:
        assert out[k].dtype == np.float64 and out[k].shape == ref[k].shape
        assert_parity(out[k], ref[k], TOL, k)


def test_odd_frame_length_last_frame_is_nan_padded_like_the_reference(ops):
    """frame_length odd and hop | len(y): the reference's padded framing is one sample short of the last frame and
    NaN-pads it (manager.py:268-271, 378-386)."""
    from sygnals_amd.core.features.manager import extract_features
    y = O.synth_clips(1, 4096, 16000, seed=2)[0].astype(np.float64)
    feats = ["mean_amplitude", "rms_energy", "zero_crossing_rate", "kurtosis"]
    out = extract_features(y, 16000, feats, frame_length=1023, hop_length=256, output_format="dict_of_arrays")
    ref = O.extract_features(y, 16000, feats, frame_length=1023, hop_length=256)
    assert len(out["time"]) == 17
    for k in feats:
        assert np.isnan(ref[k][-1]) and np.isnan(out[k][-1]) and not np.isnan(out[k][:-1]).any()
        assert_parity(out[k][:-1], ref[k][:-1], TOL, k)


def test_histogram_counts_identical_on_hard_cases(ops):
    """The entropy row bins float32 samples against float32-rounded-up edges; the counts must equal NumPy's float64
    histogram exactly: samples sitting on bin edges, a value range of a few denormals (the float guess is useless
    there), negative ranges, 2..200 bins (packed counters up to 16, the shared LDS histogram above)."""
    rng = np.random.default_rng(17)
    N = 640
    cases = []
    for nb in (2, 3, 7, 10, 16, 17, 64, 200):
        edges = np.linspace(-1.0, 3.0, nb + 1)
        on_edges = np.resize(edges.astype(np.float32), N)               # float32 neighbours of the float64 edges
        on_edges[::3] = np.nextafter(on_edges[::3], np.float32(4))
        on_edges[1::3] = np.nextafter(on_edges[1::3], np.float32(-4))
        on_edges[0], on_edges[1] = -1.0, 3.0
        cases.append((nb, on_edges))
        cases.append((nb, (rng.integers(0, 5, N) * np.float32(1e-45)).astype(np.float32)))     # denormal range
        cases.append((nb, (-rng.random(N) * 1e-3 - 5.0).astype(np.float32)))                   # narrow, negative
        cases.append((nb, rng.normal(0, 1, N).astype(np.float32) * np.float32(1e-38)))         # near-denormal scale
    for nb, x in cases:
        got = ops.frame_stats(ops.to_device_f32(x[None, :]), N, N, False, num_bins=nb, mask=64).cpu().numpy()[0, 6, 0]
        want = O.signal_entropy(x.astype(np.float64), nb)
        assert abs(got - want) <= 1e-6 * max(1.0, abs(want)), (nb, got, want)


def test_zero_crossing_threshold_and_edge_frames(ops):
    """|x| <= 1e-10 counts as zero (librosa's threshold) -- exercised with float32 values around 1e-10 -- and the frames
    that reach over a clip end see edge padding (the other rows: zeros)."""
    rng = np.random.default_rng(23)
    t = np.float32(1e-10)
    vals = np.array([t, np.nextafter(t, np.float32(0)), np.nextafter(t, np.float32(1)), -t, -np.nextafter(t, np.float32(0)),
                     -np.nextafter(t, np.float32(1)), 0.0, -0.0, 1e-3, -1e-3], dtype=np.float32)
    y = vals[rng.integers(0, len(vals), 5000)]
    y[0], y[-1] = -1e-3, -1e-3                                   # edge padding repeats a negative sample
    for fl, hop, center in ((256, 64, True), (250, 100, True), (512, 512, False), (64, 1, True)):
        got = ops.frame_stats(ops.to_device_f32(y[None, :]), fl, hop, center, mask=256).cpu().numpy()[0, 8]
        want = O.zero_crossing_rate(y.astype(np.float64), fl, hop, center)
        assert np.array_equal(np.round(got[:len(want)] * fl), np.round(want * fl)), (fl, hop, center)
