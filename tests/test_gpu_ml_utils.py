"""GPU parity of the ML formatting helpers: scalers (fit + transform on the device, a genuine
scikit-learn scaler back), sequence stacking and the image formatter -- sygnals_amd.core.ml_utils -> ops ->
libsygnals_hip.so (syg_col_stats_f32, syg_col_quantiles_f32, syg_affine_cols_f32, syg_zoom_f32).
Pinned on tests/golden/ref_ml.npz (outputs of the reference's functions), then against scikit-learn / SciPy directly
(the reference's own arithmetic) on random shapes with NaNs.  Tolerance 1e-5 of the output's peak."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")
from oracle import cpu_ref as O
from tests.gpu_util import assert_parity
from tests.test_oracle_golden import SCALER_CASES

TOL = 1e-5
G = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def gm():
    from sygnals_amd import ops
    ops.require_gpu()
    return np.load(os.path.join(G, "ref_ml.npz"))


@pytest.mark.parametrize("tag,kind,kw", SCALER_CASES)
def test_scalers_vs_reference_golden(gm, tag, kind, kw):
    from sygnals_amd.core.ml_utils import apply_scaling
    X = gm["X"]
    out, scaler = apply_scaling(X, scaler_type=kind, scaler_params=kw)
    assert out.dtype == np.float64 and out.shape == X.shape
    want = gm[f"scale_{tag}"]
    for c in range(X.shape[1]):                       # per column: the features live on very different scales
        assert_parity(out[:, c], want[:, c], TOL, f"{tag} column {c}")
    for a in ("mean_", "var_", "scale_", "min_", "center_"):
        if f"scale_{tag}_{a}" in gm.files:
            np.testing.assert_allclose(getattr(scaler, a), gm[f"scale_{tag}_{a}"], rtol=2e-6, atol=1e-9, err_msg=a)
    # the scaler that comes back is a working scikit-learn object: its own transform / inverse agree with the device
    np.testing.assert_allclose(scaler.transform(X), want, rtol=1e-5, atol=1e-5 * np.abs(want).max())
    np.testing.assert_allclose(scaler.inverse_transform(out), X, rtol=1e-4, atol=1e-4 * np.abs(X).max())
    # and a fitted instance can be applied without fitting again (scaling.py:121-141)
    out2, same = apply_scaling(X[:50], fit=False, scaler_instance=scaler)
    assert same is scaler
    np.testing.assert_allclose(out2, out[:50], rtol=0, atol=1e-6 * max(1.0, np.abs(out).max()))


def test_scaling_helpers_errors_and_1d(gm):
    from sygnals_amd.core.ml_utils import apply_scaling, minmax_scale, robust_scale, standard_scale
    out, _ = standard_scale(gm["X"][:, 1])
    assert out.shape == (300, 1)
    assert_parity(out, gm["scale_1d"], TOL, "1-D input")
    assert_parity(minmax_scale(gm["X"], (-1, 1))[0], gm["scale_mm_m11"], TOL, "minmax helper")
    assert_parity(robust_scale(gm["X"], quantile_range=(10.0, 90.0))[0][:, 0], gm["scale_rob_1090"][:, 0], TOL, "robust helper")
    with pytest.raises(ValueError, match="Unsupported scaler_type"):
        apply_scaling(gm["X"], scaler_type="log")
    with pytest.raises(ValueError, match="must be provided when `fit=False`"):
        apply_scaling(gm["X"], fit=False)
    with pytest.raises(ValueError, match="1D or 2D"):
        apply_scaling(np.zeros((2, 2, 2)))
    from sklearn.preprocessing import StandardScaler
    with pytest.raises(ValueError, match="does not appear to be fitted"):
        apply_scaling(gm["X"], fit=False, scaler_instance=StandardScaler())


def test_scalers_with_nans_and_random_shapes_vs_sklearn():
    from sygnals_amd.core.ml_utils import apply_scaling
    rng = np.random.default_rng(3)
    for n, F in ((1, 1), (2, 3), (65, 130), (1000, 7), (5000, 33), (32768, 2)):
        X = (rng.normal(0, 1, (n, F)) * rng.uniform(0.1, 50, F) + rng.uniform(-20, 20, F)).astype(np.float32).astype(np.float64)
        if n > 10:
            X[rng.integers(0, n, n // 10), rng.integers(0, F, n // 10)] = np.nan      # NaN-padded frames (odd frame lengths)
        for kind in ("standard", "minmax", "robust"):
            out, _ = apply_scaling(X, kind)
            want, _ = O.apply_scaling(X, kind)
            assert np.array_equal(np.isnan(out), np.isnan(want))
            for c in range(F):
                m = ~np.isnan(want[:, c])
                if m.any():
                    pk = max(np.abs(want[m, c]).max(), 1e-30)
                    assert np.abs(out[m, c] - want[m, c]).max() <= 2e-5 * pk + 1e-6, (kind, n, F, c)


def test_sequences_and_images_vs_reference_golden(gm):
    from sygnals_amd import ops
    from sygnals_amd.core.ml_utils import format_feature_sequences, format_features_as_image
    feats = {f"f{i}": gm["X"][:40, i] for i in range(4)}
    assert np.array_equal(format_feature_sequences(feats)[0], gm["seq_list"])
    assert np.array_equal(format_feature_sequences(feats, 64, -1.0, output_format="padded_array"), gm["seq_pad64"])
    assert np.array_equal(format_feature_sequences(feats, 16, truncation_strategy="pre", output_format="padded_array"),
                          gm["seq_cut16_pre"])
    dev = {k: ops.to_device_f32(v) for k, v in feats.items()}                 # device tensors in -> device tensor out
    seq = format_feature_sequences(dev, 64, -1.0, output_format="padded_array")
    assert seq.is_cuda and np.array_equal(seq.cpu().numpy(), gm["seq_pad64"].astype(np.float32))
    with pytest.raises(ValueError, match="same length"):
        format_feature_sequences({"a": np.zeros(3), "b": np.zeros(4)})
    assert format_feature_sequences({}) == []
    M = gm["img_in"]
    assert_parity(format_features_as_image(M), gm["img_norm"], TOL, "normalise")
    assert_parity(format_features_as_image(M, output_shape=(64, 64)), gm["img_64x64"], TOL, "64x64 linear")
    assert_parity(format_features_as_image(M, output_shape=(20, 200), normalize=False), gm["img_20x200_nonorm"], TOL, "20x200")
    assert_parity(format_features_as_image(M, output_shape=(128, 32), resize_order=0), gm["img_128x32_nearest"], TOL, "nearest")
    assert np.array_equal(format_features_as_image(np.full((5, 7), 2.5)), gm["img_const"])
    with pytest.raises(ValueError, match="must be a 2D"):
        format_features_as_image(np.zeros(5))
    with pytest.raises(ValueError, match="two positive integers"):
        format_features_as_image(M, output_shape=(0, 5))


def test_zoom_random_shapes_vs_scipy():
    from sygnals_amd.core.ml_utils import format_features_as_image
    rng = np.random.default_rng(9)
    for _ in range(25):
        h, w, h2, w2 = (int(v) for v in rng.integers(1, 200, 4))
        M = rng.normal(0, 1, (h, w)).astype(np.float32).astype(np.float64)
        for order in (0, 1):
            got = format_features_as_image(M, output_shape=(h2, w2), resize_order=order, normalize=False)
            want = O.format_features_as_image(M, (h2, w2), order, False)
            assert got.shape == (h2, w2)
            if order == 1:
                assert_parity(got, want, TOL, f"{h}x{w}->{h2}x{w2}")
            else:       # nearest: identical picks except where the sampling point is within rounding of a half-way tie
                assert (np.abs(got - want) > 1e-6).mean() < 0.02, (h, w, h2, w2)


# ---------------------------------------------------------------- the reference's own test cases, same inputs
# (tests/test_ml_utils.py:36-80 fixtures, :83-150 and :233-345 assertions); tolerances are those of an fp32 device path
def _ref_fixtures():
    rng = np.random.default_rng(123)
    arr = np.hstack([rng.normal(loc=10, scale=2, size=(50, 1)), rng.normal(loc=0, scale=0.1, size=(50, 1)),
                     rng.uniform(low=-5, high=5, size=(50, 1))]).astype(np.float64)
    one_d = np.random.default_rng(456).normal(loc=5, scale=3, size=100).astype(np.float64)
    rng = np.random.default_rng(789)
    feats = {"rms": rng.random(50) * 0.5, "zcr": rng.random(50) * 0.1, "centroid": rng.random(50) * 1000 + 500}
    fmap = np.random.default_rng(101).random((64, 80)) * 10
    return arr, one_d, feats, fmap


def test_reference_scaling_cases(gm):
    from sklearn.preprocessing import MinMaxScaler, RobustScaler, StandardScaler
    from sygnals_amd.core.ml_utils import apply_scaling
    arr, one_d, _, _ = _ref_fixtures()
    out, sc = apply_scaling(arr, scaler_type="standard")
    assert isinstance(sc, StandardScaler) and out.shape == arr.shape and out.dtype == np.float64
    np.testing.assert_allclose(out.mean(axis=0), 0.0, atol=1e-6)
    np.testing.assert_allclose(out.std(axis=0), 1.0, atol=1e-6)
    out, sc = apply_scaling(arr, scaler_type="minmax")
    assert isinstance(sc, MinMaxScaler)
    np.testing.assert_allclose(out.min(axis=0), 0.0, atol=1e-6)
    np.testing.assert_allclose(out.max(axis=0), 1.0, atol=1e-6)
    out, sc = apply_scaling(arr, scaler_type="robust")
    assert isinstance(sc, RobustScaler)
    np.testing.assert_allclose(np.median(out, axis=0), 0.0, atol=1e-6)
    out, sc = apply_scaling(one_d, scaler_type="standard")
    assert out.shape == (100, 1)
    np.testing.assert_allclose([out.mean(), out.std()], [0.0, 1.0], atol=1e-6)
    fitted = StandardScaler().fit(arr[:25])
    out, used = apply_scaling(arr[25:], fit=False, scaler_instance=fitted)
    assert used is fitted and out.shape == (25, 3)
    np.testing.assert_allclose(used.inverse_transform(out), arr[25:], atol=1e-5)
    with pytest.raises(ValueError):
        apply_scaling(arr, scaler_type="invalid_scaler")
    with pytest.raises(ValueError):
        apply_scaling(arr, fit=False, scaler_instance=None)
    with pytest.raises(ValueError):
        apply_scaling(arr, fit=False, scaler_instance=StandardScaler())


def test_reference_formatter_cases(gm):
    from sygnals_amd.core.ml_utils import format_feature_sequences, format_features_as_image
    _, _, feats, fmap = _ref_fixtures()
    res = format_feature_sequences(feats, output_format="list_of_arrays")
    assert isinstance(res, list) and len(res) == 1 and res[0].shape == (50, 3) and res[0].dtype == np.float64
    pad = format_feature_sequences(feats, max_sequence_length=60, padding_value=-1.0, output_format="padded_array")
    assert pad.shape == (1, 60, 3) and np.all(pad[0, 50:, :] == -1.0)
    np.testing.assert_allclose(pad[0, :50, list(feats).index("rms")], feats["rms"])
    post = format_feature_sequences(feats, max_sequence_length=40, truncation_strategy="post", output_format="padded_array")
    np.testing.assert_allclose(post[0, :, 0], feats["rms"][:40])
    pre = format_feature_sequences(feats, max_sequence_length=40, truncation_strategy="pre", output_format="padded_array")
    np.testing.assert_allclose(pre[0, :, 0], feats["rms"][10:])
    bad = dict(feats); bad["rms"] = bad["rms"][:-1]
    with pytest.raises(ValueError, match="All feature arrays.*must have the same length"):
        format_feature_sequences(bad)
    with pytest.raises(ValueError, match="Unknown truncation_strategy"):
        format_feature_sequences(feats, max_sequence_length=40, truncation_strategy="middle")
    same = format_features_as_image(fmap, normalize=False)
    assert same.shape == fmap.shape and same.dtype == np.float64
    np.testing.assert_allclose(same, fmap, rtol=1e-6)                      # (fp32 on the device)
    assert format_features_as_image(fmap, output_shape=(32, 40), normalize=False).shape == (32, 40)
    for shape in (None, (100, 100)):
        img = format_features_as_image(fmap, output_shape=shape, normalize=True)
        assert img.shape == (shape or fmap.shape) and img.min() >= 0.0 and img.max() <= 1.0
        assert np.isclose(img.min(), 0.0) and np.isclose(img.max(), 1.0)
    with pytest.raises(ValueError, match="Input feature_map must be a 2D array"):
        format_features_as_image(np.random.rand(10))
    with pytest.raises(ValueError, match="output_shape must be a tuple of two positive integers"):
        format_features_as_image(fmap, output_shape=(10,))
    with pytest.raises(ValueError, match="output_shape must be a tuple of two positive integers"):
        format_features_as_image(fmap, output_shape=(-10, 10))


def test_segment_vectors_vs_reference_golden_and_cases(gm):
    """format_feature_vectors_per_segment: golden vectors, then the reference's own cases (tests/test_ml_utils.py:153-230)."""
    import pandas as pd
    from sygnals_amd.core.ml_utils import format_feature_vectors_per_segment
    feats = {f"f{i}": gm["X"][:40, i] for i in range(4)}
    segs = [(0, 15), (15, 30), (30, 40), (5, 6)]
    for agg in ("mean", "std", "median", "min", "max"):
        got = format_feature_vectors_per_segment(feats, segs, aggregation=agg, output_format="numpy")
        for c in range(4):
            assert_parity(got[:, c], gm[f"vec_{agg}"][:, c], 2e-5 if agg == "std" else TOL, f"{agg} column {c}")
    got = format_feature_vectors_per_segment(feats, segs, aggregation={"f0": "max", "f1": "min", "f2": "std"}, output_format="numpy")
    for c in range(4):
        assert_parity(got[:, c], gm["vec_mixed"][:, c], 2e-5, f"mixed column {c}")
    # NaN frames are left out of every aggregate; an all-NaN segment gives NaN
    f2 = {k: v.copy() for k, v in feats.items()}
    f2["f0"][3:9] = np.nan; f2["f1"][15:30] = np.nan
    got = format_feature_vectors_per_segment(f2, segs, aggregation="median", output_format="numpy")
    want = O.format_feature_vectors_per_segment(f2, segs, "median")
    assert np.array_equal(np.isnan(got), np.isnan(want)) and np.isnan(got[1, 1])
    np.testing.assert_allclose(got[~np.isnan(want)], want[~np.isnan(want)], rtol=1e-5)
    # the reference's cases
    _, _, rf, _ = _ref_fixtures()
    rsegs = [(0, 15), (15, 30), (30, 50)]
    df = format_feature_vectors_per_segment(rf, rsegs, aggregation="mean", output_format="dataframe")
    assert isinstance(df, pd.DataFrame) and df.shape == (3, 3) and list(df.columns) == list(rf) and df.index.name == "segment_index"
    assert np.isclose(df.loc[0, "rms"], np.mean(rf["rms"][0:15]), rtol=1e-6)
    arr = format_feature_vectors_per_segment(rf, rsegs, aggregation={"rms": "max", "zcr": "min", "centroid": "std"}, output_format="numpy")
    assert arr.dtype == np.float64 and arr.shape == (3, 3)
    assert np.isclose(arr[1, 0], rf["rms"][15:30].max(), rtol=1e-6) and np.isclose(arr[1, 1], rf["zcr"][15:30].min(), rtol=1e-6)
    assert np.isclose(arr[1, 2], np.std(rf["centroid"][15:30]), rtol=1e-5)
    assert list(format_feature_vectors_per_segment(rf, rsegs, segment_labels=["speech", "music", "noise"]).index) == ["speech", "music", "noise"]
    bad = dict(rf); bad["rms"] = bad["rms"][:-1]
    with pytest.raises(ValueError, match="All feature arrays.*must have the same length"):
        format_feature_vectors_per_segment(bad, rsegs)
    with pytest.warns(UserWarning, match=r"Invalid segment indices \(10, 60\)"):
        res = format_feature_vectors_per_segment(rf, [(0, 10), (10, 60)])
        assert np.isnan(res.iloc[1]).all()
    with pytest.raises(ValueError, match="Unknown global aggregation function"):
        format_feature_vectors_per_segment(rf, rsegs, aggregation="unknown")
    with pytest.raises(ValueError, match="Unknown aggregation function 'bad' for feature 'rms'"):
        format_feature_vectors_per_segment(rf, rsegs, aggregation={"rms": "bad"})
    with pytest.raises(ValueError, match="Length of segment_labels must match"):
        format_feature_vectors_per_segment(rf, rsegs, segment_labels=["a", "b"])


@pytest.mark.parametrize("kind", ["standard", "minmax", "robust"])
def test_scaling_keeps_precision_of_columns_with_a_large_offset(kind):
    """A column with mean 1e6 and spread 1 would lose ~6e-2 to the float32 cast; apply_scaling subtracts a per-column
    pivot on the host in float64 first and shifts the fitted attributes back: values and attributes match scikit-learn
    (what the reference calls, scaling.py:49-143), and the returned scaler inverts and re-applies like the reference's."""
    from sklearn.preprocessing import MinMaxScaler, RobustScaler, StandardScaler
    from sygnals_amd.core.ml_utils.scaling import apply_scaling
    rng = np.random.default_rng(3)
    X = np.stack([1e6 + rng.normal(0, 1.0, 500), -4e4 + rng.normal(0, 0.01, 500), rng.normal(0, 5.0, 500)], axis=1)
    ref = {"standard": StandardScaler, "minmax": MinMaxScaler, "robust": RobustScaler}[kind]()
    want = ref.fit_transform(X)
    out, sc = apply_scaling(X, kind)
    assert np.abs(out - want).max() <= 1e-5 * np.abs(want).max()
    for attr in ("mean_", "data_min_", "data_max_", "min_", "center_", "scale_"):
        if hasattr(ref, attr) and getattr(ref, attr) is not None:
            np.testing.assert_allclose(getattr(sc, attr), getattr(ref, attr), rtol=1e-6, atol=1e-9, err_msg=attr)
    np.testing.assert_allclose(sc.inverse_transform(out), X, rtol=1e-9, atol=1e-4)
    again, _ = apply_scaling(X[:50], kind, fit=False, scaler_instance=sc)
    assert np.abs(again - ref.transform(X[:50])).max() <= 1e-5 * np.abs(want).max()
