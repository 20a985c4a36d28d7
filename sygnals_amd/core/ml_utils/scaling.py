"""Device-backed mirror of sygnals/core/ml_utils/scaling.py:49-175.

Same signatures and return values: (scaled float64 array, fitted scikit-learn scaler).  The column statistics of the
fit and the transform run on the device (syg_col_stats_f32 / syg_col_quantiles_f32 / syg_affine_cols_f32); the scaler
that comes back is a genuine scikit-learn object with its fitted attributes set, so `inverse_transform`, pickling and
later `transform` calls behave as with the reference.  `scale_features_device` is the form that never leaves HBM.
"""
from __future__ import annotations

import logging
from typing import Any, Dict, Optional, Tuple

import numpy as np

from ... import ops

logger = logging.getLogger(__name__)
try:
    from sklearn.preprocessing import MinMaxScaler, RobustScaler, StandardScaler
    _SKLEARN_AVAILABLE = True
except ImportError:                                           # scaling.py:17-45: same message at call time
    _SKLEARN_AVAILABLE = False
    MinMaxScaler = RobustScaler = StandardScaler = None


def _handle_zeros(scale: np.ndarray, constant_mask: Optional[np.ndarray] = None) -> np.ndarray:
    """sklearn.preprocessing._data._handle_zeros_in_scale: a (near-)zero scale becomes 1."""
    scale = np.array(scale, dtype=np.float64)
    mask = scale < 10 * np.finfo(np.float64).eps if constant_mask is None else constant_mask
    scale[mask] = 1.0
    return scale


def _fit(xd, scaler):
    """Set the fitted attributes of `scaler` from device statistics of xd [n, F]; returns (sub, mul, add) of its
    transform (x - sub) * mul + add."""
    n, F = xd.shape
    zeros, ones = np.zeros(F), np.ones(F)
    if isinstance(scaler, StandardScaler):
        cnt, mean, var = (v.cpu().numpy() for v in ops.col_stats(xd)[:3])
        seen = cnt.astype(np.int64)
        scaler.n_samples_seen_ = int(seen[0]) if (seen == seen[0]).all() else seen
        # scikit-learn keeps mean_ and var_ unless BOTH with_mean and with_std are off; scale_ only with with_std
        keep = scaler.with_mean or scaler.with_std
        scaler.mean_ = mean if keep else None
        scaler.var_ = var if keep else None
        if scaler.with_std:
            eps = np.finfo(np.float64).eps
            constant = var <= cnt * eps * var + (cnt * mean * eps) ** 2       # sklearn.utils.extmath._is_constant_feature
            scaler.scale_ = _handle_zeros(np.sqrt(var), constant)
        else:
            scaler.scale_ = None
    elif isinstance(scaler, MinMaxScaler):
        st = ops.col_stats(xd).cpu().numpy()
        lo, hi = scaler.feature_range
        if lo >= hi:
            raise ValueError(f"Minimum of desired feature range must be smaller than maximum. Got {scaler.feature_range}.")
        scaler.n_samples_seen_ = n
        scaler.data_min_, scaler.data_max_ = st[3], st[4]
        scaler.data_range_ = st[4] - st[3]
        scaler.scale_ = (hi - lo) / _handle_zeros(scaler.data_range_)
        scaler.min_ = lo - scaler.data_min_ * scaler.scale_
    elif isinstance(scaler, RobustScaler):
        qlo, qhi = scaler.quantile_range
        if not 0 <= qlo <= qhi <= 100:
            raise ValueError(f"Invalid quantile range: {scaler.quantile_range}")
        q = ops.col_quantiles(xd, [0.5, qlo / 100.0, qhi / 100.0]).cpu().numpy()
        scaler.center_ = q[0] if scaler.with_centering else None
        if scaler.with_scaling:
            scale = _handle_zeros(q[2] - q[1])
            if getattr(scaler, "unit_variance", False):
                from scipy.stats import norm
                scale = scale / (norm.ppf(qhi / 100.0) - norm.ppf(qlo / 100.0))
            scaler.scale_ = scale
        else:
            scaler.scale_ = None
    else:
        raise ValueError(f"Unsupported scaler: {type(scaler).__name__}")
    scaler.n_features_in_ = F
    return _affine_of(scaler, F)


def _affine_of(scaler, F):
    """(sub, mul, add) with transform(x) = (x - sub) * mul + add for a FITTED scaler."""
    zeros, ones = np.zeros(F), np.ones(F)
    if isinstance(scaler, StandardScaler):
        if not hasattr(scaler, "scale_"):
            raise ValueError("Provided `scaler_instance` does not appear to be fitted.")
        return (scaler.mean_ if scaler.with_mean else zeros, 1.0 / scaler.scale_ if scaler.with_std else ones, zeros)
    if isinstance(scaler, MinMaxScaler):
        if not hasattr(scaler, "min_"):
            raise ValueError("Provided `scaler_instance` does not appear to be fitted.")
        return zeros, scaler.scale_, scaler.min_
    if isinstance(scaler, RobustScaler):
        if not hasattr(scaler, "scale_"):
            raise ValueError("Provided `scaler_instance` does not appear to be fitted.")
        return (scaler.center_ if scaler.with_centering else zeros, 1.0 / scaler.scale_ if scaler.with_scaling else ones,
                zeros)
    raise ValueError(f"Unsupported scaler: {type(scaler).__name__}")


def scale_features_device(xd, scaler_type: str = "standard", scaler_params: Optional[Dict[str, Any]] = None,
                          fit: bool = True, scaler_instance=None):
    """apply_scaling on a float32 [n, F] DEVICE tensor -> (scaled device tensor, fitted scikit-learn scaler)."""
    if not _SKLEARN_AVAILABLE:
        raise ImportError("scikit-learn package is required for feature scaling. Please install it (`pip install scikit-learn`).")
    scaler_params = scaler_params or {}
    if fit:
        if scaler_instance is not None:
            logger.warning("`scaler_instance` provided but `fit=True`. Ignoring provided instance and fitting a new scaler.")
        if scaler_type == "standard":
            scaler = StandardScaler(**scaler_params)
        elif scaler_type == "minmax":
            scaler = MinMaxScaler(**scaler_params)
        elif scaler_type == "robust":
            scaler = RobustScaler(**scaler_params)
        else:
            raise ValueError(f"Unsupported scaler_type: '{scaler_type}'. Choose 'standard', 'minmax', or 'robust'.")
        sub, mul, add = _fit(xd, scaler)
    else:
        if scaler_instance is None:
            raise ValueError("`scaler_instance` must be provided when `fit=False`.")
        scaler = scaler_instance
        sub, mul, add = _affine_of(scaler, xd.shape[1])
        if len(np.atleast_1d(mul)) != xd.shape[1]:
            raise ValueError(f"X has {xd.shape[1]} features, but {type(scaler).__name__} is expecting "
                             f"{len(np.atleast_1d(mul))} features as input.")
    out = ops.affine_cols(xd, sub, mul, add)
    if isinstance(scaler, MinMaxScaler) and getattr(scaler, "clip", False):
        out = out.clamp(scaler.feature_range[0], scaler.feature_range[1])
    return out, scaler


def _pivot(features: np.ndarray) -> np.ndarray:
    """A value near every column's data (median of up to the first 64 finite entries), float64: subtracted on the host
    BEFORE the float32 cast so that a column with a large offset relative to its spread (mean 1e6, std 1) keeps its
    precision on the device; the fitted attributes are shifted back afterwards."""
    head = np.asarray(features[:64], dtype=np.float64)
    with np.errstate(all="ignore"):
        piv = np.nanmedian(np.where(np.isfinite(head), head, np.nan), axis=0)
    return np.where(np.isfinite(piv), piv, 0.0)


def _shift_fitted(scaler, piv: np.ndarray) -> None:
    """Attributes of a scaler fitted on (X - piv) -> the scaler of X (scale-type attributes are shift-invariant)."""
    if isinstance(scaler, StandardScaler):
        if scaler.mean_ is not None:
            scaler.mean_ = scaler.mean_ + piv
    elif isinstance(scaler, MinMaxScaler):
        scaler.data_min_ = scaler.data_min_ + piv
        scaler.data_max_ = scaler.data_max_ + piv
        scaler.min_ = scaler.feature_range[0] - scaler.data_min_ * scaler.scale_
    elif isinstance(scaler, RobustScaler):
        if scaler.center_ is not None:
            scaler.center_ = scaler.center_ + piv


def apply_scaling(features, scaler_type: str = "standard", scaler_params: Optional[Dict[str, Any]] = None,
                  fit: bool = True, scaler_instance=None) -> Tuple[np.ndarray, Any]:
    features = np.asarray(features)
    if features.ndim != 2:
        if features.ndim == 1:
            features = features.reshape(-1, 1)
        else:
            raise ValueError(f"Input features must be 1D or 2D (samples/frames x features), got shape {features.shape}")
    if fit and features.shape[0] > 0:
        piv = _pivot(features)
        out, scaler = scale_features_device(ops.to_device_f32(features.astype(np.float64) - piv), scaler_type,
                                            scaler_params, True, scaler_instance)
        _shift_fitted(scaler, piv)
        res = out.cpu().numpy().astype(np.float64)
        # a scaler that does not centre (with_mean / with_centering off) scales x itself, not x - piv: put piv / scale back
        uncentred = (isinstance(scaler, StandardScaler) and not scaler.with_mean) or \
                    (isinstance(scaler, RobustScaler) and not scaler.with_centering)
        if uncentred:
            sc = getattr(scaler, "scale_", None)
            res += piv / sc if sc is not None else piv
        return res, scaler
    if not fit and scaler_instance is not None and features.shape[0] > 0:
        if not _SKLEARN_AVAILABLE:
            raise ImportError("scikit-learn package is required for feature scaling. Please install it (`pip install scikit-learn`).")
        # transform with a given scaler: its own centre is the pivot (float64, on the host), the device scales
        sub, mul, add = _affine_of(scaler_instance, features.shape[1])
        if isinstance(scaler_instance, MinMaxScaler):      # x scale_ + min_ = (x - data_min_) scale_ + range minimum
            sub, add = scaler_instance.data_min_, np.full(features.shape[1], float(scaler_instance.feature_range[0]))
        if len(np.atleast_1d(mul)) != features.shape[1]:
            raise ValueError(f"X has {features.shape[1]} features, but {type(scaler_instance).__name__} is expecting "
                             f"{len(np.atleast_1d(mul))} features as input.")
        xd = ops.to_device_f32(features.astype(np.float64) - np.asarray(sub, dtype=np.float64))
        out = ops.affine_cols(xd, np.zeros(features.shape[1]), mul, add)
        if isinstance(scaler_instance, MinMaxScaler) and getattr(scaler_instance, "clip", False):
            out = out.clamp(scaler_instance.feature_range[0], scaler_instance.feature_range[1])
        return out.cpu().numpy().astype(np.float64), scaler_instance
    out, scaler = scale_features_device(ops.to_device_f32(features), scaler_type, scaler_params, fit, scaler_instance)
    return out.cpu().numpy().astype(np.float64), scaler


def standard_scale(features, with_mean: bool = True, with_std: bool = True):
    return apply_scaling(features, "standard", {"with_mean": with_mean, "with_std": with_std})


def minmax_scale(features, feature_range: Tuple[float, float] = (0, 1)):
    return apply_scaling(features, "minmax", {"feature_range": feature_range})


def robust_scale(features, with_centering: bool = True, with_scaling: bool = True,
                 quantile_range: Tuple[float, float] = (25.0, 75.0)):
    return apply_scaling(features, "robust", {"with_centering": with_centering, "with_scaling": with_scaling,
                                              "quantile_range": quantile_range})
