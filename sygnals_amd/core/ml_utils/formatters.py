"""Device-backed mirror of sygnals/core/ml_utils/formatters.py:51-334 (per-segment vectors, sequences, images).

`format_feature_sequences` is pure data movement (stack, cut, pad): it accepts NumPy arrays or device tensors and
returns the same kind, so features can go from the extraction kernels to a training loop without leaving HBM.
`format_features_as_image` resizes (scipy.ndimage.zoom order 0 / 1, mode='nearest') and normalises on the device.
`format_feature_vectors_per_segment` aggregates the frames of every segment with the NaN-aware column statistics
(mean / std / min / max: syg_col_stats_f32; median: syg_col_quantiles_f32) of the stacked [frames, features] matrix.
"""
from __future__ import annotations

import logging
import warnings
from typing import Any, Dict, List, Optional, Tuple, Union

import numpy as np
import torch

from ... import ops
from ..._lib import SygnalsHipError

logger = logging.getLogger(__name__)
_EPSILON = np.finfo(np.float64).eps


AGGREGATIONS = ("mean", "std", "median", "min", "max")          # formatters.py:39-45 (NaN-aware)


def format_feature_vectors_per_segment(features_dict: Dict[str, object], segment_indices: List[Tuple[int, int]],
                                       aggregation: Union[str, Dict[str, str]] = "mean",
                                       output_format: str = "dataframe", segment_labels: Optional[List[Any]] = None):
    import pandas as pd
    if not features_dict:
        logger.warning("Input features_dict is empty. Returning empty result.")
        return pd.DataFrame() if output_format == "dataframe" else np.empty((0, 0), dtype=np.float64)
    names = list(features_dict.keys())
    counts = [len(v) for v in features_dict.values()]
    n = counts[0]
    if not all(c == n for c in counts):
        raise ValueError(f"All feature arrays in features_dict must have the same length. Found lengths: {counts}")
    if not segment_indices:
        logger.warning("No segment indices provided. Returning empty result.")
        return pd.DataFrame() if output_format == "dataframe" else np.empty((0, len(names)), dtype=np.float64)
    if segment_labels is not None and len(segment_labels) != len(segment_indices):
        raise ValueError("Length of segment_labels must match length of segment_indices.")
    if isinstance(aggregation, str):
        if aggregation not in AGGREGATIONS:
            raise ValueError(f"Unknown global aggregation function: '{aggregation}'. Available: {list(AGGREGATIONS)}")
        how = {k: aggregation for k in names}
    elif isinstance(aggregation, dict):
        how = {}
        for k in names:
            m = aggregation.get(k, "mean")
            if m not in AGGREGATIONS:
                raise ValueError(f"Unknown aggregation function '{m}' for feature '{k}'. Available: {list(AGGREGATIONS)}")
            how[k] = m
    else:
        raise TypeError("aggregation must be a string or a dictionary.")
    if output_format not in ("dataframe", "numpy"):
        raise ValueError(f"Unknown output_format: '{output_format}'. Choose 'dataframe' or 'numpy'.")
    dev = ops.require_gpu()
    X = torch.stack([torch.as_tensor(np.asarray(features_dict[k].cpu() if isinstance(features_dict[k], torch.Tensor)
                                                else features_dict[k], dtype=np.float32)) for k in names], dim=1).to(dev)
    out = np.full((len(segment_indices), len(names)), np.nan, dtype=np.float64)
    need_median = any(m == "median" for m in how.values())
    row_of = {"mean": 1, "min": 3, "max": 4}
    for i, (a, b) in enumerate(segment_indices):
        if not (0 <= a < n and a < b and b <= n):
            msg = f"Invalid segment indices ({a}, {b}) for num_frames={n}. Skipping segment {i}."
            logger.warning(msg)
            warnings.warn(msg, UserWarning, stacklevel=2)
            continue                                              # the row stays NaN
        seg = X[a:b]
        st = ops.col_stats(seg).cpu().numpy()                     # count, mean, population variance, min, max
        med = ops.col_quantiles(seg, [0.5]).cpu().numpy()[0] if need_median else None
        for j, k in enumerate(names):
            m = how[k]
            out[i, j] = med[j] if m == "median" else (np.sqrt(st[2, j]) if m == "std" else st[row_of[m], j])
    if output_format == "numpy":
        return out
    index = segment_labels if segment_labels is not None else pd.RangeIndex(len(segment_indices), name="segment_index")
    return pd.DataFrame(out, columns=names, index=index)


def format_feature_sequences(features_dict: Dict[str, object], max_sequence_length: Optional[int] = None,
                             padding_value: float = 0.0, truncation_strategy: str = "post",
                             output_format: str = "list_of_arrays"):
    on_device = any(isinstance(v, torch.Tensor) for v in features_dict.values()) if features_dict else False

    def empty(shape):
        return torch.empty(shape, dtype=torch.float32, device=ops.require_gpu()) if on_device else np.empty(shape, dtype=np.float64)

    if not features_dict:
        logger.warning("Input features_dict is empty. Returning empty result.")
        return [] if output_format == "list_of_arrays" else empty((1, max_sequence_length or 0, 0))
    names = list(features_dict.keys())
    counts = [len(v) for v in features_dict.values()]
    if not all(c == counts[0] for c in counts):
        raise ValueError(f"All feature arrays in features_dict must have the same length. Found lengths: {counts}")
    n = counts[0]
    if n == 0:
        logger.warning("Input features have zero length (no frames). Returning empty result.")
        return [] if output_format == "list_of_arrays" else empty((1, max_sequence_length or 0, len(names)))
    if on_device:
        dev = ops.require_gpu()
        seq = torch.stack([torch.as_tensor(features_dict[k], dtype=torch.float32, device=dev) for k in names], dim=1)
    else:
        seq = np.stack([np.asarray(features_dict[k]) for k in names], axis=1).astype(np.float64)
    if max_sequence_length is not None and max_sequence_length > 0:
        if n > max_sequence_length:
            if truncation_strategy == "post":
                seq = seq[:max_sequence_length]
            elif truncation_strategy == "pre":
                seq = seq[n - max_sequence_length:]
            else:
                raise ValueError(f"Unknown truncation_strategy: '{truncation_strategy}'. Choose 'pre' or 'post'.")
        elif n < max_sequence_length:
            pad = max_sequence_length - n
            if on_device:
                seq = torch.cat([seq, torch.full((pad, seq.shape[1]), float(padding_value), dtype=seq.dtype,
                                                 device=seq.device)], dim=0)
            else:
                seq = np.pad(seq, ((0, pad), (0, 0)), mode="constant", constant_values=padding_value)
    if output_format == "list_of_arrays":
        return [seq]
    if output_format == "padded_array":
        return seq[None] if on_device else np.expand_dims(seq, axis=0)
    raise ValueError(f"Unknown output_format: '{output_format}'. Choose 'list_of_arrays' or 'padded_array'.")


def image_device(fmap: torch.Tensor, output_shape: Optional[Tuple[int, int]] = None, resize_order: int = 1,
                 normalize: bool = True) -> torch.Tensor:
    """format_features_as_image on a float32 [H, W] DEVICE tensor -> float32 device tensor."""
    img = fmap
    if output_shape is not None:
        if not isinstance(output_shape, tuple) or len(output_shape) != 2 or \
                not all(isinstance(d, (int, np.integer)) and d > 0 for d in output_shape):
            raise ValueError("output_shape must be a tuple of two positive integers (height, width).")
        if tuple(img.shape) != tuple(output_shape):
            if resize_order not in (0, 1):
                raise SygnalsHipError(f"format_features_as_image: resize_order={resize_order} is not offloaded "
                                      "(0 = nearest and 1 = linear run on the device)")
            img = ops.zoom2d(img, output_shape, resize_order)
    if normalize:
        st = ops.col_stats(img.reshape(-1, 1)).cpu().numpy()          # min and max of the whole image
        lo, hi = float(st[3, 0]), float(st[4, 0])
        rng = hi - lo
        if not rng >= _EPSILON:                                        # constant image (formatters.py:324-328)
            return torch.zeros_like(img)
        flat = ops.affine_cols(img.reshape(-1, 1), [lo], [1.0 / rng], [0.0])
        img = flat.reshape(img.shape)
    return img


def format_features_as_image(feature_map, output_shape: Optional[Tuple[int, int]] = None, resize_order: int = 1,
                             normalize: bool = True) -> np.ndarray:
    feature_map = np.asarray(feature_map)
    if feature_map.ndim != 2:
        raise ValueError("Input feature_map must be a 2D array.")
    if feature_map.size == 0:
        logger.warning("Input feature_map is empty. Returning empty array.")
        return np.empty(output_shape or (0, 0), dtype=np.float64)
    out = image_device(ops.to_device_f32(feature_map), output_shape, resize_order, normalize)
    return out.cpu().numpy().astype(np.float64)
