"""Device-backed mirror of sygnals/core/ml_utils: scaling.py and the sequence / image formatters."""
from .formatters import (format_feature_sequences, format_feature_vectors_per_segment,  # noqa: F401
                         format_features_as_image)
from .scaling import apply_scaling, minmax_scale, robust_scale, standard_scale  # noqa: F401
