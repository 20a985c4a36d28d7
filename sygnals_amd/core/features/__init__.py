from .manager import extract_features, extract_features_batch, FeatureExtractionError  # noqa: F401
