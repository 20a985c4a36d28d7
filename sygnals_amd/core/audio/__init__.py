"""Device-backed mirror of sygnals/core/audio (features only)."""
