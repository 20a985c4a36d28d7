"""Helpers shared by the GPU parity tests."""
import numpy as np


def peak_rel(a, b):
    """max|a-b| / max|b|  (the parity metric of SURVEY section 8d)."""
    a = np.asarray(a); b = np.asarray(b)
    den = np.max(np.abs(b))
    return float(np.max(np.abs(a - b)) / (den if den > 0 else 1.0))


def assert_parity(a, b, tol=1e-5, what=""):
    """fp32 device output `a` vs float64 oracle `b`: max|a-b| <= tol*max|b| and
    allclose(rtol=tol, atol=tol*max|b|)."""
    a = np.asarray(a, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, f"{what}: shape {a.shape} vs {b.shape}"
    assert np.isfinite(a).all(), f"{what}: non-finite device output"
    pk = np.max(np.abs(b)) if b.size else 0.0
    err = np.max(np.abs(a - b)) if b.size else 0.0
    assert err <= tol * pk + 1e-30, f"{what}: peak-relative error {err / max(pk, 1e-300):.3e} > {tol:g}"
    assert np.allclose(a, b, rtol=tol, atol=tol * pk + 1e-30), f"{what}: allclose failed"
