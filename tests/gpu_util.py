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


EPS32 = 2.0 ** -24


def fft_floor(S):
    """fp32 rounding floor of a float32 FFT per frame: eps32 * ||S_t||_2 for the float64 magnitudes S [F, T].
    The error of an fp32 FFT bin scales with the frame's total energy, not with the bin's own magnitude, so a bin
    (or a mean of bins) far below the frame's peaks carries this ABSOLUTE uncertainty whatever the kernel does
    (measured on MI355X, tools/parity_diag.py: worst observed valley error 0.37 floors)."""
    return EPS32 * np.linalg.norm(np.asarray(S, dtype=np.float64), axis=0)


def assert_contrast_parity(cdb, C64, valley64, floor, tol=1e-5, what="contrast"):
    """Margin-qualified gate for the dB spectral contrast (SURVEY 8d's decision-margin device, as for the rolloff bin).

    The contrast takes log10 of the VALLEY, the mean of the smallest bins of a band.  A valley error of one fp32 FFT
    floor moves the dB value by (10 / ln 10) * floor / valley.  A cell is *sure* when half a floor of valley error stays
    inside the tolerance, i.e. valley >= 0.5 * M * floor with M = (10 / ln 10) / (tol * max|C|):
      * sure cells: |dB error| <= tol * max|C|                       (the north-star gate, unchanged)
      * other cells: |dB error| <= (10 / ln 10) * floor / valley     (one floor of valley error, propagated)
    Returns (number of unsure cells, number of cells).  cdb, C64, valley64: [R, T]; floor: [T]."""
    cdb = np.asarray(cdb, dtype=np.float64)
    assert cdb.shape == C64.shape and np.isfinite(cdb).all(), what
    cmax = np.max(np.abs(C64))
    M = (10.0 / np.log(10.0)) / (tol * cmax)
    v = np.maximum(valley64, 1e-300)
    sure = v >= 0.5 * M * floor[None, :]
    err = np.abs(cdb - C64)
    assert (err[sure] <= tol * cmax).all(), \
        f"{what}: sure cells off by {err[sure].max() / cmax:.2e} peak-relative (> {tol:g})"
    bound = (10.0 / np.log(10.0)) * floor[None, :] / v
    assert (err[~sure] <= np.maximum(bound[~sure], tol * cmax)).all(), \
        f"{what}: a below-margin cell exceeds the propagated one-floor bound (worst ratio {(err[~sure] / bound[~sure]).max():.2f})"
    return int((~sure).sum()), int(sure.size)


def assert_flatness_parity(flat, ref, S, tol=1e-5, what="flatness"):
    """Margin-qualified gate for the spectral flatness exp(mean log(m + eps)) / mean m, a LOG-domain statistic: an error of
    one fp32 FFT floor on bin k moves log m_k by floor / m_k, so a frame with a bin near a spectral null (|X| of the order of
    the floor: one in a few thousand noise frames has one) carries the relative uncertainty u_t = mean_k(floor_t / (m_k +
    floor_t)) whatever the kernel does.  The floor is a conservative figure (fft_floor: measured errors stay below 0.4 of it,
    and a noise frame without a null has u_t of 4e-6 ... 2e-5): frames with u_t <= 20 tol are *sure* and held to tol of the
    row's peak (the north-star gate, unchanged); the others -- a bin within a few floors of zero -- to one propagated floor,
    2 u_t of their own value.  Returns (unsure, frames)."""
    flat = np.asarray(flat, dtype=np.float64)
    S = np.asarray(S, dtype=np.float64)
    assert flat.shape == ref.shape and np.isfinite(flat).all(), what
    floor = fft_floor(S)
    u = np.mean(floor[None, :] / (S + floor[None, :] + 1e-300), axis=0)
    pk = max(float(np.max(np.abs(ref))), 1e-300)
    err = np.abs(flat - ref)
    sure = u <= 20.0 * tol
    assert (err[sure] <= tol * pk).all(), f"{what}: sure frames off by {err[sure].max() / pk:.2e} peak-relative (> {tol:g})"
    assert (err[~sure] <= np.maximum(2.0 * u[~sure] * np.abs(ref[~sure]), tol * pk)).all(), \
        f"{what}: a below-margin frame exceeds the propagated one-floor bound"
    return int((~sure).sum()), int(sure.size)
