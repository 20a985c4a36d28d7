"""Synthetic workloads of SURVEY section 8(d) -- the inputs `bench.py` and the row benchmarks time.

NumPy only; no device work and no dependency on `oracle/` (the oracle keeps its own statement of the same
recipe, and tests/test_host_logic.py checks that the two agree bit for bit).
"""
from __future__ import annotations

import numpy as np


def synth_clips(n_clips, length=48000, sr=48000, seed=20250523, dtype=np.float32):
    """C2/C3/C4 clips: per clip 3 sines (uniform random frequency in [50, 20 000] Hz, amplitude U[0.05, 0.3], random
    phase) + white Gaussian noise (sigma 0.05), scaled to peak <= 0.9 -- non-degenerate spectra, real `top_db`
    clamp activity."""
    rng = np.random.default_rng(seed)
    t = np.arange(length, dtype=np.float64) / sr
    out = np.empty((n_clips, length), dtype=dtype)
    for i in range(n_clips):
        f = rng.uniform(50.0, 20000.0 if sr >= 44100 else 0.45 * sr, 3)
        a = rng.uniform(0.05, 0.3, 3)
        ph = rng.uniform(0, 2 * np.pi, 3)
        y = (a[:, None] * np.sin(2 * np.pi * f[:, None] * t[None, :] + ph[:, None])).sum(axis=0)
        y += rng.normal(0.0, 0.05, length)
        y *= 0.9 / max(np.abs(y).max(), 1e-12) if np.abs(y).max() > 0.9 else 1.0
        out[i] = y.astype(dtype)
    return out


def synth_stream(length, sr=48000, seed=0, dtype=np.float32, block=1 << 22):
    """C5 stream: pink-ish noise (white noise through a one-pole low-pass) + a slow chirp 30 Hz -> 12 kHz over the
    stream, so that Welch and CQT outputs are not flat.  Generated block-wise (a 1-hour stream is 172.8 M samples)."""
    rng = np.random.default_rng(20250523 + seed)
    out = np.empty(length, dtype=dtype)
    state = 0.0
    a = 0.97
    for s0 in range(0, length, block):
        n = min(block, length - s0)
        w = rng.normal(0.0, 0.05, n)
        # one-pole low-pass by a cumulative form (vectorised): y[i] = a y[i-1] + (1 - a) w[i]
        from scipy.signal import lfilter
        y, zf = lfilter([1.0 - a], [1.0, -a], w, zi=[state * a])
        state = float(y[-1])
        t = (np.arange(s0, s0 + n, dtype=np.float64)) / sr
        dur = length / sr
        f0, f1 = 30.0, 12000.0
        ph = 2 * np.pi * (f0 * t + 0.5 * (f1 - f0) * t * t / dur)
        out[s0:s0 + n] = (4.0 * y + 0.2 * np.sin(ph)).astype(dtype)
    return out
