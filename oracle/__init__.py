"""oracle/ -- TEST INFRASTRUCTURE ONLY.

CPU (float64 NumPy/SciPy) restatement of the sygnals windowed-transform /
feature-extraction hot path.  It exists to *check* the HIP product path.

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import anything from here.  ``sygnals_amd`` (the product)
never imports it and has no CPU fallback: it raises when the HIP library or
a GPU is missing.

Parity status (see oracle/cpu_ref.py header and DESIGN.md section 3):
  * rows backed by SciPy/NumPy in the reference (filters, fft/ifft, window,
    Welch, the five per-frame spectral functions) are PINNED against golden
    vectors produced by running the reference's own functions
    (tests/golden/make_golden.py, tests/golden/ref_*.npz);
  * rows whose arithmetic lives in librosa (STFT, mel, power_to_db, MFCC,
    spectral_contrast, CQT, frame times) are "PARITY UNPINNED": librosa
    (pyproject pin ``librosa>=0.10.0``, no lock file) is not vendored under
    /root/reference and is not installed; they are restated from librosa's
    published algorithm and anchored on librosa's documented known answers,
    closed-form KATs and independent SciPy cross-checks.
"""
