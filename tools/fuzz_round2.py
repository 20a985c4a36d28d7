"""Randomised shapes for the round-2 kernels against the oracle: the clip-resident and the chunked sosfiltfilt (random
designs, lengths around every chunk boundary, batches, row strides), the decimation chain against level-by-level
(identical bits), the CQT's octave kernels against each other (identical bits) and against the oracle.
usage: fuzz_round2.py SEED N"""
import os, sys
import numpy as np, torch
sys.path.insert(0, ".")
from sygnals_amd import ops
from sygnals_amd._cqt import decimation_taps
from oracle import cpu_ref as O

seed, N = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng(seed)
worst = {}
def track(name, got, want, tol=1e-5):
    got = np.asarray(got, dtype=np.complex128 if np.iscomplexobj(want) else np.float64)
    assert got.shape == want.shape, (name, got.shape, want.shape)
    pk = np.max(np.abs(want)) if want.size else 0.0
    err = (np.max(np.abs(got - want)) / pk) if pk > 0 else float(np.max(np.abs(got))) if got.size else 0.0
    worst[name] = max(worst.get(name, 0.0), err)
    assert err <= tol, (name, err)

def loglen(lo, hi):
    return int(np.exp(rng.uniform(np.log(lo), np.log(hi))))

taps = ops.to_device_f32(decimation_taps().astype(np.float32))
for it in range(N):
    # ---- zero-phase filtering
    fs = float(rng.choice([8000.0, 16000.0, 22050.0, 48000.0]))
    kind = ("lowpass", "highpass", "bandpass", "bandstop")[int(rng.integers(0, 4))]
    order = int(rng.integers(1, 7))
    if kind in ("bandpass", "bandstop"):
        lo = float(rng.uniform(0.01, 0.3)) * fs / 2
        cut = (lo, float(min(lo * rng.uniform(1.3, 6.0), 0.95 * fs / 2)))
    else:
        cut = float(rng.uniform(0.01, 0.9)) * fs / 2
    sos = O.design_butterworth_sos(cut, fs, order, kind)
    pad = O.sosfiltfilt_padlen(sos)
    bnd = int(rng.choice([64, 128, 192, 256])) * 256 - 2 * pad
    L = bnd + int(rng.integers(-3, 4)) if rng.random() < 0.3 and bnd > pad + 4 else max(pad + 1, loglen(pad + 1, 120000))
    B = int(rng.integers(1, 5))
    X = (rng.normal(0, 1, (B, L)) * 10 ** rng.uniform(-2, 2) + rng.uniform(-1, 1)).astype(np.float32)
    xd = ops.to_device_f32(np.pad(X, ((0, 0), (0, 3))))[:, :L] if rng.random() < 0.3 else ops.to_device_f32(X)
    y = ops.sosfiltfilt(xd, sos, O.sosfilt_zi(sos), pad).cpu().numpy()
    for b in range(B):
        track(f"sosfiltfilt ({len(sos)} sections)", y[b], O.apply_sos_filter(sos, X[b].astype(np.float64)))
    # ---- decimation chain
    Ld = loglen(1, 3_000_000)
    xd = ops.to_device_f32((rng.normal(0, 1, (int(rng.integers(1, 3)), Ld))).astype(np.float32))
    levels = int(rng.integers(1, 8))
    ref, cur = [], xd
    for _ in range(levels):
        cur = ops.decimate2(cur, taps, 1.4142135)
        ref.append(cur)
    got = ops.decimate2_chain(xd, taps, 1.4142135, levels)
    for g, r in zip(got, ref):
        assert g.shape == r.shape and torch.equal(g, r), ("decimation chain", Ld, levels)
    worst["decimation chain (bits)"] = 0.0
    # ---- CQT: octave kernels against each other and against the oracle
    if it % 3 == 0:
        sr = int(rng.choice([16000, 22050, 44100, 48000]))
        hop = int(rng.choice([128, 256, 512, 1024])) if rng.random() < 0.7 else int(rng.integers(1, 9)) * 64
        n_bins = int(rng.choice([36, 48, 60, 72, 84]))
        if n_bins == 84 and sr < 44100: n_bins = 72
        Lc = loglen(2 * hop + 2000, 400000)
        x = (rng.normal(0, 0.2, (1, Lc)) + np.sin(np.arange(Lc) * 2 * np.pi * 440.0 / sr)).astype(np.float32)
        xd = ops.to_device_f32(x)
        try:
            want = O.cqt(x[0].astype(np.float64), sr, hop_length=hop, n_bins=n_bins)
        except Exception:
            continue
        outs = []
        for st in (2, 1, 0):
            with ops.override(cqt_staged=st):
                outs.append(ops.cqt(xd, sr, hop_length=hop, n_bins=n_bins))
        assert torch.equal(outs[0], outs[2]) and torch.equal(outs[1], outs[2]), ("cqt staged bits", sr, hop, n_bins, Lc)
        g = outs[0][0].cpu().numpy()
        track("cqt", g[..., 0] + 1j * g[..., 1], want)
print("worst relative errors:", {k: float("%.3g" % v) for k, v in worst.items()})
