"""Randomised shapes for the f-3 operations (convolution / correlation / periodogram / analytic signal) and the PCM
ingest against the oracle.  usage: fuzz_dsp.py SEED N"""
import sys
import numpy as np, torch
sys.path.insert(0, ".")
from sygnals_amd import ops
from sygnals_amd.core import dsp as D
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

for it in range(N):
    B = int(rng.integers(1, 6))
    n, m = loglen(1, 40000), loglen(1, 5000)
    if rng.random() < 0.3: n, m = m, n
    X = (rng.normal(0, 1, (B, n)) * 10 ** rng.uniform(-3, 3)).astype(np.float32)
    shared = rng.random() < 0.5
    K = (rng.normal(0, 1, (1 if shared else B, m)) * 10 ** rng.uniform(-3, 1)).astype(np.float32)
    mode = ("full", "same", "valid")[int(rng.integers(0, 3))]
    corr = rng.random() < 0.5
    # non-contiguous row views now and then
    xd = ops.to_device_f32(np.pad(X, ((0, 0), (0, 5))))[:, :n] if rng.random() < 0.3 else ops.to_device_f32(X)
    got = D.convolve_batch(xd, ops.to_device_f32(K)[0] if shared else ops.to_device_f32(K), mode, correlate=corr).cpu().numpy()
    for b in range(B):
        kb = K[0 if shared else b].astype(np.float64)
        want = (O.compute_correlation if corr else O.apply_convolution)(X[b].astype(np.float64), kb, mode)
        track("correlation" if corr else "convolution", got[b], want)
    # periodogram / analytic signal on arbitrary lengths
    n2 = loglen(2, 20000)
    Y = (rng.normal(0, 1, (B, n2)) + rng.uniform(-2, 2)).astype(np.float32)
    nfft = [None, n2, n2 + int(rng.integers(1, 300)), max(2, n2 - int(rng.integers(0, n2 // 2 + 1)))][int(rng.integers(0, 4))]
    kw = dict(window=("hann", "hamming", "boxcar", "blackman")[int(rng.integers(0, 4))],
              detrend=("constant", False)[int(rng.integers(0, 2))], scaling=("density", "spectrum")[int(rng.integers(0, 2))])
    f, p = D.periodogram_batch(ops.to_device_f32(Y), fs=1000.0, nfft=nfft, **kw)
    for b in range(B):
        wf, wp = O.compute_psd_periodogram(Y[b].astype(np.float64), fs=1000.0, nfft=nfft, **kw)
        track("periodogram", p[b].cpu().numpy(), wp)
        assert np.allclose(f, wf, atol=1e-9)
    a = D.analytic_batch(ops.to_device_f32(Y)).cpu().numpy().astype(np.float64)
    for b in range(B):
        track("hilbert", a[b, :, 0] + 1j * a[b, :, 1], O.hilbert_transform(Y[b].astype(np.float64)))
    # PCM ingest
    ch = int(rng.integers(1, 5)); L = loglen(1, 30000)
    dt = (np.int16, np.int32, np.uint8)[int(rng.integers(0, 3))]
    info = np.iinfo(dt)
    pcm = rng.integers(info.min, info.max, (B, L, ch), dtype=dt, endpoint=True)
    got = ops.pcm_to_f32(torch.from_numpy(pcm if ch > 1 else np.ascontiguousarray(pcm[:, :, 0])).cuda()).cpu().numpy()
    off, sc = (128, 128.0) if dt == np.uint8 else (0, float(2 ** (8 * np.dtype(dt).itemsize - 1)))
    assert np.array_equal(got, ((pcm.astype(np.float64) - off) / sc).mean(axis=2).astype(np.float32)), ("pcm", dt, ch, L)
print("fuzz ok:", {k: f"{v:.1e}" for k, v in worst.items()})
