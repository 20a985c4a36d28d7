"""Randomised shape fuzz of the fused STFT(2048) paths against the float64 oracle (development aid)."""
import sys
import numpy as np, torch
sys.path.insert(0, ".")
from sygnals_amd import ops
from oracle import cpu_ref as O
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
N = int(sys.argv[2]) if len(sys.argv) > 2 else 120
worst = 0.0
for it in range(N):
    B = int(rng.choice([1, 2, 3, 5, 17, 130, 257]))
    hop = int(rng.choice([1, 7, 64, 128, 160, 256, 441, 512, 513, 700, 1024, 2048]))
    center = bool(rng.integers(0, 2))
    L = int(rng.integers(2048 if not center else 1, 30000 if hop >= 64 else 6000))
    n_mels = int(rng.choice([8, 16, 24, 40, 64, 128]))
    n_mfcc = int(rng.integers(1, min(n_mels, 24) + 1))
    sr = int(rng.choice([8000, 16000, 22050, 44100, 48000]))
    off = int(rng.integers(0, 4))
    Y = (rng.normal(0, 0.2, (B, L + 8)) + 0.1 * np.sin(np.arange(L + 8) * 0.05)[None, :]).astype(np.float32)
    big = ops.to_device_f32(Y)
    y = big[:, off:off + L]                       # unaligned views exercise the dword DMA / scalar paths
    T = ops.num_frames(L, 2048, hop, center)
    if T <= 0 or T > 4000:
        continue
    two = ops.mfcc_batch(y, sr, hop=hop, n_mels=n_mels, n_mfcc=n_mfcc, center=center, fused=False).cpu().numpy()
    chk = [0] if B > 3 else list(range(B))
    for b in chk + ([B - 1] if B > 3 else []):
        ref = O.mfcc_manager(Y[b, off:off + L].astype(np.float64), sr, 2048, hop, center, "hann", n_mels, n_mfcc)
        e = float(np.max(np.abs(two[b] - ref)) / max(np.max(np.abs(ref)), 1e-30))
        worst = max(worst, e)
        assert e <= 1e-5, (it, "two-launch", B, L, hop, center, n_mels, n_mfcc, sr, off, e)
    if ops.mfcc_fused_fits(n_mels, T, n_mfcc):
        one = ops.mfcc_batch(y, sr, hop=hop, n_mels=n_mels, n_mfcc=n_mfcc, center=center, fused=True).cpu().numpy()
        e = float(np.max(np.abs(one - two)) / max(np.max(np.abs(two)), 1e-30))
        worst = max(worst, e)
        assert e <= 1e-5, (it, "one-launch vs two-launch", B, L, hop, center, n_mels, n_mfcc, sr, off, e)
    # statistics + contrast on a sub-batch
    if it % 4 == 0:
        feats = ["spectral_centroid", "spectral_rolloff", "spectral_bandwidth"]
        from sygnals_amd.core.features.manager import extract_features_batch
        out = extract_features_batch(y[:2], sr, feats, 2048, hop, center)
        ref = O.extract_features(Y[0, off:off + L].astype(np.float64), sr, feats, 2048, hop, center)
        for k in ("spectral_centroid", "spectral_bandwidth"):
            e = float(np.max(np.abs(out[k][0] - ref[k])) / max(np.max(np.abs(ref[k])), 1e-30))
            worst = max(worst, e)
            assert e <= 2e-5, (it, k, B, L, hop, center, sr, off, e)
print(f"fuzz ok: {N} configurations, worst peak-relative error {worst:.2e}")
