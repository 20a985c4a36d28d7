"""Randomised comparison of the one-launch CQT (syg_cqt_fused_f32) with the level-by-level kernels: random lengths (1 ... 4 M
samples: one to many segments, level lengths that round up, frames on segment borders), batches, sample rates and octave
counts.  python3 tools/fuzz_cqt_fused.py <seed> <cases>"""
import sys
import numpy as np, torch
sys.path.insert(0, ".")
from sygnals_amd import ops
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
cases = int(sys.argv[2]) if len(sys.argv) > 2 else 40
rng = np.random.default_rng(seed)
worst = 0.0
shapes = [(48000, 84, None), (44100, 84, None), (48000, 36, 523.25), (48000, 48, 261.63), (48000, 72, 65.41), (44100, 60, 130.81)]
for c in range(cases):
    sr, nb, fmin = shapes[rng.integers(len(shapes))]
    kind = rng.integers(4)
    if kind == 0: L = int(rng.integers(1, 4000))
    elif kind == 1: L = int(rng.integers(4000, 300000))
    elif kind == 2: L = int(65536 * rng.integers(1, 12) + rng.integers(-3, 4))
    else: L = int(rng.integers(300000, 4000000))
    B = int(rng.integers(1, 4)) if L < 1500000 else 1
    x = torch.randn((B, L), device="cuda") * 0.3
    if rng.integers(2): x += 0.5 * torch.sin(torch.arange(L, device="cuda") * (2 * np.pi * 440.0 / sr))
    if B > 1 and rng.integers(2):                       # a batch with a row stride that is no multiple of four
        buf = torch.zeros((B, L + 3), device="cuda"); buf[:, :L] = x; x = buf[:, :L]
    from sygnals_amd._cqt import CqtPlan
    if not CqtPlan(sr, 512, fmin, nb).one_launch_shape():
        print("case", c, "not the one-launch shape:", sr, nb, fmin); continue
    a = ops.cqt(x, sr, n_bins=nb, fmin=fmin)
    with ops.override(cqt_fused=False):
        b = ops.cqt(x, sr, n_bins=nb, fmin=fmin)
    assert a.shape == b.shape and torch.isfinite(a).all()
    pk = float(b.abs().max())
    err = float((a - b).abs().max()) / max(pk, 1e-30)
    worst = max(worst, err)
    assert err <= 2e-6, (c, sr, nb, fmin, L, B, err)
print(f"seed {seed}: {cases} cases, worst difference {worst:.2e} of the peak")
