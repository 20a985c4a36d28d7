"""Both CQT paths against the float64 oracle on a few shapes (development aid)."""
import sys
import numpy as np, torch
sys.path.insert(0, ".")
from sygnals_amd import ops
from oracle import cpu_ref as O
rng = np.random.default_rng(0)
for sr, L, kind in ((48000, 100003, "noise"), (48000, 100003, "noise+tone"), (44100, 150000, "noise"), (48000, 65536 * 2, "tone")):
    t = np.arange(L) / sr
    x = {"noise": rng.normal(0, 0.3, L), "noise+tone": rng.normal(0, 0.3, L) + 0.5 * np.sin(2 * np.pi * 440 * t),
         "tone": np.sin(2 * np.pi * 440 * t)}[kind].astype(np.float32)
    ref = O.cqt(x.astype(np.float64), sr)
    xd = ops.to_device_f32(x[None])
    a = ops.cqt(xd, sr).cpu().numpy()[0]; a = a[..., 0] + 1j * a[..., 1]
    with ops.override(cqt_fused=False):
        b = ops.cqt(xd, sr).cpu().numpy()[0]; b = b[..., 0] + 1j * b[..., 1]
    pk = np.abs(ref).max()
    print(f"{sr} {L} {kind:12s} one launch vs oracle {np.abs(a - ref).max() / pk:.2e}   level by level vs oracle {np.abs(b - ref).max() / pk:.2e}   between {np.abs(a - b).max() / pk:.2e}")
    # per octave
    for o in range(7):
        sl = slice(84 - 12 * (o + 1), 84 - 12 * o)
        print(f"   octave {o}: {np.abs(a[sl] - ref[sl]).max() / pk:.2e}  {np.abs(b[sl] - ref[sl]).max() / pk:.2e}")
