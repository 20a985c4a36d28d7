"""128 / 64 bands at n_fft 2048: the one-launch clip-resident form (the clip's mel matrix in the stage buffer's place, frames
straight from global memory) against the two-launch form (staged tile kernel + logmel_dct), same box."""
import sys
import numpy as np, torch
sys.path.insert(0, ".")
from sygnals_amd import ops
from sygnals_amd.synth import synth_clips
from oracle import cpu_ref as O
from tools.row_bench_util import timeit
Y = synth_clips(64, 48000, 48000, seed=1)
y = ops.to_device_f32(np.tile(Y, (16, 1)))
for _ in range(200): ops.mfcc_batch(y, 48000, n_mels=40)
for nm in (128, 64, 40):
    a = ops.mfcc_batch(y, 48000, n_mels=nm, fused=True)
    b = ops.mfcc_batch(y, 48000, n_mels=nm, fused=False)
    ref = O.mfcc(Y[3].astype(np.float64), 48000, n_mfcc=13, n_fft=2048, hop_length=512, n_mels=nm) if hasattr(O, "mfcc") else None
    d = float((a - b).abs().max() / b.abs().max())
    t1 = min(timeit(lambda: ops.mfcc_batch(y, 48000, n_mels=nm, fused=True), 50, 10) for _ in range(3))
    t2 = min(timeit(lambda: ops.mfcc_batch(y, 48000, n_mels=nm, fused=False), 50, 10) for _ in range(3))
    t0 = min(timeit(lambda: ops.mfcc_batch(y, 48000, n_mels=nm), 50, 10) for _ in range(3))
    print(f"n_mels {nm:3d}: one launch {t1 * 1e6:7.1f} us   two launches {t2 * 1e6:7.1f} us   default {t0 * 1e6:7.1f} us   difference {d:.2e}")
