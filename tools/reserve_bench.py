import os, sys
import numpy as np, torch
sys.path.insert(0, ".")
from sygnals_amd import ops
from oracle import cpu_ref as O
from tools.row_bench_util import timeit
B, L, SR = 1024, 48000, 48000
y = ops.to_device_f32(np.tile(O.synth_clips(64, L, SR, seed=1), (B // 64, 1)))
for _ in range(300): ops.mfcc_batch(y, SR, n_mels=40)
for r in (0, 8, 16, 24, 32, 48):
    ops.set_reserved_cus(r)
    print(r, "reserved:", round(timeit(lambda: ops.mfcc_batch(y, SR, n_mels=40, fused=False), 200, 50) * 1e6, 1), "us two-launch")
