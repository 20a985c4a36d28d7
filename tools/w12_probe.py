import sys
import numpy as np, torch
sys.path.insert(0, ".")
from sygnals_amd import ops
from sygnals_amd.synth import synth_clips
from oracle import cpu_ref as O
from tools.row_bench_util import timeit
Y = synth_clips(64, 48000, 48000, seed=1)
y = ops.to_device_f32(np.tile(Y, (16, 1)))
out = ops.mfcc_batch(y, 48000, n_mels=40)
ref = O.mfcc_batch(Y[:4], 48000, n_mels=40, n_mfcc=13)
err = max(float(np.abs(out[i].cpu().numpy() - ref[i]).max() / np.abs(ref[i]).max()) for i in range(4))
for _ in range(300): ops.mfcc_batch(y, 48000, n_mels=40)
t = [timeit(lambda: ops.mfcc_batch(y, 48000, n_mels=40), 100, 10) * 1e6 for _ in range(3)]
print("C2 one launch: err %.2e  us" % err, " ".join("%.1f" % v for v in t))
