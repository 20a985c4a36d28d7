"""Cost of the parts of frame_stats_kernel on the C2 batch: row masks (development aid)."""
import sys
import numpy as np, torch
sys.path.insert(0, ".")
from sygnals_amd import ops
from sygnals_amd.synth import synth_clips
from tools.row_bench_util import timeit
y = ops.to_device_f32(np.tile(synth_clips(64, 48000, 48000, seed=1), (16, 1)))
for _ in range(100): ops.frame_stats(y, 2048, 512, True)
import inspect
print(inspect.signature(ops.frame_stats))
for name, mask in (("all nine rows", 511), ("without entropy", 511 & ~64), ("mean|x| + std + peak + rms only", 1 | 2 | 16 | 128),
                   ("rms only", 128), ("zcr only", 256), ("entropy only", 64), ("skew + kurtosis only", 4 | 8)):
    try:
        t = min(timeit(lambda: ops.frame_stats(y, 2048, 512, True, mask=mask), 30, 5) for _ in range(3))
        print(f"{name:40s} {t * 1e6:8.1f} us")
    except TypeError as e:
        print("no mask argument:", e); break
