"""Where the time-domain frame feature kernel spends its time: the C2 batch with subsets of the rows (development aid)."""
import sys
import numpy as np, torch
sys.path.insert(0, ".")
from sygnals_amd import ops
from sygnals_amd.synth import synth_clips
from tools.row_bench_util import timeit
y = ops.to_device_f32(np.tile(synth_clips(64, 48000, 48000, seed=1), (16, 1)))
for _ in range(50): ops.frame_stats(y)
for name, m in (("all nine rows", 0x1FF), ("without entropy", 0x1FF & ~64), ("without zcr", 0x1FF & ~256), ("without entropy, zcr", 0x1FF & ~(64 | 256)),
                ("mean |x| + rms only", 1 | 128), ("std only", 2), ("skew + kurtosis", 4 | 8), ("entropy only", 64), ("zcr only", 256)):
    t = min(timeit(lambda: ops.frame_stats(y, mask=m), 20, 5) for _ in range(3))
    print(f"{name:28s} {t * 1e6:8.1f} us")
