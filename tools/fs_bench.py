"""frame_stats time per selected row set (mask bits: 1 mean 2 std 4 skew 8 kurt 16 peak 32 crest 64 entropy 128 rms 256 zcr)."""
import sys, json
import numpy as np, torch
sys.path.insert(0, ".")
from sygnals_amd import ops
from oracle import cpu_ref as O
from tools.row_bench_util import timeit
B, L, SR = 1024, 48000, 48000
Y = O.synth_clips(64, L, SR, seed=1)
y = ops.to_device_f32(np.tile(Y, (B // 64, 1)))
for _ in range(200): ops.frame_stats(y, 2048, 512, True)
for name, mask in (("all", 511), ("mean", 1), ("moments", 2 | 4 | 8), ("peak+crest", 48), ("entropy", 64), ("rms", 128), ("zcr", 256),
                   ("all but entropy", 511 - 64), ("all but zcr", 511 - 256)):
    print(name, round(timeit(lambda: ops.frame_stats(y, 2048, 512, True, mask=mask), 30) * 1e3, 4), "ms", flush=True)
