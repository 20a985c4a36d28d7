"""Time-domain frame-feature kernel by row subset (which rows cost what)."""
import sys, numpy as np, torch
sys.path.insert(0, ".")
from sygnals_amd import ops
from sygnals_amd.synth import synth_clips
B = 1024
Y = synth_clips(64, 48000, 48000, seed=1); y = ops.to_device_f32(np.tile(Y, (B // 64, 1)))
def t(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for _ in range(100): ops.frame_stats(y)
for name, m in (("all 9", 0x1FF), ("mean only", 1), ("moments (mean std skew kurt peak crest rms)", 0x1 | 2 | 4 | 8 | 16 | 32 | 128), ("entropy only", 64),
                ("zcr only", 256), ("all but entropy", 0x1FF & ~64), ("all but zcr", 0x1FF & ~256)):
    print(f"{name:50s} {t(lambda: ops.frame_stats(y, mask=m)):8.1f} us")
