"""Time the fused kernel variants of the C4 probe with whatever library SYGNALS_AMD_LIB names (timing ablations)."""
import sys, numpy as np, torch
sys.path.insert(0, ".")
from sygnals_amd import ops, _tables as T
from sygnals_amd.synth import synth_clips
B = 1024
Y = synth_clips(32, 48000, 48000, seed=1); y = ops.to_device_f32(np.tile(Y, (B // 32, 1)))
CP = T.contrast_plan(np.fft.rfftfreq(2048, 1 / 48000), 48000)
def t(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for _ in range(200): ops.stft2048_mel(y, 48000, n_mels=40)
print("mel only %.1f  centroid %.1f  c4 %.1f" % (t(lambda: ops.stft2048_mel(y, 48000, n_mels=40)),
      t(lambda: ops.stft2048_mel(y, 48000, n_mels=40, want_stats=1)),
      t(lambda: ops.stft2048_mel(y, 48000, n_mels=40, want_stats=9, contrast=CP))))
