"""Time the fused kernel per statistics mask (development aid)."""
import sys
import numpy as np, torch
sys.path.insert(0, ".")
from sygnals_amd import ops
from oracle import cpu_ref as O
B = 1024
Y = O.synth_clips(32, 48000, 48000, seed=1)
y = ops.to_device_f32(np.tile(Y, (B // 32, 1)))
for mask in (0, 1, 2, 4, 8, 16, 3, 9, 31):
    fn = (lambda: ops.stft2048_mel(y, 48000, n_mels=40)) if mask == 0 else (lambda: ops.stft2048_mel(y, 48000, n_mels=40, want_stats=mask))
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): fn()
    e1.record(); torch.cuda.synchronize()
    print(f"mask {mask:2d}: {e0.elapsed_time(e1)/10*1e3:.1f} us")
