import sys
import numpy as np, torch
sys.path.insert(0, ".")
from sygnals_amd import ops
from oracle import cpu_ref as O
B, L, SR = 1024, 48000, 48000
Y = O.synth_clips(64, L, SR, seed=1)
y = ops.to_device_f32(np.tile(Y, (B // 64, 1)))
sos = O.design_butterworth_sos((300.0, 3400.0), SR, 4, "bandpass")
zi = O.sosfilt_zi(sos); padlen = O.sosfiltfilt_padlen(sos)
for _ in range(10): ops.sosfiltfilt(y, sos, zi, padlen)
torch.cuda.synchronize()
