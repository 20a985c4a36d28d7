"""Step time vs iteration from a cold start (GPU clock ramp): development aid."""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from sygnals_amd import ops
from oracle import cpu_ref as O
B = 1024
Y = O.synth_clips(32, 48000, 48000, seed=1)
y = ops.to_device_f32(np.tile(Y, (B // 32, 1)))
ops.mfcc_batch(y, 48000, n_mels=40); torch.cuda.synchronize()
time.sleep(2.0)
evs = [torch.cuda.Event(enable_timing=True) for _ in range(41)]
evs[0].record()
for i in range(40):
    for _ in range(50): ops.mfcc_batch(y, 48000, n_mels=40)
    evs[i + 1].record()
torch.cuda.synchronize()
print("us/step per block of 50:", " ".join(f"{evs[i].elapsed_time(evs[i+1])/50*1e3:.0f}" for i in range(40)))
