import ctypes as C, sys
import numpy as np, torch
sys.path.insert(0, ".")
from sygnals_amd import ops
from oracle import cpu_ref as O
spin = C.CDLL("/tmp/libspin.so")
spin.spin_launch.argtypes = [C.c_int, C.c_longlong, C.c_void_p]
B, L, SR = 1024, 48000, 48000
y = ops.to_device_f32(np.tile(O.synth_clips(64, L, SR, seed=1), (B // 64, 1)))
for _ in range(300): ops.mfcc_batch(y, SR, n_mels=40)
side = torch.cuda.Stream()
dyn = sys.argv[1] == "1"
for _ in range(30):
    side.wait_stream(torch.cuda.current_stream())
    spin.spin_launch(16, 250000, C.c_void_p(side.cuda_stream))
    ops.mfcc_batch(y, SR, n_mels=40, dynamic=dyn)
torch.cuda.synchronize()
