"""Kernel timeline (run under rocprofv3 --kernel-trace) of MFCC steps with a stand-in communication kernel on a second
stream: see tools/queue_bench.py.  Build the stand-in first: hipcc --offload-arch=gfx950 -O3 -fPIC -shared
tools/ubench/spin.hip -o /tmp/libspin.so"""
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
two_launch = len(sys.argv) > 1 and sys.argv[1] == "1"     # 1: two-launch form, else the one-launch kernel
for _ in range(30):
    side.wait_stream(torch.cuda.current_stream())
    spin.spin_launch(16, 250000, C.c_void_p(side.cuda_stream))
    ops.mfcc_batch(y, SR, n_mels=40, fused=not two_launch)
torch.cuda.synchronize()
