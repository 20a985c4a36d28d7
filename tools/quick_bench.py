"""Quick on-GPU timing of the fused path (development aid)."""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from sygnals_amd import ops
from sygnals_amd.synth import synth_clips
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
Y = synth_clips(32, 48000, 48000, seed=1)
y = ops.to_device_f32(np.tile(Y, (B // 32, 1)))
from sygnals_amd import _tables as TT
CPLAN = TT.contrast_plan(np.fft.rfftfreq(2048, 1/48000), 48000)
for _ in range(3):
    ops.mfcc_batch(y, 48000, n_mels=40)
torch.cuda.synchronize()
for name, fn in (("mfcc", lambda: ops.mfcc_batch(y, 48000, n_mels=40)),
                 ("mel only", lambda: ops.stft2048_mel(y, 48000, n_mels=40)),
                 ("mel+stats(all)", lambda: ops.stft2048_mel(y, 48000, n_mels=40, want_stats=True)),
                 ("mel+centroid+rolloff", lambda: ops.stft2048_mel(y, 48000, n_mels=40, want_stats=9)),
                 ("C4: mel+cen+roll+contrast", lambda: ops.stft2048_mel(y, 48000, n_mels=40, want_stats=9, contrast=CPLAN))):
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print(f"{name}: {ms*1e3:.1f} us/step  {B*48000/ms/1e3:.0f} Msamples/s  roofline {B*196888/ms/1e-3/8e12*100:.2f}%")

# --- raw C-ABI call loop of the one-launch MFCC (no Python wrapper work per step) ---
import ctypes as C
from sygnals_amd._lib import lib
from sygnals_amd import _tables as T2
cfg16 = ops.mel_config(48000, 2048, 40, waves=16)
dct = ops._dev(T2.dct_matrix(13, 40))
Tn = 94
mf = torch.empty((B, 13, Tn), dtype=torch.float32, device=y.device)
win = ops.window_dev("hann", 2048, 2048); tw = ops.twiddle_dev(2048)
args = (ops._ptr(y), B, 48000, 48000, 512, 1, Tn, ops._ptr(win), ops._ptr(tw), ops._ptr(cfg16.wpacked),
        cfg16.plan.ctypes.data_as(C.c_void_p), 40, ops._ptr(dct), 13, None, 1e-10, 80.0, 1, 1.0, None, ops._ptr(mf),
        C.c_void_p(torch.cuda.current_stream().cuda_stream))
h = lib()
for _ in range(3): h.syg_stft2048_mfcc_f32(*args)
torch.cuda.synchronize()
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): h.syg_stft2048_mfcc_f32(*args)
e1.record(); torch.cuda.synchronize()
print(f"one-launch MFCC, raw C calls: {e0.elapsed_time(e1)/20*1e3:.1f} us/step")

def timeit(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for rep in range(3):
    print(f"rep {rep}: wrapper {timeit(lambda: ops.mfcc_batch(y, 48000, n_mels=40)):.1f} us | raw {timeit(lambda: h.syg_stft2048_mfcc_f32(*args)):.1f} us | "
          f"two-launch {timeit(lambda: ops.mfcc_batch(y, 48000, n_mels=40, fused=False)):.1f} us | mel only {timeit(lambda: ops.stft2048_mel(y, 48000, n_mels=40)):.1f} us")
