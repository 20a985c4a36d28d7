"""Development probe (needs the -DSYG_SOSC_STAMP build): per-clip phase stamps of the clip-resident sosfiltfilt kernel
on the C3 batch (1024 clips x 1 s @ 48 kHz)."""
import ctypes, os, sys
import numpy as np
import torch
sys.path.insert(0, ".")
from sygnals_amd import ops
from oracle import cpu_ref as O
sos = O.design_butterworth_sos((300.0, 3400.0), 48000.0, 4, "bandpass")
x = torch.randn(1024, 48000, device="cuda") * 0.1
for _ in range(3):
    ops.sosfiltfilt(x, sos, O.sosfilt_zi(sos), 27)
torch.cuda.synchronize()
h = ctypes.CDLL(os.environ["SYGNALS_AMD_LIB"])
buf = np.zeros(8 * 1024, dtype=np.uint64)
print("rc", h.syg_debug_sosc_stamps(buf.ctypes.data_as(ctypes.c_void_p), buf.size))
s = buf.reshape(1024, 8).astype(np.int64)
t0 = s[:, 0].min()
us = (s - t0) / 100.0
names = ["start", "loaded+transposed", "pass 1 done", "scan 1 done", "pass 3 done", "fill + pass 1b done", "scan 2 done", "pass 3b done"]
for k, n in enumerate(names):
    print("%-22s median %7.2f  min %7.2f  max %7.2f" % (n, np.median(us[:, k]), us[:, k].min(), us[:, k].max()))
d = np.diff(us, axis=1)
print("phase durations (median us):", np.round(np.median(d, axis=0), 2))
first = us[:, 0] < 1.0
print("clips started in the first microsecond:", int(first.sum()), " their median end-of-pass-3b:", np.median(us[first, 7]))
