"""Development probe (needs the -DSYG_CQF_STAMP build, SYGNALS_AMD_LIB): per-phase time of the one-launch CQT kernel on a 1-hour
stream, summed over a workgroup's steps (lane 0's wall clock at the phase boundaries)."""
import ctypes, os, sys
import numpy as np
import torch
sys.path.insert(0, ".")
from sygnals_amd import ops
x = torch.randn(1, 48000 * 3600, device="cuda") * 0.1
for _ in range(3):
    ops.cqt(x, 48000)
torch.cuda.synchronize()
h = ctypes.CDLL(os.environ["SYGNALS_AMD_LIB"])
buf = np.zeros(8 * 256, dtype=np.uint64)
rc = h.syg_debug_cqf_stamps(buf.ctypes.data_as(ctypes.c_void_p), buf.size)
s = buf.reshape(256, 8).astype(np.float64) / 100.0          # microseconds
s = s[s.sum(axis=1) > 0]
names = ["phase 0: samples, level 4 | octaves 2, 6", "phase 1: levels 1 + 5 | octave 3", "phase 2: levels 2 + 6 | octaves 0, 4",
         "phase 3: levels 3 + 7 | octaves 1, 5"]
print("rc", rc, "workgroups", len(s), "total per workgroup: median %.1f us" % np.median(s.sum(axis=1)))
for i, nme in enumerate(names):
    print("%-40s median %7.1f us  (%4.1f %%)   min %7.1f  max %7.1f" % (nme, np.median(s[:, i]), 100 * np.median(s[:, i]) / np.median(s.sum(axis=1)), s[:, i].min(), s[:, i].max()))
