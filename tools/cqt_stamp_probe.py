"""Development probe (needs the -DSYG_CQT_STAMP build): per-wave start / table-ready / end stamps of the last staged
octave kernel of one CQT call on a 1-hour stream."""
import ctypes, os, sys
import numpy as np
import torch
sys.path.insert(0, ".")
from sygnals_amd import ops, _lib
x = torch.randn(1, 48000 * 3600, device="cuda") * 0.1
for _ in range(3):
    ops.cqt(x, 48000)
torch.cuda.synchronize()
h = ctypes.CDLL(os.environ["SYGNALS_AMD_LIB"])
buf = np.zeros(4 * 4096, dtype=np.uint64)
rc = h.syg_debug_cqt_stamps(buf.ctypes.data_as(ctypes.c_void_p), buf.size)
s = buf.reshape(4096, 4).astype(np.int64)
s = s[s[:, 0] > 0]
t0 = s[:, 0].min()
us = (s[:, :3] - t0) / 100.0
print("rc", rc, "waves", len(s))
cyc = (s[:, 3] - s[:, 1]) / ((s[:, 2] - s[:, 0]) / 100.0)       # shader cycles per microsecond over each wave's life
print("shader clock over wave lifetimes: median %.0f MHz (min %.0f, max %.0f)" % (np.median(cyc), cyc.min(), cyc.max()))
for k, name in ((0, "start"), (2, "end")):
    v = us[:, k]
    print("%-12s min %7.2f  p10 %7.2f  median %7.2f  p90 %7.2f  max %7.2f us" % (name, v.min(), np.percentile(v, 10), np.median(v), np.percentile(v, 90), v.max()))
life = us[:, 2] - us[:, 0]
print("lifetime     min %7.2f  median %7.2f  max %7.2f us" % (life.min(), np.median(life), life.max()))
