"""Per-kernel times of the f-3 rows (run under rocprofv3 --kernel-trace --stats)."""
import sys
import numpy as np, torch
sys.path.insert(0, ".")
from sygnals_amd import ops
from sygnals_amd.core import dsp as D
which = sys.argv[1] if len(sys.argv) > 1 else "conv"
y = torch.randn((1024, 48000), dtype=torch.float32, device="cuda") * 0.3
kern = torch.randn(1023, dtype=torch.float32, device="cuda") / 32
for _ in range(10):
    if which == "conv": D.convolve_batch(y, kern, "same")
    elif which == "hilbert": D.analytic_batch(y)
    elif which == "pgram": D.periodogram_batch(y, fs=48000.0)
torch.cuda.synchronize()
