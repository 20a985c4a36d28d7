"""Development probe: one CQT call on streams of different lengths (kernel times come from rocprofv3 --kernel-trace)."""
import sys, torch
sys.path.insert(0, ".")
from sygnals_amd import ops
for mins in (60, 30, 15, 5):
    L = 48000 * 60 * mins
    x = torch.randn(1, L, device="cuda") * 0.1
    for _ in range(3):
        ops.cqt(x, 48000)
    torch.cuda.synchronize()
    del x
