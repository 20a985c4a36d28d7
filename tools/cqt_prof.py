import sys
import numpy as np, torch
sys.path.insert(0, ".")
from sygnals_amd import ops
SR = 48000; Ls = SR * 600
t = torch.arange(Ls, device="cuda", dtype=torch.float32) / SR
stream = (0.3 * torch.sin(2 * np.pi * (30.0 + 10.0 * t) * t) + 0.05 * torch.randn(Ls, device="cuda")).reshape(1, Ls)
for _ in range(5): ops.cqt(stream, SR)
torch.cuda.synchronize()
