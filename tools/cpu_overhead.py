import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from sygnals_amd import ops
y = torch.zeros((8, 48000), dtype=torch.float32, device="cuda")
ops.mfcc_batch(y, 48000, n_mels=40, fused=True); torch.cuda.synchronize()
import cProfile, pstats
N = 200
t0 = time.perf_counter()
for _ in range(N): ops.mfcc_batch(y, 48000, n_mels=40, fused=True)
t1 = time.perf_counter(); torch.cuda.synchronize()
print(f"CPU per call (small batch, GPU ~idle): {(t1-t0)/N*1e6:.1f} us")
pr = cProfile.Profile(); pr.enable()
for _ in range(N): ops.mfcc_batch(y, 48000, n_mels=40, fused=True)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
