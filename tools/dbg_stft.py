import sys; sys.path.insert(0, '.')
import numpy as np, torch
from sygnals_amd import ops
n = np.arange(4096)
for k0 in (1, 2, 8, 16, 17, 100, 257, 300, 1000):
    y = np.cos(2*np.pi*k0*n/2048).astype(np.float32)[None]
    X = ops.stft2048_c2c(ops.to_device_f32(y), 512, center=False, window="boxcar").cpu().numpy()
    X = X[0, 1, :, 0] + 1j*X[0, 1, :, 1]
    top = np.argsort(-np.abs(X))[:4]
    print("k0", k0, "-> top bins", top, np.round(np.abs(X[top]), 1), " X[k0]=", np.round(X[k0], 1))
