import sys
import numpy as np, torch
sys.path.insert(0, ".")
from sygnals_amd import ops
for B, L in ((40, 200001), (7, 8192 * 24), (300, 30000), (2, 8192 * 83 * 3 + 17)):
    x = torch.randn((B, L), device="cuda") * 0.2
    a = ops.cqt(x, 48000)
    with ops.override(cqt_fused=False):
        b = ops.cqt(x, 48000)
    pk = float(b.abs().max()); err = float((a - b).abs().max()) / pk
    print(B, L, tuple(a.shape), "difference %.2e" % err)
    assert err <= 2e-6 and torch.isfinite(a).all()
print("ok")
