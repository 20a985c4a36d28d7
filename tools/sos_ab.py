"""A/B of two builds of the clip-resident sosfiltfilt on the C3 batch (same box, alternating): SYGNALS_AMD_LIB selects the
library of a run; run twice and compare, or use tools/r04g.sh."""
import sys
import numpy as np, torch
sys.path.insert(0, ".")
from sygnals_amd import ops
from sygnals_amd.core import filters as FL
from sygnals_amd.synth import synth_clips
from tools.row_bench_util import timeit
y = ops.to_device_f32(np.tile(synth_clips(64, 48000, 48000, seed=1), (16, 1)))
sos = FL.design_butterworth_sos((300.0, 3400.0), 48000, 4, "bandpass")
for _ in range(200): FL.apply_sos_filter_batch(sos, y)
print("sosfiltfilt order-4 band-pass, 1024 clips:", " ".join(f"{timeit(lambda: FL.apply_sos_filter_batch(sos, y), 50, 10) * 1e6:.1f}" for _ in range(4)), "us")
