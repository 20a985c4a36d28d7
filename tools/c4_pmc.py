"""One variant of the fused C4 kernel, launched back to back (the target of a rocprofv3 --pmc / --kernel-trace pass).
    python3 tools/c4_pmc.py <mel|stats|contrast|c4|c4one|mfcc> [launches]
mel: stft2048_kernel<16,2,0> (mel only); stats: <16,2,1> with centroid + rolloff; contrast: <16,2,1> with the seven
contrast bands; c4: <16,2,1> with everything BASELINE config 4 asks for.  2048 clips (the per-GPU share of C4)."""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from sygnals_amd import ops, _tables as T
from sygnals_amd.synth import synth_clips

variant = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
B = 2048
Y = synth_clips(64, 48000, 48000, seed=1)
y = ops.to_device_f32(np.tile(Y, (B // 64, 1)))
CP = T.contrast_plan(np.fft.rfftfreq(2048, 1 / 48000), 48000)
from sygnals_amd.core.features.manager import feature_block_dominant
FBD = feature_block_dominant(y, 48000, 512, 40, 13)
fn = {
    "mel": lambda: ops.stft2048_mel(y, 48000, n_mels=40),
    "stats": lambda: ops.stft2048_mel(y, 48000, n_mels=40, want_stats=9),
    "contrast": lambda: ops.stft2048_mel(y, 48000, n_mels=40, contrast=CP),
    "c4": lambda: ops.stft2048_mel(y, 48000, n_mels=40, want_stats=9, contrast=CP),
    # c4one: <16,2,7>, the one-launch form that is feature_block's default since round 3 (segment-sum projection, MFCC
    # rows + centroid + rolloff + contrast means from one launch); mfcc: <16,2,6>, the headline kernel on the same batch
    "c4one": lambda: FBD[1](),
    "mfcc": lambda: ops.stft2048_mfcc(y, 48000, n_mels=40, n_mfcc=13),
}[variant]
for _ in range(n):
    fn()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(n):
    fn()
e1.record()
torch.cuda.synchronize()
print(f"{variant}: {e0.elapsed_time(e1) / n * 1e3:.1f} us per launch of {B} clips ({B * 94} frames)")
