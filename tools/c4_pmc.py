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
def one_launch_variant(smask, plan):
    """stft2048_kernel<16,2,7> with a cut-down request: the C4 launch with only `smask` statistics / the bands of `plan`."""
    import ctypes as C
    from sygnals_amd._lib import check, lib
    cfg = ops.mel_config(48000, 2048, 40, 0.0, None, waves=16)
    dct = ops._dev(T.dct_matrix(13, 40, 2, "ortho"))
    R = int(plan[0]) if plan is not None else 1
    out = torch.empty((B, 13 + 2 + R, 94), dtype=torch.float32, device="cuda")
    st = torch.zeros((B, 8, 94), dtype=torch.float32, device="cuda")
    cpv = torch.empty((B, 2, R, 94), dtype=torch.float32, device="cuda")
    keep = (cfg, dct, out, st, cpv, plan)

    def run(_keep=keep):
        check(lib().syg_stft2048_features_tri_f32(
            ops._ptr(y), B, 48000, y.stride(0), 512, 1, 94, ops._ptr(ops.window_dev("hann", 2048, 2048)), ops._ptr(ops.twiddle_dev(2048)),
            ops._ptr(cfg.segtab), int(cfg.segtab.numel()), 40, ops._ptr(dct), 13, None, 1e-10, 80.0, 1, 1.0, 48000.0, 0.85, 2.0,
            smask or 1, ops._ptr(st) if smask else None, plan.ctypes.data_as(C.c_void_p) if plan is not None else None,
            ops._ptr(cpv) if plan is not None else None, ops._ptr(out), 13 + 2 + R, C.c_void_p(ops._stream_ptr())), "features_tri")
    return run


CP0 = np.zeros_like(CP); CP0[0] = 1; CP0[1] = CP[1]; CP0[17] = CP[17]; CP0[33] = CP[33]      # band 0 alone
fn = {
    "one_cen": one_launch_variant(1 | 32, None),            # <16,2,7>: MFCC + centroid only
    "one_cenroll": one_launch_variant(1 | 8 | 32, None),    # ... + rolloff
    "one_band0": one_launch_variant(0, CP0),                # <16,2,7>: MFCC + contrast band 0 alone (min / max of 8 bins)
    "one_contrast": one_launch_variant(0, np.ascontiguousarray(CP, np.int32)),
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
