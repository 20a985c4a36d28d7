"""A/B of the forms of the 2048 front end (development aid): clip-resident one launch (MODE 6 / 7), matrix-form tile kernel
(MODE 0 / 1), tile form of the segment-sum projection with 16 / 8 waves per workgroup (MODE 8 ... 11), each with its second
launch.  microseconds per call, C2 = 1024 clips, C4 = 2048 clips."""
import sys
import numpy as np, torch
sys.path.insert(0, ".")
from sygnals_amd import ops
from sygnals_amd.core.features.manager import feature_block
from sygnals_amd.synth import synth_clips
from tools.row_bench_util import timeit

SR, L = 48000, 48000
Y = synth_clips(64, L, SR, seed=1)
y = ops.to_device_f32(np.tile(Y, (1024 // 64, 1)))
y4 = ops.to_device_f32(np.tile(Y, (2048 // 64, 1)))
for _ in range(300): ops.mfcc_batch(y, SR, n_mels=40)


def t(name, fn, n=100, warm=20):
    best = min(timeit(fn, n, warm) for _ in range(3))
    print(f"{name:88s} {best * 1e6:8.1f} us", flush=True)


def two(yy, nm, proj, wv):
    return ops.mel_mfcc(ops.stft2048_mel(yy, SR, n_mels=nm, projection=proj, tri_waves=wv)[0], 13)


print("== C2: 1024 clips, 40 bands -> 13 MFCC")
t("one launch, clip-resident, segment sums (MODE 6; the headline)", lambda: ops.mfcc_batch(y, SR, n_mels=40))
t("two launches: matrix-form tile kernel (MODE 0) + dB / DCT", lambda: two(y, 40, "matrix", 16))
t("two launches: segment sums, tile form, 16 waves (MODE 10) + dB / DCT", lambda: two(y, 40, "segments", 16))
t("two launches: segment sums, tile form, 8 waves x 2 workgroups per CU (MODE 10) + dB / DCT", lambda: two(y, 40, "segments", 8))
t("   the mel launch alone: matrix", lambda: ops.stft2048_mel(y, SR, n_mels=40, projection="matrix"))
t("   the mel launch alone: segments, 16 waves", lambda: ops.stft2048_mel(y, SR, n_mels=40, projection="segments", tri_waves=16))
t("   the mel launch alone: segments, 8 waves", lambda: ops.stft2048_mel(y, SR, n_mels=40, projection="segments", tri_waves=8))
for nm in (128, 64):
    print(f"== 1024 clips, {nm} bands -> 13 MFCC")
    t("two launches: matrix-form tile kernel + dB / DCT", lambda: two(y, nm, "matrix", 16))
    t("two launches: segment sums (four passes), 16 waves + dB / DCT", lambda: two(y, nm, "segments", 16))
    t("two launches: segment sums (four passes), 8 waves + dB / DCT", lambda: two(y, nm, "segments", 8))
print("== C4: 2048 clips -> [2048, 22, 94] block")
t("one launch (MODE 7) + rows kernel (feature_block default)", lambda: feature_block(y4, SR), 50, 10)
t("two launches: matrix form + rows (MODE 1) + block kernel", lambda: feature_block(y4, SR, one_launch=False, projection="matrix"), 50, 10)
t("two launches: segment sums + rows, 16 waves (MODE 11) + block kernel", lambda: feature_block(y4, SR, one_launch=False, projection="segments"), 50, 10)
t("two launches: segment sums + rows, 8 waves (MODE 11) + block kernel", lambda: feature_block(y4, SR, one_launch=False, projection="segments", tri_waves=8), 50, 10)
# parity of the forms against each other (bits where the same row functions run on the same rows)
a = feature_block(y4[:64], SR).cpu().numpy()
for proj, wv in (("matrix", 16), ("segments", 16), ("segments", 8)):
    b = feature_block(y4[:64], SR, one_launch=False, projection=proj, tri_waves=wv).cpu().numpy()
    print(f"block vs one-launch, {proj}/{wv}: rows 13.. identical {np.array_equal(a[:, 13:], b[:, 13:])}, mfcc peak-rel "
          f"{np.abs(a[:, :13] - b[:, :13]).max() / np.abs(a[:, :13]).max():.2e}")
# C3
from sygnals_amd.core import filters as FL
sos = FL.design_butterworth_sos((300.0, 3400.0), SR, 4, "bandpass")
print("== C3")
t("sosfiltfilt order-4 band-pass, 1024 clips (persistent workgroups, in-wave prefix rounds)", lambda: FL.apply_sos_filter_batch(sos, y), 50, 10)
t("C3 filtfilt + MFCC", lambda: ops.mfcc_batch(FL.apply_sos_filter_batch(sos, y), SR, n_mels=40), 50, 10)
