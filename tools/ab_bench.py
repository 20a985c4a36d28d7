"""A/B timing of library variants on one box: alternates the variants (subprocess per run, SYGNALS_AMD_LIB) so that
clock drift and box-to-box differences cancel; prints min / median per variant.
    python3 tools/ab_bench.py [--what mfcc|c4|mel] [--rounds 3] name=path.so [name=path.so ...]
(`product` as a path means the in-tree product library)"""
import os, subprocess, sys, statistics
what, rounds, libs = "mfcc", 3, []
args = sys.argv[1:]
while args:
    a = args.pop(0)
    if a == "--what": what = args.pop(0)
    elif a == "--rounds": rounds = int(args.pop(0))
    else: libs.append(a.split("=", 1))
CHILD = r'''
import sys, numpy as np, torch
sys.path.insert(0, ".")
from sygnals_amd import ops, _tables as T
from sygnals_amd.synth import synth_clips
what = sys.argv[1]
B = 1024
Y = synth_clips(64, 48000, 48000, seed=1); y = ops.to_device_f32(np.tile(Y, (B // 64, 1)))
CP = T.contrast_plan(np.fft.rfftfreq(2048, 1 / 48000), 48000)
from sygnals_amd.core.features.manager import feature_block as FB, extract_features_batch as EFB
stream = None
from sygnals_amd.core import filters as FL
SOS = FL.design_butterworth_sos((300.0, 3400.0), 48000, 4, "bandpass")
if what == "cqt":
    g = torch.Generator(device="cuda").manual_seed(5)
    stream = (torch.randn(48000 * 3600, device="cuda", generator=g, dtype=torch.float32) * 0.05).reshape(1, -1)
y2 = ops.to_device_f32(np.tile(Y, (2048 // 64, 1))) if what.startswith("c4blk") else None
XC = torch.randn((1024, 48000 if what == "fft48k" else 65536, 2), dtype=torch.float32, device="cuda") if what.startswith("fft") else None
fn = {"mfcc": lambda: ops.stft2048_mfcc(y, 48000, 512, True, "hann", 40, 13),
      "mfcc441": lambda: ops.stft2048_mfcc(y, 44100, 512, True, "hann", 40, 13),       # (a 3-step scan layout of the segment projection)
      "mfccmat": lambda: ops.stft2048_mfcc(y, 48000, 512, True, "hann", 40, 13, projection="matrix"),
      "mel": lambda: ops.stft2048_mel(y, 48000, n_mels=40),
      "mfcc1024": lambda: ops.stft_mfcc_pow2(y, 48000, 1024, 256, True, "hann", None, 40, 13),
      "mel1024": lambda: ops.stft_mel_pow2(y, 48000, 1024, 256, True, "hann", None, 40),
      "mfcc512": lambda: ops.stft_mfcc_pow2(y, 48000, 512, 128, True, "hann", None, 40, 13),
      "c4blk1": lambda: FB(y2, 48000, one_launch="segments"),
      "c4blk22k": lambda: FB(y2, 22050),        # the C4 block at other sample rates: contrast bands of 298 + 431 bins (k = 6, 9),
      "c4blk24k": lambda: FB(y2, 24000),        # 273 + 479 (5, 10),
      "c4blk32k": lambda: FB(y2, 32000),        # 206 + 616 (4, 12)
      "c4blk2": lambda: FB(y2, 48000, one_launch=False),
      "mel256seg": lambda: ops.stft_mel_wseg_small(y, 48000, 256, 64, True, "hann", None, 40),
      "mel512seg": lambda: ops.stft_mel_wseg_small(y, 48000, 512, 128, True, "hann", None, 40),
      "mel1024seg": lambda: ops.stft_mel_w1024_seg(y, 48000, 256, True, "hann", None, 40),
      "mel4096seg": lambda: ops.stft_mel_w4096(y, 48000, 1024, True, "hann", None, 40),
      "cqt": lambda: ops.cqt(stream, 48000),
      "fft48k": lambda: ops.fft_any(XC),                      # 1024 x 48000 complex, mixed radix 200 x 240
      "fft64k": lambda: ops.fft_pow2_any(XC),                 # 1024 x 65536 complex, four-step 256 x 256
      "stats5": lambda: ops.stft2048_mel(y, 48000, n_mels=40, want_stats=31),       # the a6-a9 row: all five statistics + mel (MODE 1)
      "stats5only": lambda: ops.stft2048_stats(y, 48000, want_stats=31),           # the same rows without the mel spectrogram
      "efb_c4": lambda: EFB(y, 48000, ["mfcc", "spectral_centroid", "spectral_rolloff", "spectral_contrast"], feature_params={"mfcc": {"n_mels": 40}}, to_host=False),   # the reference-API path of C4's features, 1024 clips
      "efb_mfcc": lambda: EFB(y, 48000, ["mfcc"], feature_params={"mfcc": {"n_mels": 40}}, to_host=False),
      "sos": lambda: FL.apply_sos_filter_batch(SOS, y),                               # C3's filter alone, 1024 clips
      "c4": lambda: ops.stft2048_mel(y, 48000, n_mels=40, want_stats=9, contrast=CP)}[what]
for _ in range(400 if what != "cqt" and not what.startswith("fft") else 20): fn()
torch.cuda.synchronize()
best = 1e9
for rep in range(5):
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    nrep = 100 if what != "cqt" and not what.startswith("fft") else 10
    for _ in range(nrep): fn()
    e1.record(); torch.cuda.synchronize()
    best = min(best, e0.elapsed_time(e1) * 1000 / nrep)
print("US", best)
'''
res = {n: [] for n, _ in libs}
for r in range(rounds):
    for n, path in libs:
        env = dict(os.environ)
        if path != "product":
            env["SYGNALS_AMD_LIB"] = os.path.abspath(path); env["SYGNALS_AMD_ALLOW_VARIANT"] = env.get("SYGNALS_AMD_ALLOW_VARIANT", "0")
        out = subprocess.run([sys.executable, "-c", CHILD, what], env=env, capture_output=True, text=True)
        us = [float(l.split()[1]) for l in out.stdout.splitlines() if l.startswith("US")]
        if not us:
            print(n, "FAILED", out.stderr[-400:]); continue
        res[n].append(us[0])
for n, v in res.items():
    if v: print(f"{what:5s} {n:14s} min {min(v):7.1f}  median {statistics.median(v):7.1f}  runs {['%.1f' % x for x in v]}")
