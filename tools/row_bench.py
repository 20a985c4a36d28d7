"""Timing of every SURVEY-8 row at its BASELINE configuration on one MI355X (development / documentation aid).
Prints one line per row: time, input Msamples/s and algorithmic GB/s (bytes per SURVEY 8d)."""
import json, sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from sygnals_amd import ops, _tables as T
from sygnals_amd.core import dsp as D
from oracle import cpu_ref as O

def timeit(fn, n=20, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3

rows = []
def report(name, secs, samples, algo_bytes, note=""):
    r = {"row": name, "ms": round(secs * 1e3, 4), "Msamples_per_s": round(samples / secs / 1e6, 1),
         "algorithmic_GBps": round(algo_bytes / secs / 1e9, 1), "hbm_frac": round(algo_bytes / secs / 8e12, 4), "note": note}
    rows.append(r); print(json.dumps(r), flush=True)

B, L, SR = 1024, 48000, 48000
Y = O.synth_clips(64, L, SR, seed=1)
y = ops.to_device_f32(np.tile(Y, (B // 64, 1)))
ops.mfcc_batch(y, SR, n_mels=40); torch.cuda.synchronize()
for _ in range(300): ops.mfcc_batch(y, SR, n_mels=40)       # clock warm-up
Tn = 94
report("C2 a1-a5 STFT->mel->MFCC (one launch)", timeit(lambda: ops.mfcc_batch(y, SR, n_mels=40), 100), B * L, B * (4 * L + 4 * 13 * Tn))
report("a1 complex STFT 2048/512 (frame-major c64 out)", timeit(lambda: ops.stft2048_c2c(y[:256])), 256 * L, 256 * (4 * L + 8 * 1025 * Tn), "256 clips")
# C3: band-pass filtfilt then MFCC
sos = O.design_butterworth_sos((300.0, 3400.0), SR, 4, "bandpass")
zi = O.sosfilt_zi(sos); padlen = O.sosfiltfilt_padlen(sos)
report("a13 sosfiltfilt order-4 band-pass (C3 filter alone)", timeit(lambda: ops.sosfiltfilt(y, sos, zi, padlen)), B * L, B * 8 * L)
report("C3 filtfilt + MFCC", timeit(lambda: ops.mfcc_batch(ops.sosfiltfilt(y, sos, zi, padlen), SR, n_mels=40)), B * L, B * (4 * L + 4 * 13 * Tn),
       "filtered waveform is a non-algorithmic intermediate")
# C4 per-GPU share: 2048 clips, MFCC + centroid + rolloff + contrast
y4 = ops.to_device_f32(np.tile(Y, (2048 // 64, 1)))
CP = T.contrast_plan(np.fft.rfftfreq(2048, 1 / SR), SR)
def c4():
    mel, st, cpv = ops.stft2048_mel(y4, SR, n_mels=40, want_stats=9, contrast=CP)
    ops.logmel_dct(mel, 13); ops.contrast_db(cpv)
report("C4 share: MFCC + centroid + rolloff + contrast, 2048 clips", timeit(c4), 2048 * L, 2048 * (4 * L + 4 * 22 * Tn))
report("a6-a9 all five spectral statistics + mel, 1024 clips", timeit(lambda: ops.stft2048_mel(y, SR, n_mels=40, want_stats=31)), B * L, B * (4 * L + 4 * (40 + 5) * Tn))
report("f-1 time-domain frame features (9 rows), 1024 clips", timeit(lambda: ops.frame_stats(y, 2048, 512, True)), B * L, B * (4 * L + 4 * 9 * Tn))
# a10: batched FFT
xf = torch.randn((4096, 4096, 2), dtype=torch.float32, device=y.device)
report("a10 complex FFT n=4096 x 4096 signals", timeit(lambda: ops.fft_pow2(xf)), 4096 * 4096, 2 * 8 * 4096 * 4096)
x48 = torch.randn((1024, 48000, 2), dtype=torch.float32, device=y.device)
report("a10 complex FFT n=48000 (2^7 3 5^3) x 1024 signals", timeit(lambda: ops.fft_any(x48), 10, 3), 1024 * 48000, 2 * 8 * 1024 * 48000)
x6k = torch.randn((8192, 6000, 2), dtype=torch.float32, device=y.device)
report("a10 complex FFT n=6000 (2^4 3 5^3) x 8192 signals", timeit(lambda: ops.fft_any(x6k), 10, 3), 8192 * 6000, 2 * 8 * 8192 * 6000)
# C5: one 1-hour stream per GPU (here 10 minutes to bound the run; rates are per sample)
Ls = 48000 * 600
t = torch.arange(Ls, device=y.device, dtype=torch.float32) / SR
stream = (0.3 * torch.sin(2 * np.pi * (30.0 + 10.0 * t) * t) + 0.05 * torch.randn(Ls, device=y.device)).reshape(1, Ls)
report("a14 Welch nperseg 4096 / 50 %, 10 min stream", timeit(lambda: D.welch_batch(stream, fs=SR, nperseg=4096), 5, 2), Ls, 4 * Ls + 4 * 2049)
report("a15 CQT 84 bins hop 512, 10 min stream", timeit(lambda: ops.cqt(stream, SR), 3, 1), Ls, 4 * Ls + 8 * 84 * (1 + Ls // 512))
# f-3: FFT-backed 1-D operations on the 1024-clip batch
kern = ops.to_device_f32((np.random.default_rng(0).normal(0, 1, 1023) / 32).astype(np.float32))
report("f-3 convolution, 1023-tap shared kernel, mode=same, 1024 clips", timeit(lambda: D.convolve_batch(y, kern, "same")), B * L, B * 8 * L)
report("f-3 autocorrelation (full), 1024 clips", timeit(lambda: D.convolve_batch(y, y, "full", correlate=True)), B * L, B * (4 * L + 4 * (2 * L - 1)))
y65 = torch.randn((1024, 65536), dtype=torch.float32, device=y.device)
report("f-3 Hilbert envelope, 1024 rows x 65536", timeit(lambda: ops.cabs_pow(D.analytic_batch(y65), 1)), 1024 * 65536, 1024 * 8 * 65536)
report("f-3 Hilbert envelope, 1024 clips x 48000 (mixed radix 200 x 240)", timeit(lambda: ops.cabs_pow(D.analytic_batch(y), 1), 5, 2), B * L, B * 8 * L)
report("f-3 periodogram, 1024 clips x 48000 (mixed radix 200 x 240)", timeit(lambda: D.periodogram_batch(y, fs=SR), 5, 2), B * L, B * (4 * L + 4 * (L // 2 + 1)))
json.dump(rows, open("gpurun_out/rows_r01.json", "w"), indent=1)
