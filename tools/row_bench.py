"""Timing of every SURVEY-8 row at its BASELINE configuration on one MI355X (development / documentation aid).
Prints one line per row: time, input Msamples/s and algorithmic GB/s (bytes per SURVEY 8d).

    python tools/row_bench.py [substring ...]     only the rows whose name contains one of the substrings
    ROWS_OUT=path                                 where the JSON goes (default gpurun_out/rows_r04.json)
"""
import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from sygnals_amd import ops, _tables as T
from sygnals_amd.core import dsp as D
from sygnals_amd.core import filters as FL
from sygnals_amd.synth import synth_clips
ONLY = [a.lower() for a in sys.argv[1:]]

def timeit(fn, n=20, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3

rows = []
def want(name):
    return not ONLY or any(a in name.lower() for a in ONLY)        # (a substring of the row name, e.g. "c5", "cqt")
def report(name, fn, samples, algo_bytes, note="", n=20, warm=5):
    if not want(name):
        return
    secs = timeit(fn, n, warm)
    r = {"row": name, "ms": round(secs * 1e3, 4), "Msamples_per_s": round(samples / secs / 1e6, 1),
         "algorithmic_GBps": round(algo_bytes / secs / 1e9, 1), "hbm_frac": round(algo_bytes / secs / 8e12, 4), "note": note}
    rows.append(r); print(json.dumps(r), flush=True)

B, L, SR = 1024, 48000, 48000
Y = synth_clips(64, L, SR, seed=1)
y = ops.to_device_f32(np.tile(Y, (B // 64, 1)))
ops.mfcc_batch(y, SR, n_mels=40); torch.cuda.synchronize()
for _ in range(300): ops.mfcc_batch(y, SR, n_mels=40)       # clock warm-up
Tn = 94
report("C2 a1-a5 STFT->mel->MFCC (one launch)", lambda: ops.mfcc_batch(y, SR, n_mels=40), B * L, B * (4 * L + 4 * 13 * Tn), n=100)
# the reference's DEFAULT filterbank (manager.py:214: n_mels = 128; features_cmd.py:82-90 passes no feature_params, so
# `sygnals features extract -f mfcc` -- BASELINE config C1 -- always runs 128 bands), and 64 bands
for nm in (128, 64):
    report(f"a1-a5 STFT->mel->MFCC n_fft=2048 n_mels={nm}" + (" (the reference's default)" if nm == 128 else "") + ", 1024 clips",
           lambda: ops.mfcc_batch(y, SR, n_mels=nm), B * L, B * (4 * L + 4 * 13 * Tn), n=50)
if want("C1 one clip"):
    y1 = ops.to_device_f32(synth_clips(1, 160000, 16000, seed=3))
    report("C1 one clip 10 s @ 16 kHz n_mels=128 (device part of `features extract -f mfcc`)", lambda: ops.mfcc_batch(y1, 16000, n_mels=128),
           160000, 4 * 160000 + 4 * 13 * 313, "one clip: launch-bound, one workgroup busy", n=50)
    y1b = ops.to_device_f32(synth_clips(256, 160000, 16000, seed=3))
    report("C1-shaped batch: 256 clips x 10 s @ 16 kHz n_mels=128", lambda: ops.mfcc_batch(y1b, 16000, n_mels=128),
           256 * 160000, 256 * (4 * 160000 + 4 * 13 * 313), n=20)
    del y1, y1b
# the reference's own manager tests: frame_length=1024 with spectral features (tests/test_features_manager.py:58-62,167-174)
if not ONLY or any("1024" in a or "c4-style" in a or "4096" in a for a in ONLY):
    from sygnals_amd.core.features.manager import extract_features_batch
    T4 = 1 + L // 256
    report("a6-a9 n_fft=1024 hop=256: centroid + rolloff through extract_features_batch (device resident), 1024 clips",
           lambda: extract_features_batch(y, SR, ["spectral_centroid", "spectral_rolloff"], 1024, 256, to_host=False), B * L,
           B * (4 * L + 4 * 2 * T4), n=5, warm=2)
    for nf, hp in ((512, 128), (256, 64)):
        report(f"a6-a9 n_fft={nf} hop={hp}: centroid + rolloff through extract_features_batch (device resident), 1024 clips",
               lambda: extract_features_batch(y, SR, ["spectral_centroid", "spectral_rolloff"], nf, hp, to_host=False), B * L,
               B * (4 * L + 4 * 2 * (1 + L // hp)), n=5, warm=2)
    report("a6-a9 n_fft=4096 hop=1024: centroid + rolloff through extract_features_batch (device resident), 1024 clips",
           lambda: extract_features_batch(y, SR, ["spectral_centroid", "spectral_rolloff"], 4096, 1024, to_host=False), B * L,
           B * (4 * L + 4 * 2 * (1 + L // 1024)), n=5, warm=2)
    report("C4-style block n_fft=4096 hop=1024: mfcc(40) + centroid + rolloff + contrast through extract_features_batch, 1024 clips",
           lambda: extract_features_batch(y, SR, ["mfcc", "spectral_centroid", "spectral_rolloff", "spectral_contrast"], 4096, 1024,
                                          feature_params={"mfcc": {"n_mels": 40}}, to_host=False), B * L,
           B * (4 * L + 4 * 22 * (1 + L // 1024)), n=5, warm=2)
    report("C4-style block n_fft=1024 hop=256: mfcc(40) + centroid + rolloff + contrast through extract_features_batch, 1024 clips",
           lambda: extract_features_batch(y, SR, ["mfcc", "spectral_centroid", "spectral_rolloff", "spectral_contrast"], 1024, 256,
                                          feature_params={"mfcc": {"n_mels": 40}}, to_host=False), B * L, B * (4 * L + 4 * 22 * T4), n=5, warm=2)
report("a1 complex STFT 2048/512 (frame-major c64 out)", lambda: ops.stft2048_c2c(y[:256]), 256 * L, 256 * (4 * L + 8 * 1025 * Tn), "256 clips")
# the other power-of-two frame lengths (the reference's tests: 1024 / 256): one launch, clip-resident
for nf, hp in ((1024, 256), (512, 128), (256, 64)):
    Tf = 1 + L // hp
    report(f"a1-a5 STFT->mel->MFCC n_fft={nf} hop={hp} (mfcc_batch default, 1024 clips)", lambda: ops.mfcc_batch(y, SR, nf, hp, n_mels=40), B * L,
           B * (4 * L + 4 * 13 * Tf))
    report(f"a1-a5 n_fft={nf} hop={hp} ONE launch (clip-resident, fused=True)", lambda: ops.mfcc_batch(y, SR, nf, hp, n_mels=40, fused=True), B * L,
           B * (4 * L + 4 * 13 * Tf))
    report(f"a1-a5 n_fft={nf} hop={hp} two launches (tile form + logmel_dct)",
           lambda: ops.logmel_dct(ops.stft_mel_pow2(y, SR, nf, hp, True, "hann", None, 40), 13), B * L, B * (4 * L + 4 * 13 * Tf))
    report(f"a1-a5 n_fft={nf} hop={hp} generic chain of round 2 (STFT -> |X|^2 -> dense mel -> dB/DCT, 4 launches)",
           lambda: ops.mfcc_batch(y, SR, nf, hp, n_mels=40, fused=False), B * L, B * (4 * L + 4 * 13 * Tf), n=5, warm=2)
report("a1-a3 n_fft=1024 hop=256 mel only: segment-sum kernel (free-running waves)", lambda: ops.stft_mel_w1024_seg(y, SR, 256, True, "hann", None, 40),
       B * L, B * (4 * L + 4 * 40 * (1 + L // 256)))
report("a1-a3 n_fft=1024 hop=256 mel only: dense-matrix tile kernel", lambda: ops.stft_mel_pow2(y, SR, 1024, 256, True, "hann", None, 40),
       B * L, B * (4 * L + 4 * 40 * (1 + L // 256)))
for nf, hp in ((512, 128), (256, 64)):
    report(f"a1-a3 n_fft={nf} hop={hp} mel only: segment-sum kernel (free-running waves)", lambda: ops.stft_mel_wseg_small(y, SR, nf, hp, True, "hann", None, 40),
           B * L, B * (4 * L + 4 * 40 * (1 + L // hp)))
    report(f"a1-a3 n_fft={nf} hop={hp} mel only: dense-matrix tile kernel", lambda: ops.stft_mel_pow2(y, SR, nf, hp, True, "hann", None, 40),
           B * L, B * (4 * L + 4 * 40 * (1 + L // hp)))
# frame length 4096 (round 3: one wave per frame, mel by segment sums) against the generic chain
Tf = 1 + L // 1024
report("a1-a5 STFT->mel->MFCC n_fft=4096 hop=1024 (fused mel kernel + logmel_dct, 1024 clips)", lambda: ops.mfcc_batch(y, SR, 4096, 1024, n_mels=40),
       B * L, B * (4 * L + 4 * 13 * Tf))
report("a1-a5 n_fft=4096 hop=1024 generic chain (STFT -> |X|^2 -> dense mel -> dB/DCT, 4 launches)",
       lambda: ops.mfcc_batch(y, SR, 4096, 1024, n_mels=40, fused=False), B * L, B * (4 * L + 4 * 13 * Tf), n=5, warm=2)
# C3: band-pass filtfilt then MFCC
sos = FL.design_butterworth_sos((300.0, 3400.0), SR, 4, "bandpass")
report("a13 sosfiltfilt order-4 band-pass (C3 filter alone)", lambda: FL.apply_sos_filter_batch(sos, y), B * L, B * 8 * L)
report("C3 filtfilt + MFCC", lambda: ops.mfcc_batch(FL.apply_sos_filter_batch(sos, y), SR, n_mels=40), B * L, B * (4 * L + 4 * 13 * Tn),
       "filtered waveform is a non-algorithmic intermediate")
# C4 per-GPU share: 2048 clips, MFCC + centroid + rolloff + contrast -> one [2048, 22, 94] block
if want("C4 share"):
    from sygnals_amd.core.features.manager import feature_block
    y4 = ops.to_device_f32(np.tile(Y, (2048 // 64, 1)))
    for _ in range(100): feature_block(y4, SR)           # (the clock and the allocator settle: 20 calls from cold read 10 % high)
    report("C4 share: MFCC + centroid + rolloff + contrast -> [2048, 22, 94] block", lambda: feature_block(y4, SR), 2048 * L, 2048 * (4 * L + 4 * 22 * Tn), n=100)
    del y4
report("a6-a9 all five spectral statistics + mel, 1024 clips", lambda: ops.stft2048_mel(y, SR, n_mels=40, want_stats=31), B * L, B * (4 * L + 4 * (40 + 5) * Tn))
report("a6-a9 all five spectral statistics, no mel (syg_stft2048_stats_f32), 1024 clips", lambda: ops.stft2048_stats(y, SR, want_stats=31), B * L, B * (4 * L + 4 * 5 * Tn))
report("f-1 time-domain frame features (9 rows), 1024 clips", lambda: ops.frame_stats(y, 2048, 512, True), B * L, B * (4 * L + 4 * 9 * Tn))
# a10: batched FFT
if want("a10 complex FFT"):
    xf = torch.randn((4096, 4096, 2), dtype=torch.float32, device=y.device)
    report("a10 complex FFT n=4096 x 4096 signals", lambda: ops.fft_pow2(xf), 4096 * 4096, 2 * 8 * 4096 * 4096)
    x48 = torch.randn((1024, 48000, 2), dtype=torch.float32, device=y.device)
    report("a10 complex FFT n=48000 (2^7 3 5^3) x 1024 signals", lambda: ops.fft_any(x48), 1024 * 48000, 2 * 8 * 1024 * 48000, n=10, warm=3)
    x6k = torch.randn((8192, 6000, 2), dtype=torch.float32, device=y.device)
    report("a10 complex FFT n=6000 (2^4 3 5^3) x 8192 signals", lambda: ops.fft_any(x6k), 8192 * 6000, 2 * 8 * 8192 * 6000, n=10, warm=3)
    del xf, x48, x6k
# C5: ONE 1-hour stream (172.8 M samples), generated on the device: noise + a slow chirp + tones (tone phases in float64)
if want("C5 a14 Welch") or want("C5 a15 CQT") or want("C5 both"):
    Ls = 48000 * 3600
    g = torch.Generator(device="cuda").manual_seed(5)
    stream = torch.randn(Ls, device="cuda", generator=g, dtype=torch.float32) * 0.05
    for c0 in range(0, Ls, 1 << 24):
        tt = torch.arange(c0, min(c0 + (1 << 24), Ls), device="cuda", dtype=torch.float64)
        stream[c0:c0 + tt.numel()] += (0.3 * torch.sin(2 * np.pi * (30.0 + 12000.0 * tt / Ls / 2) / SR * tt)
                                       + 0.2 * torch.sin(2 * np.pi * 3000.5 / SR * tt)).float()
    del tt
    stream = stream.reshape(1, Ls)
    report("C5 a14 Welch nperseg 4096 / 50 %, one 1-hour stream", lambda: D.welch_batch(stream, fs=SR, nperseg=4096), Ls, 4 * Ls + 4 * 2049,
           "input read in its own pass (CQT separate)", n=5, warm=2)
    report("C5 a15 CQT 84 bins hop 512, one 1-hour stream", lambda: ops.cqt(stream, SR), Ls, 4 * Ls + 8 * 84 * (1 + Ls // 512),
           "input read in its own pass (Welch separate)", n=3, warm=1)
    # C5 as ONE configuration: the same stream through both, Welch on a second HIP stream beside the CQT (the Welch wave
    # kernel holds two waves per SIMD in 190-245 registers and is bound by vector issue; the CQT's decimation chain waits on
    # memory and its octave products on the matrix pipe: they share the CUs)
    if want("C5 both"):
        side = ops._side_stream(stream.device)
        def both():
            main = torch.cuda.current_stream()
            side.wait_stream(main)
            with torch.cuda.stream(side):
                w = D.welch_batch(stream, fs=SR, nperseg=4096)
            c = ops.cqt(stream, SR)
            main.wait_stream(side)
            return w, c
        report("C5 both: Welch on a second stream beside the CQT, one 1-hour stream", both, Ls,
               4 * Ls + 4 * 2049 + 8 * 84 * (1 + Ls // 512), "the stream is read by both; algorithmic bytes counted once", n=3, warm=1)
        def serial():
            return D.welch_batch(stream, fs=SR, nperseg=4096), ops.cqt(stream, SR)
        report("C5 both, one after the other on one stream", serial, Ls, 4 * Ls + 4 * 2049 + 8 * 84 * (1 + Ls // 512), "", n=3, warm=1)
    del stream
# f-3: FFT-backed 1-D operations on the 1024-clip batch
if want("f-3 convolution autocorrelation Hilbert periodogram"):
    kern = ops.to_device_f32((np.random.default_rng(0).normal(0, 1, 1023) / 32).astype(np.float32))
    report("f-3 convolution, 1023-tap shared kernel, mode=same, 1024 clips", lambda: D.convolve_batch(y, kern, "same"), B * L, B * 8 * L)
    report("f-3 autocorrelation (full), 1024 clips", lambda: D.convolve_batch(y, y, "full", correlate=True), B * L, B * (4 * L + 4 * (2 * L - 1)))
    y65 = torch.randn((1024, 65536), dtype=torch.float32, device=y.device)
    report("f-3 Hilbert envelope, 1024 rows x 65536", lambda: D.envelope_batch(y65), 1024 * 65536, 1024 * 8 * 65536)
    report("f-3 Hilbert envelope, 1024 clips x 48000 (mixed radix 200 x 240)", lambda: D.envelope_batch(y), B * L, B * 8 * L, n=5, warm=2)
    report("f-3 periodogram, 1024 clips x 48000 (mixed radix 200 x 240)", lambda: D.periodogram_batch(y, fs=SR), B * L, B * (4 * L + 4 * (L // 2 + 1)), n=5, warm=2)
out = os.environ.get("ROWS_OUT", "gpurun_out/rows_r04.json")
os.makedirs(os.path.dirname(out) or ".", exist_ok=True)
json.dump(rows, open(out, "w"), indent=1)
