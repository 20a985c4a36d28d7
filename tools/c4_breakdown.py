"""Where the C4 launch (stft2048_kernel<16,2,7>) spends its time beyond the MFCC path: the same launch with parts of the
row functions left out (statistics only, contrast only, the contrast plan cut to its first bands).
    python3 tools/c4_breakdown.py [B [n_cases]]   -> microseconds per launch, B clips (default 2048)"""
import ctypes as C, sys
import numpy as np, torch
sys.path.insert(0, ".")
from sygnals_amd import ops, _tables as T
from sygnals_amd._lib import check, lib
from sygnals_amd.synth import synth_clips

B = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
sr, hop, n_mels, n_mfcc = 48000, 512, 40, 13
Y = synth_clips(64, 48000, 48000, seed=1)
y = ops.to_device_f32(np.tile(Y, (B // 64, 1)))
L = y.shape[1]
Tn = ops.num_frames(L, 2048, hop, True)
cfg = ops.mel_config(sr, 2048, n_mels, 0.0, None, waves=16)
dct = ops._dev(T.dct_matrix(n_mfcc, n_mels, 2, "ortho"))
full = T.contrast_plan(np.fft.rfftfreq(2048, 1 / sr), sr)
MB = (len(full) - 1) // 3


def cut(nb):
    p = full.copy(); p[0] = nb
    return np.ascontiguousarray(p, np.int32)


def only(b):                     # the plan with band b alone
    p = np.zeros_like(full); p[0] = 1
    p[1], p[1 + MB], p[1 + 2 * MB] = full[1 + b], full[1 + MB + b], full[1 + 2 * MB + b]
    return np.ascontiguousarray(p, np.int32)


def timed(stats_mask, plan):
    R = int(plan[0]) if plan is not None else 1
    rows = n_mfcc + 2 + R
    out = torch.empty((B, rows, Tn), dtype=torch.float32, device="cuda")
    stats = torch.empty((B, 8, Tn), dtype=torch.float32, device="cuda")
    cpv = torch.empty((B, 2, R, Tn), dtype=torch.float32, device="cuda")
    head = (ops._ptr(y), B, L, y.stride(0), hop, 1, Tn, ops._ptr(ops.window_dev("hann", 2048, 2048)), ops._ptr(ops.twiddle_dev(2048)))
    st = C.c_void_p(ops._stream_ptr())
    if stats_mask == 0 and plan is None:
        def run():
            check(lib().syg_stft2048_mfcc_tri_f32(*head, ops._ptr(cfg.segtab), int(cfg.segtab.numel()), n_mels, ops._ptr(dct), n_mfcc,
                                                  None, 1e-10, 80.0, 1, 1.0, ops._ptr(out), st), "mfcc_tri")
    else:
        tail = (n_mels, ops._ptr(dct), n_mfcc, None, 1e-10, 80.0, 1, 1.0, float(sr), 0.85, 2.0, stats_mask or 1,
                ops._ptr(stats) if stats_mask else None, plan.ctypes.data_as(C.c_void_p) if plan is not None else None,
                ops._ptr(cpv) if plan is not None else None)

        def run():
            check(lib().syg_stft2048_features_tri_f32(*head, ops._ptr(cfg.segtab), int(cfg.segtab.numel()), *tail, ops._ptr(out), rows, st), "features_tri")
    for _ in range(100): run()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(4):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50): run()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1000 / 50)
    return best


cases = [("mfcc only (MODE 6)", 0, None), ("+ centroid", 1 | 32, None), ("+ rolloff", 8 | 32, None), ("+ centroid + rolloff", 1 | 8 | 32, None),
         ("+ contrast, all bands", 0, cut(int(full[0])))]
for nb in range(1, int(full[0])):
    cases.append((f"+ contrast, bands 0..{nb - 1}", 0, cut(nb)))
for b in range(int(full[0])):
    cases.append((f"+ contrast, band {b} alone [{full[1 + b]}, {full[1 + MB + b]}) k={full[1 + 2 * MB + b]}", 0, only(b)))
cases.append(("+ centroid + rolloff + contrast (C4)", 1 | 8 | 32, cut(int(full[0]))))
if len(sys.argv) > 2: cases = cases[:int(sys.argv[2])]
for name, sm, plan in cases:
    print(f"{name:64s} {timed(sm, plan):8.1f} us")
