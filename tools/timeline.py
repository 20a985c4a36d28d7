"""Per-wave phase timeline of the fused kernel (needs a -DSYG_DEV=1 build, loaded through SYGNALS_AMD_LIB +
SYGNALS_AMD_ALLOW_VARIANT=9; development aid).  Variants: mfcc | all | cen | cenroll | c4 | band0 (matrix-form kernels),
c4one | c4one_band0 | c4one_cen (the one-launch C4 kernel, MODE 7)."""
import sys
import numpy as np, torch
sys.path.insert(0, ".")
from sygnals_amd import ops
from oracle import cpu_ref as O
B = 1024
Y = O.synth_clips(32, 48000, 48000, seed=1)
y = ops.to_device_f32(np.tile(Y, (B // 32, 1)))
# variants: mfcc (MODE 3) | all (MODE 1, five statistics) | cen | cenroll | c4 (centroid + rolloff + contrast) | band0
VAR = sys.argv[1] if len(sys.argv) > 1 else "all"
MODE3 = VAR == "mfcc"
from sygnals_amd import _tables as T
CP = T.contrast_plan(np.fft.rfftfreq(2048, 1 / 48000), 48000)
CP0 = np.zeros_like(CP); CP0[0] = 1; CP0[1] = CP[1]; CP0[17] = CP[17]; CP0[33] = CP[33]
ONE = VAR.startswith("c4one")
if ONE:
    import ctypes as C
    from sygnals_amd._lib import check, lib
    cfg = ops.mel_config(48000, 2048, 40, 0.0, None, waves=16)
    dct = ops._dev(T.dct_matrix(13, 40, 2, "ortho"))
    plan = {"c4one": CP, "c4one_band0": CP0, "c4one_cen": None}[VAR]
    smask = 1 | 32 if VAR != "c4one" else 1 | 8 | 32
    R = int(plan[0]) if plan is not None else 1
    out = torch.empty((B, 13 + 2 + R, 94), dtype=torch.float32, device="cuda")
    st = torch.zeros((B, 8, 94), dtype=torch.float32, device="cuda")
    cpv = torch.empty((B, 2, R, 94), dtype=torch.float32, device="cuda")
for _ in range(2):
    if ONE:
        check(lib().syg_stft2048_features_tri_f32(
            ops._ptr(y), B, 48000, y.stride(0), 512, 1, 94, ops._ptr(ops.window_dev("hann", 2048, 2048)), ops._ptr(ops.twiddle_dev(2048)),
            ops._ptr(cfg.segtab), int(cfg.segtab.numel()), 40, ops._ptr(dct), 13, None, 1e-10, 80.0, 1, 1.0, 48000.0, 0.85, 2.0, smask,
            ops._ptr(st), plan.ctypes.data_as(C.c_void_p) if plan is not None else None, ops._ptr(cpv) if plan is not None else None,
            ops._ptr(out), 13 + 2 + R, C.c_void_p(ops._stream_ptr())), "features_tri")
    elif MODE3:
        st = ops.stft2048_mfcc(y, 48000, n_mels=40, keep_mel=True)[1]
    elif VAR == "all":
        res = ops.stft2048_mel(y, 48000, n_mels=40, want_stats=True); st = res[1]
    elif VAR == "cen":
        st = ops.stft2048_mel(y, 48000, n_mels=40, want_stats=1)[1]
    elif VAR == "cenroll":
        st = ops.stft2048_mel(y, 48000, n_mels=40, want_stats=9)[1]
    elif VAR == "band0":
        st = ops.stft2048_mel(y, 48000, n_mels=40, want_stats=1, contrast=CP0)[1]
    else:
        st = ops.stft2048_mel(y, 48000, n_mels=40, want_stats=9, contrast=CP)[1]
torch.cuda.synchronize()
W = ops.fused_waves()
d = st.reshape(-1)[: 256 * W * 16].reshape(256, W, 16).cpu().numpy()[:, :, :12]
names = ["load+window", "pass1+tw1", "exchange1", "pass2+tw2", "exchange2+pass3", "split+P store", "barrier A",
         "MFMA", "barrier B", "reduce+store", "clip DCT (MODE 3)", "row functions (MODE 1)"]
if ONE:
    names = ["load+window", "pass1+tw1", "exchange1", "pass2+tw2", "exchange2+pass3", "split+P store", "projection / clip epilogue",
             "row functions (early half)", "wait at X1", "fetch next + wait at X2", "row functions (late half)", "stores + stage refill"]
tot = d.sum(axis=2)
if len(sys.argv) > 2 and sys.argv[2] == "waves":      # per-wave means of the barrier waits and the whole tile
    print("per-wave: wave | barrier A | barrier B | reduce | total")
    for wv in range(W):
        print(f"  {wv:2d}  {d[:, wv, 6].mean():8.0f} {d[:, wv, 8].mean():8.0f} {d[:, wv, 9].mean():8.0f} {tot[:, wv].mean():8.0f}")
print(f"cycles per tile per wave: mean {tot.mean():.0f}  min {tot.min():.0f} max {tot.max():.0f}")
for i, n in enumerate(names[:12]):
    col = d[:, :, i]
    print(f"{n:18s} mean {col.mean():8.0f}  ({100*col.mean()/tot.mean():5.1f}%)   wave0 {d[:, 0, i].mean():8.0f}  wave{W-1} {d[:, W-1, i].mean():8.0f}")
