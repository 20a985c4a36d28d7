"""Per-wave phase timeline of the fused kernel (needs a -DSYG_ABL=9 build; development aid)."""
import sys
import numpy as np, torch
sys.path.insert(0, ".")
from sygnals_amd import ops
from oracle import cpu_ref as O
B = 1024
Y = O.synth_clips(32, 48000, 48000, seed=1)
y = ops.to_device_f32(np.tile(Y, (B // 32, 1)))
MODE3 = len(sys.argv) > 1 and sys.argv[1] == "mfcc"
for _ in range(2):
    if MODE3:
        st = ops.stft2048_mfcc(y, 48000, n_mels=40, keep_mel=True)[1]
    else:
        res = ops.stft2048_mel(y, 48000, n_mels=40, want_stats=True); st = res[1]
torch.cuda.synchronize()
W = ops.fused_waves()
d = st.reshape(-1)[: 256 * W * 16].reshape(256, W, 16).cpu().numpy()[:, :, :12]
names = ["load+window", "pass1+tw1", "exchange1", "pass2+tw2", "exchange2+pass3", "split+P store", "barrier A",
         "MFMA", "barrier B", "reduce+store", "clip DCT (MODE 3)", "-"]
tot = d.sum(axis=2)
print(f"cycles per tile per wave: mean {tot.mean():.0f}  min {tot.min():.0f} max {tot.max():.0f}")
for i, n in enumerate(names[:11]):
    col = d[:, :, i]
    print(f"{n:18s} mean {col.mean():8.0f}  ({100*col.mean()/tot.mean():5.1f}%)   wave0 {d[:, 0, i].mean():8.0f}  wave{W-1} {d[:, W-1, i].mean():8.0f}")
