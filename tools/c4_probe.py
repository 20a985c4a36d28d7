import sys, numpy as np, torch
sys.path.insert(0, ".")
from sygnals_amd import ops, _tables as T
from sygnals_amd.synth import synth_clips
B=1024
Y=synth_clips(32,48000,48000,seed=1); y=ops.to_device_f32(np.tile(Y,(B//32,1)))
fr=np.fft.rfftfreq(2048,1/48000)
CP=T.contrast_plan(fr,48000)
print("plan", CP[:1], CP[1:8], CP[17:24], CP[33:40])
def sub(rows):
    p=np.zeros_like(CP); p[0]=len(rows)
    for i,r in enumerate(rows):
        p[1+i]=CP[1+r]; p[1+16+i]=CP[1+16+r]; p[1+32+i]=CP[1+32+r]
    return p
def t(fn,n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/n*1e3
for _ in range(200): ops.stft2048_mel(y,48000,n_mels=40)
print("mel only", t(lambda: ops.stft2048_mel(y,48000,n_mels=40)))
print("centroid", t(lambda: ops.stft2048_mel(y,48000,n_mels=40,want_stats=1)))
print("rolloff", t(lambda: ops.stft2048_mel(y,48000,n_mels=40,want_stats=8)))
print("cen+roll", t(lambda: ops.stft2048_mel(y,48000,n_mels=40,want_stats=9)))
for name,rows in (("band6 only",[6]),("band5 only",[5]),("band0 only",[0]),("bands0-4",[0,1,2,3,4]),("bands0-5",[0,1,2,3,4,5]),("all7",list(range(7)))):
    p=sub(rows)
    print("contrast", name, t(lambda: ops.stft2048_mel(y,48000,n_mels=40,contrast=p)))
print("C4 all", t(lambda: ops.stft2048_mel(y,48000,n_mels=40,want_stats=9,contrast=CP)))
