import sys
import numpy as np, scipy.signal
sys.path.insert(0, ".")
from oracle import cpu_ref as O
from sygnals_amd.synth import synth_stream
def halfband(ntaps, beta):
    h = scipy.signal.firwin(ntaps, 0.5, window=("kaiser", beta))
    h[np.abs(h) < 1e-15 * np.abs(h).max()] = 0.0
    return h
def cqt_with(y, sr, taps):
    def res2(x):
        n = int(np.ceil(x.shape[-1] * 0.5))
        full = np.convolve(np.asarray(x, np.float64), taps)[(len(taps) - 1) // 2:]
        z = full[::2][:n]
        if z.shape[-1] < n: z = np.pad(z, (0, n - z.shape[-1]))
        return z * np.sqrt(2.0)
    old = O.cqt_resample2; O.cqt_resample2 = res2
    try: return O.cqt(y, sr)
    finally: O.cqt_resample2 = old
sr=48000; secs=8.0
rng=np.random.default_rng(1)
sigs={"C5 recipe": synth_stream(int(secs*sr), sr).astype(np.float64), "white": rng.normal(0,0.2,int(secs*sr))}
REFS={"ref301_b15": halfband(301,15.0), "ref601_b16": halfband(601,16.0)}
CAND={f"k{n}_b{b}": halfband(n,b) for n,b in [(41,10.0),(57,12.0),(65,13.0),(81,13.0),(97,14.0),(129,14.0)]}
for sname,y in sigs.items():
    R={k:np.abs(cqt_with(y,sr,h)) for k,h in REFS.items()}
    edge=int(1.5*sr/512)
    pk=R["ref301_b15"].max()
    print(sname, "ref301 vs ref601:", np.abs(R["ref301_b15"]-R["ref601_b16"])[:,edge:-edge].max()/pk)
    for name,h in CAND.items():
        C=np.abs(cqt_with(y,sr,h))
        print(f"  {name:10s} nonzero {int((h!=0).sum()):3d}  vs ref301 {np.abs(C-R['ref301_b15'])[:,edge:-edge].max()/pk:.2e}  vs ref601 {np.abs(C-R['ref601_b16'])[:,edge:-edge].max()/pk:.2e}")
