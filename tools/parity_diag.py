"""Error distributions behind the margin-qualified parity gates (development aid; runs on the GPU box).

Prints, per case, how the fp32 device error relates to the fp32 FFT noise floor of the frame:
  * spectral contrast: dB error against valley / floor;
  * C3 (filtfilt -> MFCC): errors of mel power, log-mel cells and MFCCs, and the propagated noise budget;
  * pure-tone MFCC, flatness.
"""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from oracle import cpu_ref as O
from sygnals_amd import ops, _tables as T

EPS32 = 2.0 ** -24
fr = O.fft_frequencies(48000, 2048)


def contrast_case(clips, tag):
    plan = T.contrast_plan(fr, 48000)
    y = ops.to_device_f32(clips)
    _, _, pvd = ops.stft2048_mel(y, 48000, n_mels=40, contrast=plan)
    cdb = ops.contrast_db(pvd).cpu().numpy()
    pv = pvd.cpu().numpy()
    rows = []
    for i in range(clips.shape[0]):
        S = np.abs(O.stft(clips[i].astype(np.float64), 2048, 512))
        C = O.spectral_contrast(S, 48000, freqs=fr)
        bands = O.contrast_bands(fr, 48000)
        floor = EPS32 * np.linalg.norm(S, axis=0)                # per frame
        for k, (bins, kk) in enumerate(bands):
            srt = np.sort(S[bins], axis=0)
            val = srt[:kk].mean(axis=0)
            pk = srt[-kk:].mean(axis=0)
            err = np.abs(cdb[i, k] - C[k])
            lin = np.abs(pv[i, 1, k] - val)
            for t in range(S.shape[1]):
                rows.append((err[t], val[t] / floor[t], lin[t] / floor[t], np.abs(C).max(), k))
    r = np.array(rows)
    cmax = r[:, 3].max()
    print(f"[contrast {tag}] n={len(r)} max dB err {r[:,0].max():.3e} (rel {r[:,0].max()/cmax:.2e}); "
          f"max linear valley err / floor = {r[:,2].max():.3f}")
    M = (10 / np.log(10)) / (1e-5 * cmax)
    for mult in (0.25, 0.5, 1, 2, 4):
        sure = r[:, 1] >= M * mult
        bad = (r[:, 0] > 1e-5 * cmax) & sure
        print(f"   valley >= {mult} x M floors (M={M:.0f}): sure {sure.mean()*100:.1f}% of cells, violations {bad.sum()}, "
              f"max sure err rel {r[sure,0].max()/cmax if sure.any() else 0:.2e}")
    # ratio of the dB error to the propagated bound 4.34 * floor / valley
    bound = (10 / np.log(10)) / r[:, 1]
    print(f"   max err / (4.34 floor/valley) = {(r[:,0] / bound).max():.3f}; 99.9% {np.quantile(r[:,0]/bound, 0.999):.3f}")
    for k in range(7):
        m = r[:, 4] == k
        print(f"   band {k}: max rel err {r[m,0].max()/cmax:.2e}  median valley/floor {np.median(r[m,1]):.0f}")


def mfcc_budget_case(clips64, tag, dev_in=None):
    """clips64: float64 signals handed to the oracle; dev_in: float32 device tensor handed to the device path."""
    y = ops.to_device_f32(clips64.astype(np.float32)) if dev_in is None else dev_in
    mfd, meld = ops.stft2048_mfcc(y, 48000, n_mels=40, n_mfcc=13, keep_mel=True)
    mfd = mfd.cpu().numpy(); meld = meld.cpu().numpy()
    D = T.dct_matrix(13, 40).astype(np.float64)
    for i in range(clips64.shape[0]):
        S = np.abs(O.stft(clips64[i], 2048, 512))
        mel = O.melspectrogram(S ** 2, 48000, 2048, 40, 0.0, 24000.0)
        ldb = O.power_to_db(mel, ref=np.max)
        mf = O.mfcc(S=ldb, sr=48000, n_mfcc=13)
        e_mel = np.abs(meld[i] - mel).max() / mel.max()
        ldb_dev = O.power_to_db(meld[i].astype(np.float64), ref=np.max)
        e_ldb = np.abs(ldb_dev - ldb)
        e_mf = np.abs(mfd[i] - mf)
        pk = np.abs(mf).max()
        # noise budget: power error of a mel cell <= 2 sqrt(P_bin) * floor summed with the weights; estimate with
        # floor_t = EPS32 ||S_t||_2 on every bin: d mel[m,t] <= sum_f w[m,f] (2 S[f,t] floor_t + floor_t^2)
        W = T.mel_filterbank(48000, 2048, 40, 0.0, 24000.0).astype(np.float64)
        floor = EPS32 * np.linalg.norm(S, axis=0)
        dmel = W @ (2 * S * floor[None, :] + floor[None, :] ** 2)
        live = ldb > ldb.max() - 80.0 + 1e-9            # cells above the top_db clamp
        bdb = np.where(live, (10 / np.log(10)) * dmel / np.maximum(mel, 1e-300), 0.0)
        # a clamped cell can become live on the device when mel + dmel crosses the floor: bounded by the same term
        near = (~live) & (10 * np.log10(np.maximum(mel + dmel, 1e-10) / mel.max()) > -80.0)
        bdb = np.where(near, (10 / np.log(10)) * dmel / np.maximum(mel, 1e-300), bdb)
        budget = np.abs(D) @ bdb
        ratio = (e_mf - 1e-5 * pk) / np.maximum(budget, 1e-300)
        print(f"[{tag} clip {i}] mel lin err {e_mel:.2e} | logmel max err {e_ldb.max():.3e} dB (cells>1e-5*pk: "
              f"{(e_ldb > 1e-5 * np.abs(ldb).max()).mean()*100:.1f}%) | mfcc rel err {e_mf.max()/pk:.2e} "
              f"(cells over 1e-5: {(e_mf > 1e-5*pk).sum()}/{e_mf.size}) | max (err-1e-5pk)/budget {ratio.max():.3f} "
              f"| logmel err/bound max {np.max(e_ldb[live] / np.maximum(bdb[live], 1e-300)):.3f}")


if __name__ == "__main__":
    torch.cuda.set_device(0)
    clips = O.synth_clips(8, 48000, 48000, seed=20250523)
    contrast_case(clips, "synth seed 20250523")
    contrast_case(O.synth_clips(8, 48000, 48000, seed=21), "synth seed 21")
    # C2 (sanity: budget unused)
    mfcc_budget_case(clips[:4].astype(np.float64), "C2")
    # C3
    sos = O.design_butterworth_sos((300.0, 3400.0), 48000, 4, "bandpass")
    from sygnals_amd.core.filters import apply_sos_filter_batch
    Y = O.synth_clips(16, 48000, 48000, seed=31)
    sel = [0, 9, 10, 13]
    yf_dev = apply_sos_filter_batch(sos, ops.to_device_f32(Y[sel]))
    yf64 = np.stack([O.apply_sos_filter(sos, Y[i].astype(np.float64)) for i in sel])
    print("filter output rel err:", float(np.abs(yf_dev.cpu().numpy() - yf64).max() / np.abs(yf64).max()))
    mfcc_budget_case(yf64, "C3 (device filter -> device mfcc)", dev_in=yf_dev)
    mfcc_budget_case(yf64, "C3' (float64 filter rounded to f32 -> device mfcc)")
    # pure tone
    t = np.arange(48000) / 48000.0
    tone = (0.8 * np.sin(2 * np.pi * 1000.0 * t))[None].astype(np.float32)
    mfcc_budget_case(tone.astype(np.float64), "pure 1 kHz tone")
    # flatness
    y = ops.to_device_f32(clips)
    _, st, _ = ops.stft2048_mel(y, 48000, n_mels=40, want_stats=True)
    st = st.cpu().numpy()
    worst = 0
    for i in range(8):
        S = np.abs(O.stft(clips[i].astype(np.float64), 2048, 512))
        ref = O.spectral_stats_frames(S, fr)
        worst = max(worst, np.abs(st[i, 2] - ref["spectral_flatness"]).max() / ref["spectral_flatness"].max())
    print(f"flatness worst peak-rel err {worst:.2e}")

    # CQT: device vs the oracle restatement, per bin (the long-stream tests used 1e-3)
    rng = np.random.default_rng(5)
    sr = 48000
    n = sr * 8
    x = rng.normal(0, 0.05, n)
    tt = np.arange(n)
    x += 0.3 * np.sin(2 * np.pi * 440 / sr * tt) + 0.2 * np.sin(2 * np.pi * 3000.5 / sr * tt)
    x = x.astype(np.float32)
    ref = O.cqt(x[: sr * 4].astype(np.float64), sr)
    for name, sig in (("8 s stream", x), ("4 s excerpt", x[: sr * 4])):
        C = ops.cqt(ops.to_device_f32(sig[None]), sr).cpu().numpy()[0]
        C = C[..., 0] + 1j * C[..., 1]
        pk = np.abs(ref).max()
        d = np.abs(C[:, :300] - ref[:, :300])
        print(f"[cqt {name}] peak-rel err frames<300: {d.max()/pk:.2e}; frames<100: {d[:, :100].max()/pk:.2e}")
        print("   per-bin log10 err:", np.round(np.log10(d.max(axis=1) / pk + 1e-30), 1).tolist())
        print("   worst frame per octave:", [int(np.argmax(d[12 * o:12 * o + 12].max(axis=0))) for o in range(7)])
