"""Randomised shape fuzz of round 4's kernels against the float64 oracle and against the forms they must agree with
(development aid): the tile form of the segment-sum projection (MODE 8 ... 11: two- / four-pass tables, 16 / 8 waves, with
and without row functions, any hop / alignment), the row functions inside the frame-length 1024 / 512 / 256 / 4096 kernels.
    python3 tools/fuzz_round4.py [seed [n]]"""
import sys
import numpy as np, torch
sys.path.insert(0, ".")
from sygnals_amd import ops, _tables as T
from oracle import cpu_ref as O
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
N = int(sys.argv[2]) if len(sys.argv) > 2 else 60
worst = {"mel segments vs oracle": 0.0, "mel segments vs matrix": 0.0, "rows 2048 tile form vs stats kernel (bits)": 0.0,
         "rows other frame lengths vs oracle": 0.0}
n_tri = n_rows = 0
for it in range(N):
    B = int(rng.choice([1, 2, 3, 5, 9, 40]))
    sr = int(rng.choice([8000, 16000, 22050, 32000, 44100, 48000]))
    center = bool(rng.integers(0, 2))
    off = int(rng.integers(0, 4))
    # ---- 2048: tile-form segment kernels
    hop = int(rng.choice([1, 7, 128, 160, 256, 441, 512, 513, 1024]))
    L = int(rng.integers(2048 if not center else 1, 24000 if hop >= 128 else 5000))
    n_mels = int(rng.choice([24, 40, 64, 96, 128, 160]))
    waves = int(rng.choice([8, 16]))
    Y = (rng.normal(0, 0.2, (B, L + 8)) * rng.random((B, 1)) + 0.1 * np.sin(np.arange(L + 8) * 0.05)[None, :]).astype(np.float32)
    y = ops.to_device_f32(Y)[:, off:off + L]
    cfg = ops.mel_config(sr, 2048, n_mels)
    if (cfg.segtab is not None or cfg.segtab4 is not None) and ops.num_frames(L, 2048, hop, center) > 0:
        n_tri += 1
        want_rows = bool(rng.integers(0, 2))
        cplan = T.contrast_plan(np.fft.rfftfreq(2048, 1.0 / sr), sr) if (want_rows and sr >= 32000) else None
        smask = int(rng.choice([1, 9, 31])) if want_rows else 0
        a, sa, pa = ops.stft2048_mel(y, sr, hop, center, "hann", 2048, n_mels, 0.0, None, smask, 0.85, 2.0, cplan, "segments", waves)
        m, _, _ = ops.stft2048_mel(y, sr, hop, center, "hann", 2048, n_mels, 0.0, None, 0, 0.85, 2.0, None, "matrix")
        a_, m_ = a.cpu().numpy(), m.cpu().numpy()
        pk = max(float(np.abs(m_).max()), 1e-30)
        worst["mel segments vs matrix"] = max(worst["mel segments vs matrix"], float(np.abs(a_ - m_).max()) / pk)
        assert np.abs(a_ - m_).max() <= 1e-5 * pk, (it, "segments vs matrix", B, L, hop, center, n_mels, sr, off, waves)
        S = np.abs(O.stft(Y[0, off:off + L].astype(np.float64), 2048, hop, center=center)) ** 2
        ref = O.melspectrogram(S, sr, 2048, n_mels)
        e = float(np.abs(a_[0] - ref).max()) / max(float(ref.max()), 1e-30)
        worst["mel segments vs oracle"] = max(worst["mel segments vs oracle"], e)
        assert e <= 1e-5, (it, "segments vs oracle", B, L, hop, center, n_mels, sr, off, waves, e)
        if want_rows and hop <= 512 and ops.stft2048_stats_fits(hop, L):
            sb, pb = ops.stft2048_stats(y, sr, hop, center, want_stats=smask, contrast=cplan)
            assert torch.equal(sa, sb), (it, "statistics rows differ", B, L, hop, center, sr, smask)
            if cplan is not None:
                assert torch.equal(pa, pb), (it, "contrast rows differ", B, L, hop, center, sr)
    # ---- 1024 / 512 / 256 / 4096: the rows kernels
    nf = int(rng.choice([1024, 512, 256, 4096]))
    hop2 = int(rng.choice([nf // 8, nf // 4, nf // 2, 100, nf]))
    L2 = int(rng.integers(nf if not center else 1, 9000 if nf < 4096 else 30000))
    if ops.num_frames(L2, nf, hop2, center) > 0:
        n_rows += 1
        Y2 = (rng.normal(0, 0.3, (B, L2)) * rng.random((B, 1))).astype(np.float32)
        y2 = ops.to_device_f32(Y2)
        fr = O.fft_frequencies(sr, nf)
        _, st, _ = ops.stft_rows_seg(y2, sr, nf, hop2, center, "hann", None, None, 0.0, None, 31, 0.85, 2.0, None)
        st = st.cpu().numpy()
        Sm = np.abs(O.stft(Y2[0].astype(np.float64), nf, hop2, center=center))
        ref = O.spectral_stats_frames(Sm, fr)
        from tests.gpu_util import assert_flatness_parity
        assert_flatness_parity(st[0, 2], ref["spectral_flatness"], Sm, 1e-5, f"flatness {(it, nf, hop2, L2, center, sr)}")
        for row, key, tol in ((0, "spectral_centroid", 1e-5), (1, "spectral_bandwidth", 1e-5)):
            e = float(np.abs(st[0, row] - ref[key]).max()) / max(float(np.abs(ref[key]).max()), 1e-30)
            worst["rows other frame lengths vs oracle"] = max(worst["rows other frame lengths vs oracle"], e)
            if e > tol:
                d = np.abs(st[0, row] - ref[key]); w = int(d.argmax())
                print("DIAG", it, key, nf, hop2, L2, center, sr, "err", e, "worst frame", w, "of", len(d), "device", st[0, row][w], "ref", ref[key][w],
                      "clip rms", float(np.sqrt(np.mean(Y2[0].astype(np.float64) ** 2))), "frame min/max |S|", Sm[:, w].min(), Sm[:, w].max(),
                      "frames off", np.nonzero(d > tol * np.abs(ref[key]).max())[0][:12])
            assert e <= tol, (it, key, nf, hop2, L2, center, sr, e)
        sure = ref["rolloff_margin"] > 1e-6
        assert (st[0, 3].astype(int)[sure] == ref["rolloff_bin"][sure]).all(), (it, "rolloff", nf, hop2, L2, center, sr)
print(f"fuzz ok: {N} rounds ({n_tri} tile-form segment launches, {n_rows} row-kernel launches); worst peak-relative errors:",
      {k: float("%.3g" % v) for k, v in worst.items()})
