"""GPU parity of the per-frame row functions (spectral statistics, contrast tail means) inside the fused kernels of the
frame lengths other than 2048 -- the reference's own manager tests call extract_features with frame_length=1024 and
spectral features (tests/test_features_manager.py:58-62, 167-174); 512 / 256 are the frame lengths of its short-signal
cases (:183-220).  Calls go through the C ABI (sygnals_amd.ops -> libsygnals_hip.so)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")
from oracle import cpu_ref as O
from tests.gpu_util import assert_contrast_parity, assert_flatness_parity, assert_parity, fft_floor, peak_rel

TOL = 1e-5


@pytest.fixture(scope="module")
def ops():
    from sygnals_amd import ops
    ops.require_gpu()
    return ops


def check_rows(ops, n_fft, hop, sr, L, center, n_mels, B=5, window="hann"):
    from sygnals_amd import _tables as T
    Y = O.synth_clips(B, L, sr, seed=n_fft + L)
    if B >= 4:
        Y[B - 2] *= 1e-3
        Y[B - 1][:] = 0.0
    y = ops.to_device_f32(Y)
    fr = O.fft_frequencies(sr, n_fft)
    F = n_fft // 2 + 1
    plan = T.contrast_plan(fr, sr)
    mel, st, pv = ops.stft_rows_seg(y, sr, n_fft, hop, center, window, None, n_mels, 0.0, None, 31, 0.85, 2.0, plan)
    st, pv = st.cpu().numpy(), pv.cpu().numpy()
    cdb = ops.contrast_db(torch.from_numpy(pv).cuda()).cpu().numpy()
    bands = O.contrast_bands(fr, sr)
    for i in range(B):
        S = np.abs(O.stft(Y[i].astype(np.float64), n_fft, hop, window=window, center=center))
        assert st.shape[2] == S.shape[1]
        if S.max() == 0:
            assert (st[i, 0] == 0).all() and (st[i, 1] == 0).all() and (st[i, 2] == 0).all()
            assert (st[i, 3] == F - 1).all() and (st[i, 4] == 0).all() and (pv[i] == 0).all()
            continue
        ref = O.spectral_stats_frames(S, fr)
        assert_parity(st[i, 0], ref["spectral_centroid"], TOL, "centroid")
        assert_parity(st[i, 1], ref["spectral_bandwidth"], TOL, "bandwidth")
        # flatness is a LOG-domain statistic: frames with a bin near a spectral null are held to one propagated fp32 FFT
        # floor, every other frame to 1e-5 of the row's peak (tests/gpu_util.py: assert_flatness_parity)
        unsure, nfr = assert_flatness_parity(st[i, 2], ref["spectral_flatness"], S, TOL, "flatness")
        assert unsure <= max(2, 0.05 * nfr)
        rb, db = st[i, 3].astype(int), st[i, 4].astype(int)
        sure = ref["rolloff_margin"] > 1e-6
        assert (rb[sure] == ref["rolloff_bin"][sure]).all() and (np.abs(rb - ref["rolloff_bin"]) <= 1).all()
        sure = ref["dominant_margin"] > 1e-6
        assert (db[sure] == ref["dominant_bin"][sure]).all()
        atol = TOL * S.max()
        for k, (bins, kk) in enumerate(bands):
            srt = np.sort(S[bins], axis=0)
            assert np.abs(pv[i, 1, k] - srt[:kk].mean(axis=0)).max() <= atol, f"valley band {k}"
            assert np.abs(pv[i, 0, k] - srt[-kk:].mean(axis=0)).max() <= atol, f"peak band {k}"
        valley = np.stack([np.sort(S[bins], axis=0)[:kk].mean(axis=0) for bins, kk in bands])
        assert_contrast_parity(cdb[i], O.spectral_contrast(S, sr, freqs=fr), valley, fft_floor(S), TOL, f"contrast dB clip {i}")
    if n_mels is not None:
        mel = mel.cpu().numpy()
        alone = ops.stft_mel_segments(y, sr, n_fft, hop, center, window, None, n_mels, 0.0, None)
        if alone is not None:
            assert np.array_equal(mel, alone.cpu().numpy())             # the same projection on the same rows
        for i in range(B):
            S = np.abs(O.stft(Y[i].astype(np.float64), n_fft, hop, window=window, center=center)) ** 2
            assert_parity(mel[i], O.melspectrogram(S, sr, n_fft, n_mels), TOL, f"mel clip {i}")
    # the rows alone (no filterbank) and single rows: the same bits
    _, st1, _ = ops.stft_rows_seg(y, sr, n_fft, hop, center, window, None, None, 0.0, None, 1 | 8, 0.85, 2.0, None)
    assert np.array_equal(st1.cpu().numpy()[:, [0, 3, 5, 6, 7]], st[:, [0, 3, 5, 6, 7]])
    _, none, pv1 = ops.stft_rows_seg(y, sr, n_fft, hop, center, window, None, None, 0.0, None, 0, 0.85, 2.0, plan)
    assert none is None and np.array_equal(pv1.cpu().numpy(), pv)


@pytest.mark.parametrize("hop,sr,L,center,n_mels", [
    (256, 48000, 30011, True, 40),         # the reference tests' frame / hop (tests/test_features_manager.py:58-62), ragged length
    (512, 22050, 22050, True, None),       # no filterbank: statistics only
    (256, 16000, 512, True, 40),           # :183-220's short signal: L = 512, frame 1024, hop 256 -> 3 frames
    (300, 44100, 9000, False, 26),         # center=False, a hop that is no power of two, an odd frame count
])
def test_rows_frame_length_1024(ops, hop, sr, L, center, n_mels):
    check_rows(ops, 1024, hop, sr, L, center, n_mels, B=5 if L > 512 else 2)


@pytest.mark.parametrize("n_fft,hop,sr,L,center,n_mels", [
    (512, 128, 48000, 30011, True, 40),        # four frames per transform, ragged clip length
    (512, 200, 22050, 9000, False, None),      # center=False, statistics only
    (256, 64, 16000, 20000, True, 26),         # eight frames per transform, 129-bin rows
    (256, 128, 48000, 100, True, None),        # the reference's short-signal case: L = 100 -> 1 frame (tests/test_features_manager.py:183-220)
    (256, 64, 44100, 4000, False, 40),
])
def test_rows_frame_lengths_512_256(ops, n_fft, hop, sr, L, center, n_mels):
    check_rows(ops, n_fft, hop, sr, L, center, n_mels, B=5 if L > 512 else 2)


@pytest.mark.parametrize("hop,sr,L,center,n_mels", [
    (1024, 48000, 30011, True, 40),        # 2049-bin rows (two 16-bin blocks per lane), top contrast band of 1502 bins (radix select)
    (2048, 22050, 22050, True, None),      # statistics only, nothing projected
    (1024, 22050, 20000, True, 128),       # the default filterbank
    (1000, 44100, 9000, False, 64),        # center=False, a hop that is no power of two
    (1024, 16000, 512, True, None),        # a clip shorter than the frame: one frame of mostly padding
])
def test_rows_frame_length_4096(ops, hop, sr, L, center, n_mels):
    check_rows(ops, 4096, hop, sr, L, center, n_mels, B=5 if L > 512 else 2)


def test_manager_frame_4096_takes_the_rows_kernel(ops, monkeypatch):
    """extract_features_batch(frame_length=4096, spectral features + mfcc): rows and mel block from ONE launch."""
    from sygnals_amd.core.features.manager import extract_features_batch
    called = []
    real = ops.stft_any
    monkeypatch.setattr(ops, "stft_any", lambda *a, **k: (called.append(1), real(*a, **k))[1])
    sr = 48000
    Y = O.synth_clips(2, 30000, sr, seed=5)
    feats = ["spectral_centroid", "spectral_bandwidth", "spectral_flatness", "spectral_rolloff", "spectral_contrast", "mfcc"]
    fp = {"mfcc": {"n_mels": 40}}
    d = extract_features_batch(Y, sr, feats, frame_length=4096, hop_length=1024, feature_params=fp)
    assert not called
    for b in range(2):
        ref = O.extract_features(Y[b].astype(np.float64), sr, feats, frame_length=4096, hop_length=1024, feature_params=fp)
        S = np.abs(O.stft(Y[b].astype(np.float64), 4096, 1024))
        st = O.spectral_stats_frames(S, O.fft_frequencies(sr, 4096))
        for k in ref:
            if k == "time":
                assert np.allclose(d[k], ref[k], rtol=0, atol=1e-12)
            elif k == "spectral_rolloff":
                m = st["rolloff_margin"] > 1e-6
                assert np.array_equal(d[k][b][m], ref[k][m]), k
            elif k == "spectral_flatness":
                assert_flatness_parity(d[k][b], ref[k], S, TOL, k)
            elif k.startswith("contrast"):
                assert peak_rel(d[k][b], ref[k]) <= 20 * TOL, k
            else:
                assert_parity(d[k][b], ref[k], TOL, k)


def test_manager_frame_1024_takes_the_rows_kernel(ops, monkeypatch):
    """extract_features_batch(frame_length=1024, spectral features [+ mfcc]) runs ONE fused launch for the rows (plus the
    dB / DCT launch for the MFCC): no complex STFT in HBM.  Columns against the oracle's."""
    from sygnals_amd.core.features.manager import extract_features_batch
    called = []
    real = ops.stft_any
    monkeypatch.setattr(ops, "stft_any", lambda *a, **k: (called.append(1), real(*a, **k))[1])
    sr = 22050
    Y = O.synth_clips(3, 20000, sr, seed=4)
    feats = ["spectral_centroid", "spectral_rolloff", "spectral_contrast", "mfcc"]
    for fp in ({"mfcc": {"n_mels": 40}}, {}):                            # with a piece table / the default 128 bands (dense kernel)
        d = extract_features_batch(Y, sr, feats, frame_length=1024, hop_length=256, feature_params=fp)
        assert not called
        for b in range(3):
            ref = O.extract_features(Y[b].astype(np.float64), sr, feats, frame_length=1024, hop_length=256, feature_params=fp)
            S = np.abs(O.stft(Y[b].astype(np.float64), 1024, 256))
            st = O.spectral_stats_frames(S, O.fft_frequencies(sr, 1024))
            for k in ref:
                if k == "time":
                    assert np.allclose(d[k], ref[k], rtol=0, atol=1e-12)
                elif k == "spectral_rolloff":
                    m = st["rolloff_margin"] > 1e-6
                    assert np.array_equal(d[k][b][m], ref[k][m]), k
                elif k.startswith("contrast"):
                    assert peak_rel(d[k][b], ref[k]) <= 20 * TOL, k        # (dB of a valley: margin-qualified in check_rows)
                else:
                    assert_parity(d[k][b], ref[k], TOL, k)
