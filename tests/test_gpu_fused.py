"""GPU parity of the fused STFT(2048)->mel->MFCC path against the float64 oracle.
Calls go through the C ABI (sygnals_amd.ops -> libsygnals_hip.so)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")
from oracle import cpu_ref as O
from tests.gpu_util import assert_contrast_parity, assert_parity, fft_floor, peak_rel

TOL = 1e-5  # fp32 parity tolerance stated by BASELINE.json north_star


@pytest.fixture(scope="module")
def ops():
    from sygnals_amd import ops
    ops.require_gpu()
    return ops


@pytest.fixture(scope="module")
def clips():
    return O.synth_clips(8, 48000, 48000, seed=0)


def kat_clips(L=48000, sr=48000):
    t = np.arange(L) / sr
    z = np.zeros(L, np.float32)
    imp = z.copy(); imp[L // 2] = 1.0
    dc = np.full(L, 0.5, np.float32)
    sine = (0.9 * np.sin(2 * np.pi * 1000.0 * t)).astype(np.float32)
    return np.stack([z, imp, dc, sine])


def test_stft_complex_matches_oracle(ops, clips):
    y = ops.to_device_f32(clips)
    X = ops.stft2048_c2c(y, hop=512).cpu().numpy()
    X = X[..., 0] + 1j * X[..., 1]                       # [B, T, F]
    for i in range(clips.shape[0]):
        ref = O.stft(clips[i].astype(np.float64), 2048, 512).T
        assert X[i].shape == ref.shape and np.isfinite(X[i]).all()
        assert peak_rel(X[i], ref) <= TOL


def test_mel_power_matches_oracle(ops, clips):
    y = ops.to_device_f32(clips)
    mel, _, _ = ops.stft2048_mel(y, 48000, n_mels=40)
    mel = mel.cpu().numpy()
    for i in range(clips.shape[0]):
        S = np.abs(O.stft(clips[i].astype(np.float64), 2048, 512)) ** 2
        ref = O.melspectrogram(S, 48000, 2048, 40)
        assert_parity(mel[i], ref, TOL, f"mel clip {i}")


@pytest.mark.parametrize("waves", [16, 8])
@pytest.mark.parametrize("n_mels,sr,fmin,fmax", [(1, 48000, 0.0, None), (3, 48000, 0.0, None), (40, 16000, 100.0, 6000.0),
                                                 (128, 22050, 0.0, None), (200, 48000, 0.0, None), (256, 48000, 20.0, 20000.0)])
def test_mel_projection_plans(ops, clips, monkeypatch, waves, n_mels, sr, fmin, fmax):
    """The block-sparse projection (v_mfma_f32_4x4x1_16b_f32, four-row mel groups) over the plan shapes it can meet:
    a single row, a partial group, band limits, 128 / 200 / 256 rows (more than 28 positions per slot: the kernel's
    extra-step loop), both workgroup shapes -- mel power against the float64 product with the same float32 basis."""
    if n_mels > 128 and waves == 8:
        pytest.skip("more than 32 groups of four rows need the 16-wave plan")
    monkeypatch.setattr(ops.settings, "waves", waves)
    y = ops.to_device_f32(clips[:3])
    mel, _, _ = ops.stft2048_mel(y, sr, n_mels=n_mels, fmin=fmin, fmax=fmax)
    mel = mel.cpu().numpy()
    assert mel.shape == (3, n_mels, 94)
    for i in range(3):
        S = np.abs(O.stft(clips[i].astype(np.float64), 2048, 512)) ** 2
        assert_parity(mel[i], O.melspectrogram(S, sr, 2048, n_mels, fmin, fmax), TOL, f"mel n_mels={n_mels} clip {i}")


@pytest.mark.parametrize("n_mels,n_mfcc", [(40, 13), (128, 13), (64, 20), (24, 24)])
def test_mfcc_c2_matches_oracle(ops, clips, n_mels, n_mfcc):
    y = ops.to_device_f32(clips)
    out = ops.mfcc_batch(y, 48000, n_mels=n_mels, n_mfcc=n_mfcc).cpu().numpy()
    ref = O.mfcc_batch(clips, 48000, n_mels=n_mels, n_mfcc=n_mfcc)
    assert out.shape == ref.shape == (8, n_mfcc, 94)
    for i in range(8):
        assert_parity(out[i], ref[i], TOL, f"mfcc clip {i}")


def test_mfcc_degenerate_clips(ops):
    K = kat_clips()
    out = ops.mfcc_batch(ops.to_device_f32(K), 48000, n_mels=40).cpu().numpy()
    ref = O.mfcc_batch(K, 48000, n_mels=40)
    assert_parity(out[0], ref[0], TOL, "all-zero clip")   # amin/amin -> 0 dB everywhere
    assert_parity(out[1], ref[1], TOL, "impulse")
    assert_parity(out[2], ref[2], TOL, "dc")
    assert_parity(out[3], ref[3], TOL, "pure sine")


@pytest.mark.parametrize("L,hop,center", [(100, 512, True), (2048, 512, False), (5000, 160, True),
                                          (4999, 441, True), (48000, 512, False), (3000, 1, False),
                                          (48000, 1024, True), (8001, 700, True), (22050, 256, True)])
def test_ragged_lengths_and_hops(ops, L, hop, center):
    rng = np.random.default_rng(L + hop)
    Y = rng.normal(0, 0.2, (3, L)).astype(np.float32)
    out = ops.mfcc_batch(ops.to_device_f32(Y), 16000, hop=hop, n_mels=40, center=center).cpu().numpy()
    ref = np.stack([O.mfcc_manager(y.astype(np.float64), 16000, 2048, hop, center, "hann", 40, 13) for y in Y])
    assert out.shape == ref.shape
    assert_parity(out, ref, TOL, f"L={L} hop={hop} center={center}")


def test_c1_single_clip_10s_16k(ops):
    y = O.synth_clips(1, 160000, 16000, seed=3)
    out = ops.mfcc_batch(ops.to_device_f32(y), 16000, n_mels=128).cpu().numpy()
    ref = O.mfcc_batch(y, 16000, n_mels=128)
    assert out.shape == (1, 13, 313)
    assert_parity(out, ref, TOL, "C1")


def test_window_and_lifter_variants(ops, clips):
    y = ops.to_device_f32(clips[:2])
    for window in ("hamming", "blackman"):
        mel, _, _ = ops.stft2048_mel(y, 48000, window=window, n_mels=40)
        _, mf = ops.logmel_dct(mel, 13, lifter=22.0)
        ref = np.stack([O.mfcc_manager(c.astype(np.float64), 48000, window=window, n_mels=40, lifter=22.0)
                        for c in clips[:2]])
        assert_parity(mf.cpu().numpy(), ref, TOL, f"window={window} lifter=22")


def test_strided_batch_rows(ops, clips):
    """Row stride (ldy) larger than L and a non-8-byte-aligned view exercise the scalar load path."""
    big = np.zeros((4, 48000 + 7), np.float32)
    big[:, 3:48003] = clips[:4]
    t = ops.to_device_f32(big)[:, 3:48003]
    out = ops.mfcc_batch(t, 48000, n_mels=40).cpu().numpy()
    ref = O.mfcc_batch(clips[:4], 48000, n_mels=40)
    assert_parity(out, ref, TOL, "strided rows")


def test_spectral_stats_match_oracle(ops, clips):
    y = ops.to_device_f32(clips)
    _, st, _ = ops.stft2048_mel(y, 48000, n_mels=40, want_stats=True)
    st = st.cpu().numpy()
    fr = O.fft_frequencies(48000, 2048)
    flips = 0
    for i in range(clips.shape[0]):
        S = np.abs(O.stft(clips[i].astype(np.float64), 2048, 512))
        ref = O.spectral_stats_frames(S, fr)
        assert_parity(st[i, 0], ref["spectral_centroid"], TOL, "centroid")
        assert_parity(st[i, 1], ref["spectral_bandwidth"], TOL, "bandwidth")
        assert_parity(st[i, 2], ref["spectral_flatness"], TOL, "flatness")
        rb = st[i, 3].astype(int); db = st[i, 4].astype(int)
        # bin-valued outputs: identical wherever the float64 decision margin exceeds 1e-6
        sure = ref["rolloff_margin"] > 1e-6
        assert (rb[sure] == ref["rolloff_bin"][sure]).all()
        assert (np.abs(rb - ref["rolloff_bin"]) <= 1).all()
        flips += int((rb != ref["rolloff_bin"]).sum())
        sure = ref["dominant_margin"] > 1e-6
        assert (db[sure] == ref["dominant_bin"][sure]).all()
    print("rolloff bins differing inside the 1e-6 margin:", flips)


def test_spectral_stats_degenerate(ops):
    K = kat_clips()
    _, st, _ = ops.stft2048_mel(ops.to_device_f32(K), 48000, n_mels=40, want_stats=True)
    st = st.cpu().numpy()
    # all-zero frames: centroid 0, bandwidth 0, flatness 0, rolloff = last bin, dominant = bin 0
    assert (st[0, 0] == 0).all() and (st[0, 1] == 0).all() and (st[0, 2] == 0).all()
    assert (st[0, 3] == 1024).all() and (st[0, 4] == 0).all()
    # 1 kHz sine: dominant bin = round(1000/23.4375) = 43 on interior frames
    assert (st[3, 4, 4:-4] == 43).all()


def test_spectral_contrast_matches_oracle(ops, clips):
    from sygnals_amd import _tables as T
    fr = O.fft_frequencies(48000, 2048)
    plan = T.contrast_plan(fr, 48000)
    y = ops.to_device_f32(clips)
    _, _, _pv_dev = ops.stft2048_mel(y, 48000, n_mels=40, contrast=plan)
    pv = _pv_dev.cpu().numpy()
    cdb = ops.contrast_db(_pv_dev).cpu().numpy()
    unsure = cells = 0
    for i in range(clips.shape[0]):
        S = np.abs(O.stft(clips[i].astype(np.float64), 2048, 512))
        bands = O.contrast_bands(fr, 48000)
        # the tail means are raw STFT magnitudes: like the STFT itself they are held to
        # 1e-5 of the spectrogram peak (a valley is orders of magnitude below that peak)
        atol = TOL * S.max()
        for k, (bins, kk) in enumerate(bands):
            srt = np.sort(S[bins], axis=0)
            assert np.abs(pv[i, 1, k] - srt[:kk].mean(axis=0)).max() <= atol, f"valley band {k}"
            assert np.abs(pv[i, 0, k] - srt[-kk:].mean(axis=0)).max() <= atol, f"peak band {k}"
        # the dB contrast: 1e-5 peak-relative on every cell whose float64 valley clears the fp32 FFT floor of its
        # frame by the decision margin; one propagated floor on the rest (gpu_util.assert_contrast_parity)
        valley = np.stack([np.sort(S[bins], axis=0)[:kk].mean(axis=0) for bins, kk in bands])
        u, n = assert_contrast_parity(cdb[i], O.spectral_contrast(S, 48000, freqs=fr), valley, fft_floor(S), TOL,
                                      f"contrast dB clip {i}")
        unsure += u; cells += n
    print(f"contrast cells below the decision margin: {unsure} of {cells}")
    assert unsure <= 0.10 * cells           # (measured: 5 %; a gate that excluded most cells would prove nothing)


@pytest.mark.parametrize("hop,center,L", [(512, True, 48000), (256, True, 20000), (441, False, 30011), (512, True, 2047)])
def test_stats_without_mel_is_bit_identical(ops, hop, center, L):
    """syg_stft2048_stats_f32 (transform + row functions, nothing projected) returns the statistics rows and the contrast
    tail means of syg_stft2048_mel_f32 bit for bit: same transform, same rows, same row functions.  Ragged clip lengths,
    a hop that leaves waves without a frame, an all-zero clip; every mask the callers use."""
    from sygnals_amd import _tables as T
    rng = np.random.default_rng(5)
    Y = (rng.normal(0, 0.2, (37, L)) * rng.random((37, 1))).astype(np.float32)
    Y[3] = 0.0
    y = ops.to_device_f32(Y)
    plan = T.contrast_plan(O.fft_frequencies(48000, 2048), 48000)
    for mask, cp in ((31, None), (1 | 8, plan), (0, plan), (4, None), (1 | 8 | 32, plan)):
        _, st_ref, pv_ref = ops.stft2048_mel(y, 48000, hop, center, n_mels=40, want_stats=mask, contrast=cp)
        st, pv = ops.stft2048_stats(y, 48000, hop, center, want_stats=mask, contrast=cp)
        if mask:
            assert torch.equal(st, st_ref), f"statistics rows differ (mask {mask})"
        else:
            assert st is None
        if cp is not None:
            assert torch.equal(pv, pv_ref), f"contrast tail means differ (mask {mask})"
        else:
            assert pv is None
    with pytest.raises(ValueError):
        ops.stft2048_stats(y, 48000, hop, center, want_stats=0, contrast=None)


def test_stats_without_mel_bad_arguments(ops):
    """The C entry refuses what it cannot run (hop > 512: not staged) with an error code, not a launch."""
    from sygnals_amd._lib import SygnalsHipError
    y = ops.to_device_f32(np.zeros((2, 48000), np.float32))
    with pytest.raises(SygnalsHipError):
        ops.stft2048_stats(y, 48000, 1024, True, want_stats=1)


@pytest.mark.parametrize("sr", [16000, 22050, 24000, 32000, 44100])
def test_spectral_contrast_tail_means_other_sample_rates(ops, sr):
    """The band plan changes with the sample rate (top band of 206 / 431 / 479 / 616 / 728 bins, k = 4 ... 15): every
    selection form of band_contrast against sorting, through the statistics-only launch and the mel launch."""
    from sygnals_amd import _tables as T
    rng = np.random.default_rng(sr)
    L = 20000
    t = np.arange(L) / sr
    K = np.stack([rng.normal(0, 0.3, L), np.sin(2 * np.pi * 0.31 * sr * t) + rng.normal(0, 1e-3, L), np.zeros(L),
                  rng.normal(0, 1.0, L) * np.linspace(0, 1, L)]).astype(np.float32)
    fr = O.fft_frequencies(sr, 2048)
    plan = T.contrast_plan(fr, sr)
    y = ops.to_device_f32(K)
    _, pv_dev = ops.stft2048_stats(y, sr, want_stats=0, contrast=plan)
    _, _, pv_mel = ops.stft2048_mel(y, sr, n_mels=40, contrast=plan)
    assert torch.equal(pv_dev, pv_mel)
    pv = pv_dev.cpu().numpy()
    for i in range(K.shape[0]):
        S = np.abs(O.stft(K[i].astype(np.float64), 2048, 512))
        atol = TOL * max(S.max(), 1e-30)
        for k, (bins, kk) in enumerate(O.contrast_bands(fr, sr)):
            srt = np.sort(S[bins], axis=0)
            assert np.abs(pv[i, 1, k] - srt[:kk].mean(axis=0)).max() <= atol, f"sr {sr} clip {i} valley band {k}"
            assert np.abs(pv[i, 0, k] - srt[-kk:].mean(axis=0)).max() <= atol, f"sr {sr} clip {i} peak band {k}"


def test_spectral_contrast_tail_selection_cases(ops):
    """The wide band's tails are taken by selection (k-th largest of the lanes' two top values, count, sum, take back the
    extras) with a fall-back to k extraction rounds when too many values tie at the threshold or a lane holds more than
    three candidates: clips that drive both ways -- noise (few extras), an impulse (flat spectrum: every bin ties),
    silence, a tone over a noise floor, a step, and twelve tones 64 bins apart (bin 278 + 64 r: every peak of the wide
    band in the registers of the SAME three lanes)."""
    from sygnals_amd import _tables as T
    rng = np.random.default_rng(77)
    L = 48000
    t = np.arange(L) / 48000.0
    imp = np.zeros(L); imp[5 * 512 + 1024] = 1.0
    step = np.zeros(L); step[L // 2:] = 0.5
    comb = sum((1.0 + 0.05 * r) * np.sin(2 * np.pi * (278 + 64 * r) * 48000.0 / 2048.0 * t + r) for r in range(12)) * 0.05
    K = np.stack([rng.normal(0, 0.3, L), imp, np.zeros(L), np.sin(2 * np.pi * 9000.0 * t) + rng.normal(0, 1e-4, L), step,
                  rng.normal(0, 1.0, L) * np.linspace(0, 1, L), comb + rng.normal(0, 1e-3, L)]).astype(np.float32)
    fr = O.fft_frequencies(48000, 2048)
    plan = T.contrast_plan(fr, 48000)
    _, _, pv_dev = ops.stft2048_mel(ops.to_device_f32(K), 48000, n_mels=40, contrast=plan)
    pv = pv_dev.cpu().numpy()
    for i in range(K.shape[0]):
        S = np.abs(O.stft(K[i].astype(np.float64), 2048, 512))
        atol = TOL * max(S.max(), 1e-30)
        for k, (bins, kk) in enumerate(O.contrast_bands(fr, 48000)):
            srt = np.sort(S[bins], axis=0)
            assert np.abs(pv[i, 1, k] - srt[:kk].mean(axis=0)).max() <= atol, f"clip {i} valley band {k}"
            assert np.abs(pv[i, 0, k] - srt[-kk:].mean(axis=0)).max() <= atol, f"clip {i} peak band {k}"


def test_large_batch_consistency(ops):
    """Full C2 batch size: every clip of a 1024-clip batch equals the same clip run alone
    (size-independent property; the oracle is only run on a sample)."""
    Y = O.synth_clips(16, 48000, 48000, seed=11)
    big = np.tile(Y, (64, 1))
    out = ops.mfcc_batch(ops.to_device_f32(big), 48000, n_mels=40).cpu().numpy()
    assert out.shape == (1024, 13, 94)
    # (the 1024-clip batch takes the one-launch form by itself; run the small batch through the same form)
    small = ops.mfcc_batch(ops.to_device_f32(Y), 48000, n_mels=40, fused=True).cpu().numpy()
    assert np.array_equal(out.reshape(64, 16, 13, 94), np.broadcast_to(small, (64, 16, 13, 94)))
    two = ops.mfcc_batch(ops.to_device_f32(big), 48000, n_mels=40, fused=False).cpu().numpy()
    assert np.array_equal(two.reshape(64, 16, 13, 94), np.broadcast_to(two[:16], (64, 16, 13, 94)))
    assert_parity(out, two, TOL, "one-launch vs two-launch form at the full batch")
    ref = O.mfcc_batch(Y[:4], 48000, n_mels=40)
    assert_parity(small[:4], ref, TOL, "sample of the large batch")


@pytest.mark.parametrize("n_mels,n_mfcc,lifter", [(40, 13, 0.0), (24, 24, 22.0), (44, 16, 0.0)])
def test_mfcc_one_launch_matches_oracle(ops, clips, n_mels, n_mfcc, lifter):
    """syg_stft2048_mfcc_f32 (clip-resident mel matrix) against the oracle and against the two-launch form."""
    y = ops.to_device_f32(clips)
    mf, mel = ops.stft2048_mfcc(y, 48000, n_mels=n_mels, n_mfcc=n_mfcc, lifter=lifter, keep_mel=True)
    ref = np.stack([O.mfcc_manager(c.astype(np.float64), 48000, n_mels=n_mels, n_mfcc=n_mfcc, lifter=lifter)
                    for c in clips])
    assert mf.shape == ref.shape
    assert_parity(mf.cpu().numpy(), ref, TOL, f"one-launch MFCC n_mels={n_mels}")
    two = ops.mfcc_batch(y, 48000, n_mels=n_mels, n_mfcc=n_mfcc, lifter=lifter, fused=False)
    assert_parity(mf.cpu().numpy(), two.cpu().numpy(), TOL, "one-launch vs two-launch")
    mel2, _, _ = ops.stft2048_mel(y, 48000, n_mels=n_mels)
    assert torch.equal(mel, mel2), "the stored mel power must be the two-launch kernel's, bit for bit"


def test_mfcc_one_launch_ragged_and_degenerate(ops):
    """Clips whose frame count is not a multiple of 16, more clips than workgroups, all-zero clip, fixed ref."""
    rng = np.random.default_rng(11)
    Y = rng.normal(0, 0.1, (300, 5000)).astype(np.float32)
    Y[7] = 0.0
    y = ops.to_device_f32(Y)
    mf, _ = ops.stft2048_mfcc(y, 16000, hop=160, n_mels=40)
    two = ops.mfcc_batch(y, 16000, hop=160, n_mels=40, fused=False)
    assert mf.shape == two.shape == (300, 13, 32)
    assert_parity(mf.cpu().numpy(), two.cpu().numpy(), TOL, "ragged one-launch vs two-launch")
    ref = np.stack([O.mfcc_manager(c.astype(np.float64), 16000, 2048, 160, True, "hann", 40, 13) for c in Y[:9]])
    assert_parity(mf[:9].cpu().numpy(), ref, TOL, "ragged one-launch vs oracle")
    mf1, _ = ops.stft2048_mfcc(y[:5], 16000, hop=160, n_mels=40, ref=1.0, top_db=None)
    mel, _, _ = ops.stft2048_mel(y[:5], 16000, hop=160, n_mels=40)
    _, two1 = ops.logmel_dct(mel, 13, ref=1.0, top_db=None)
    assert_parity(mf1.cpu().numpy(), two1.cpu().numpy(), TOL, "fixed reference, no clamp")


@pytest.mark.parametrize("sr,hop,n_mels,n_mfcc,L,B", [(48000, 512, 40, 13, 48000, 8), (16000, 160, 40, 13, 5000, 300),
                                                       (22050, 512, 40, 20, 22050, 37), (48000, 256, 32, 13, 9000, 50),
                                                       (48000, 512, 40, 13, 3000, 19), (8000, 128, 26, 13, 4001, 21)])
def test_mfcc_segment_projection_matches_matrix_form_and_oracle(ops, sr, hop, n_mels, n_mfcc, L, B):
    """syg_stft2048_mfcc_tri_f32 (each wave projects its own row by segment sums, MODE 6) against the matrix form and
    the oracle: whole tiles and ragged ones, one-tile clips (the DCT of a clip runs beside the next clip's only tile),
    more clips than workgroups, an all-zero clip, lengths that rule out the 16-byte staged loads."""
    Y = O.synth_clips(B, L, sr, seed=5)
    Y[B // 2] = 0.0
    y = ops.to_device_f32(Y)
    seg, _ = ops.stft2048_mfcc(y, sr, hop=hop, n_mels=n_mels, n_mfcc=n_mfcc, projection="segments")
    mat, _ = ops.stft2048_mfcc(y, sr, hop=hop, n_mels=n_mels, n_mfcc=n_mfcc, projection="matrix")
    assert seg.shape == mat.shape
    assert_parity(seg.cpu().numpy(), mat.cpu().numpy(), TOL, "segment sums vs matrix form")
    idx = sorted(set([0, 1, B // 2, B - 1]))
    ref = np.stack([O.mfcc_manager(Y[i].astype(np.float64), sr, 2048, hop, True, "hann", n_mels, n_mfcc) for i in idx])
    assert_parity(seg[idx].cpu().numpy(), ref, TOL, "segment sums vs oracle")
    again, _ = ops.stft2048_mfcc(y, sr, hop=hop, n_mels=n_mels, n_mfcc=n_mfcc, projection="segments")
    assert torch.equal(seg, again), "fixed summation order: repeated launches give the same bits"


@pytest.mark.parametrize("L,hop,center", [(7001, 250, False), (6143, 511, True), (2048, 512, False), (1500, 64, True)])
def test_mfcc_segment_projection_odd_shapes(ops, L, hop, center):
    """Hops / lengths that rule out the 16-byte staged loads, center=False, a single frame, a clip shorter than a frame."""
    Y = O.synth_clips(9, L, 16000, seed=8)
    y = ops.to_device_f32(Y)
    seg, _ = ops.stft2048_mfcc(y, 16000, hop=hop, center=center, n_mels=40, projection="segments")
    ref = np.stack([O.mfcc_manager(c.astype(np.float64), 16000, 2048, hop, center, "hann", 40, 13) for c in Y])
    assert seg.shape == ref.shape
    assert_parity(seg.cpu().numpy(), ref, TOL, f"segment sums L={L} hop={hop} center={center}")


def test_mfcc_segment_projection_variants(ops, clips):
    """Fixed reference / no clamp / lifter / other windows through the segment form; shapes without a piece table or
    with keep_mel fall back to the matrix form ("auto") or are refused ("segments")."""
    y = ops.to_device_f32(clips)
    a, _ = ops.stft2048_mfcc(y, 48000, n_mels=40, ref=1.0, top_db=None, lifter=22.0, window="hamming", projection="segments")
    b, _ = ops.stft2048_mfcc(y, 48000, n_mels=40, ref=1.0, top_db=None, lifter=22.0, window="hamming", projection="matrix")
    assert_parity(a.cpu().numpy(), b.cpu().numpy(), TOL, "fixed reference, lifter, hamming")
    with pytest.raises(Exception, match="no segment-sum projection"):
        ops.stft2048_mfcc(y, 48000, n_mels=128, projection="segments")
    with pytest.raises(Exception, match="no segment-sum projection"):
        ops.stft2048_mfcc(y, 48000, n_mels=40, keep_mel=True, projection="segments")
    mf, mel = ops.stft2048_mfcc(y, 48000, n_mels=40, keep_mel=True)          # auto: the matrix form stores the copy
    assert mel is not None and mf.shape == (8, 13, 94)


def test_mfcc_one_launch_default_filterbank_in_the_stage_buffers_place(ops):
    """The reference's default 128 bands (manager.py:214): a 128 x 96 clip matrix does not fit beside the tile stage buffer
    but does in its place -- the launch then loads its frames straight from global memory.  Against the oracle and the
    two-launch form; 64 bands fit the same way (the default call keeps two launches there: measured faster)."""
    lib = ops.lib()
    assert lib.syg_stft2048_mfcc_fits(128, 94, 13) == 1 and lib.syg_stft2048_mfcc_fits(64, 94, 13) == 1
    assert lib.syg_stft2048_mfcc_fits(40, 94, 13) == 2 and lib.syg_stft2048_mfcc_fits(128, 120, 13) == 0
    assert ops.mfcc_fused_pays(128, 94) and not ops.mfcc_fused_pays(64, 94) and ops.mfcc_fused_pays(40, 94)
    Y = O.synth_clips(130, 48000, 48000, seed=12)
    y = ops.to_device_f32(Y)
    for nm in (128, 64):
        one = ops.mfcc_batch(y, 48000, n_mels=nm, fused=True).cpu().numpy()
        two = ops.mfcc_batch(y, 48000, n_mels=nm, fused=False).cpu().numpy()
        assert np.abs(one - two).max() <= 2e-6 * np.abs(two).max()
        idx = [0, 77, 129]
        ref = O.mfcc_batch(Y[idx], 48000, n_mels=nm, n_mfcc=13)
        for j, i in enumerate(idx):
            assert_parity(one[i], ref[j], TOL, f"{nm} bands clip {i}")
    assert np.array_equal(ops.mfcc_batch(y, 48000, n_mels=128).cpu().numpy(), ops.mfcc_batch(y, 48000, n_mels=128, fused=True).cpu().numpy())


def test_mfcc_one_launch_rejects_oversized_clip(ops):
    y = ops.to_device_f32(np.zeros((2, 160000), np.float32))
    assert not ops.mfcc_fused_fits(128, 313)
    with pytest.raises(Exception, match="does not fit"):
        ops.stft2048_mfcc(y, 16000, n_mels=128)
    out = ops.mfcc_batch(y, 16000, n_mels=128)          # falls back to the two-launch form
    assert out.shape == (2, 13, 313)


@pytest.mark.parametrize("n_fft,hop,n_mels", [(1024, 256, 40), (512, 128, 64), (1000, 250, 40), (4096, 1024, 128), (2048, 512, 300)])
def test_mfcc_batch_other_frame_lengths(n_fft, hop, n_mels):
    """mfcc_batch away from the fused shape (n_fft != 2048, or more mel bands than the fused kernel holds): complex
    STFT of any frame length -> |X|^2 -> dense mel -> dB + DCT, same numbers as the oracle's chain."""
    from sygnals_amd import ops
    Y = O.synth_clips(3, 24000, 16000, seed=31)
    out = ops.mfcc_batch(ops.to_device_f32(Y), 16000, n_fft=n_fft, hop=hop, n_mels=n_mels).cpu().numpy()
    ref = O.mfcc_batch(Y, 16000, n_fft=n_fft, hop_length=hop, n_mels=n_mels)
    assert out.shape == ref.shape
    for b in range(3):
        assert_parity(out[b], ref[b], 1e-5, f"n_fft={n_fft} clip {b}")


def test_reserved_cus_change_shares_not_results(monkeypatch):
    """syg_set_option(SYG_OPT_RESERVED_CUS) leaves compute units out of the persistent grids (room for a collective's workgroups):
    other tile / clip shares per workgroup, the same bits."""
    from sygnals_amd import ops
    Y = ops.to_device_f32(O.synth_clips(300, 30000, 48000, seed=3))
    ref1 = ops.mfcc_batch(Y, 48000, n_mels=40, fused=True)
    ref2 = ops.mfcc_batch(Y, 48000, n_mels=40, fused=False)
    for r in (32, 200, 255):
        with ops.override(reserved_cus=r):
            assert ops.get_option("reserved_cus") == r
            assert torch.equal(ops.mfcc_batch(Y, 48000, n_mels=40, fused=True), ref1), r
            assert torch.equal(ops.mfcc_batch(Y, 48000, n_mels=40, fused=False), ref2), r
    assert ops.get_option("reserved_cus") == 0
    for bad in (-1, 256, 1000):                                  # out of range: refused, the option keeps its value
        with pytest.raises(ops.SygnalsHipError):
            ops.set_reserved_cus(bad)
    assert ops.get_option("reserved_cus") == 0


@pytest.mark.parametrize("form", ["segments"])
def test_c4_features_one_launch_matches_two_launch_and_oracle(form):
    """MODE 7 (syg_stft2048_features_tri_f32): MFCC + centroid + rolloff + contrast
    from ONE fused launch -- the block it builds (with the small rows kernel behind it) equals the mel -> feature_block
    form and the oracle's columns."""
    from sygnals_amd import ops
    from sygnals_amd.core.features.manager import feature_block
    sr = 48000
    Y = O.synth_clips(20, 48000, sr, seed=77)
    Y[5] *= 1e-3
    Y[7][:] = 0.0
    y = ops.to_device_f32(Y)
    one = feature_block(y, sr, one_launch=form).cpu().numpy()
    two = feature_block(y, sr, one_launch=False).cpu().numpy()
    assert one.shape == two.shape == (20, 22, 94)
    # the statistics / contrast rows come from the same row functions: identical; the MFCC rows differ by the dB form
    # (hardware log2 vs log10f) within the parity gate
    assert np.array_equal(one[:, 13:], two[:, 13:])
    for b in range(20):
        assert peak_rel(one[b, :13], two[b, :13]) <= 2e-6 or np.abs(two[b, :13]).max() == 0
    feats = ["mfcc", "spectral_centroid", "spectral_rolloff"]
    for b in (0, 5, 11):
        ref = O.extract_features(Y[b].astype(np.float64), sr, feats, feature_params={"mfcc": {"n_mels": 40}})
        want = np.stack([ref[f"mfcc_{i}"] for i in range(13)])
        assert_parity(one[b, :13], want, TOL, f"one-launch mfcc clip {b}")
        assert_parity(one[b, 13], ref["spectral_centroid"], TOL, f"one-launch centroid clip {b}")


def test_mfcc_projection_forms_agree_on_random_shapes(ops):
    """Thirty seeded random calls of the one-launch MFCC: whatever form "auto" picks (segment sums where the filterbank,
    the hop and the LDS allow it) agrees with the matrix form at the parity gate; degenerate clips included."""
    rng = np.random.default_rng(77)
    seg = 0
    for _ in range(30):
        sr = int(rng.choice([8000, 16000, 22050, 32000, 44100, 48000]))
        n_mels = int(rng.choice([20, 26, 32, 40, 44]))
        hop = int(rng.choice([128, 160, 256, 400, 512, 441]))
        center = bool(rng.random() < 0.8)
        B = int(rng.integers(1, 40))
        tmax = 96 if n_mels <= 40 else 80
        L = int(rng.integers(2048 if not center else 1, hop * (tmax - 2)))
        n_mfcc = int(rng.integers(1, min(n_mels, 20) + 1))
        Y = (rng.normal(0, 0.3, (B, L)) * rng.random((B, 1)) ** 3).astype(np.float32)
        if B > 2:
            Y[1] = 0.0
        y = ops.to_device_f32(Y)
        Tn = ops.num_frames(L, 2048, hop, center)
        if not ops.mfcc_fused_fits(n_mels, Tn, n_mfcc):
            continue
        kw = dict(hop=hop, center=center, n_mels=n_mels, n_mfcc=n_mfcc, lifter=float(rng.choice([0.0, 22.0])),
                  top_db=(None if rng.random() < 0.2 else 80.0), ref=("max" if rng.random() < 0.7 else 1.0))
        a, _ = ops.stft2048_mfcc(y, sr, projection="auto", **kw)
        m, _ = ops.stft2048_mfcc(y, sr, projection="matrix", **kw)
        assert a.shape == m.shape == (B, n_mfcc, Tn)
        assert_parity(a.cpu().numpy(), m.cpu().numpy(), TOL, f"sr={sr} n_mels={n_mels} hop={hop} L={L} B={B} {kw}")
        key = [k for k in ops._mfcc_calls if k[4] == float(sr) and k[8] == n_mels and k[-1] == "auto" and k[2] == L and k[5] == hop]
        seg += any(ops._mfcc_calls[k].tri for k in key)
    assert seg >= 8


# ---------------------------------------------------------------- MODE 8 / 9: the tile form with a four-pass piece table
@pytest.mark.parametrize("sr,n_mels,hop,center,L,fmin,fmax", [
    (48000, 128, 512, True, 48000, 0.0, None),        # the reference's default filterbank (manager.py:214) on the C2 clip
    (16000, 128, 512, True, 160000, 0.0, None),       # BASELINE config C1: 10 s @ 16 kHz (nearly empty low filters)
    (44100, 64, 441, True, 30011, 0.0, None),         # odd hop: unaligned staged runs
    (48000, 96, 1024, True, 50000, 100.0, 12000.0),   # hop > 512: direct frame loads
    (22050, 128, 256, False, 9000, 0.0, None),        # center=False, four frames per 1024 samples
    (48000, 40, 512, True, 2047, 0.0, None),          # a clip shorter than a frame (all padding but the middle)
])
def test_mel_tri_four_pass_matches_oracle_and_matrix_form(ops, sr, n_mels, hop, center, L, fmin, fmax):
    """syg_stft2048_mel_tri_f32 (per-wave segment sums, four passes, mel columns written by the transforming wave) against
    the float64 oracle at 1e-5 and against the matrix form of the projection (same float32 weights to 2e-7)."""
    Y = O.synth_clips(5, L, sr, seed=L % 97)
    Y[3] *= 1e-3
    Y[4][:] = 0.0
    y = ops.to_device_f32(Y)
    a, _, _ = ops.stft2048_mel(y, sr, hop, center, "hann", 2048, n_mels, fmin, fmax, projection="segments")
    b, _, _ = ops.stft2048_mel(y, sr, hop, center, "hann", 2048, n_mels, fmin, fmax, projection="matrix")
    a, b = a.cpu().numpy(), b.cpu().numpy()
    assert a.shape == b.shape and np.isfinite(a).all()
    for i in range(5):
        S = np.abs(O.stft(Y[i].astype(np.float64), 2048, hop, center=center)) ** 2
        ref = O.melspectrogram(S, sr, 2048, n_mels, fmin, fmax if fmax is not None else sr / 2.0)
        assert a[i].shape == ref.shape
        assert_parity(a[i], ref, TOL, f"segments clip {i}")
        assert_parity(b[i], ref, TOL, f"matrix clip {i}")
    assert (a[4] == 0).all()


def test_default_mfcc_128_bands_takes_the_segment_kernel(ops, clips):
    """mfcc_batch with the reference's default n_mels = 128: the four-pass segment kernel + the dB / DCT launch, 1e-5 against
    the oracle; C1's shape (one 10 s clip @ 16 kHz) likewise."""
    cfg = ops.mel_config(48000, 2048, 128)
    assert cfg.segtab is None and cfg.segtab4 is not None
    out = ops.mfcc_batch(ops.to_device_f32(clips), 48000).cpu().numpy()          # (n_mels defaults to 128)
    ref = O.mfcc_batch(clips, 48000, n_mels=128, n_mfcc=13)
    assert peak_rel(out, ref) <= TOL
    Y1 = O.synth_clips(1, 160000, 16000, seed=11)
    out1 = ops.mfcc_batch(ops.to_device_f32(Y1), 16000).cpu().numpy()
    assert out1.shape == (1, 13, 313)
    assert peak_rel(out1, O.mfcc_batch(Y1, 16000, n_mels=128, n_mfcc=13)) <= TOL


def test_mel_tri_rows_are_the_rows_of_the_statistics_kernel(ops):
    """MODE 9 = MODE 8 + the row functions: statistics / contrast rows bit for bit those of syg_stft2048_stats_f32 (the same
    functions on the same power rows), the mel block bit for bit MODE 8's."""
    from sygnals_amd import _tables as T
    sr = 48000
    Y = O.synth_clips(6, 40000, sr, seed=5)
    y = ops.to_device_f32(Y)
    cplan = T.contrast_plan(np.fft.rfftfreq(2048, 1.0 / sr), sr)
    mel, st, cpv = ops.stft2048_mel(y, sr, n_mels=128, want_stats=31, contrast=cplan, projection="segments")
    st2, cpv2 = ops.stft2048_stats(y, sr, want_stats=31, contrast=cplan)
    assert torch.equal(st, st2) and torch.equal(cpv, cpv2)
    mel0, _, _ = ops.stft2048_mel(y, sr, n_mels=128, projection="segments")
    # MODE 8 keeps 4 |X|^2 in the rows and scales the bands back (exact); MODE 9 keeps |X|^2: same sums up to that scaling
    assert peak_rel(mel.cpu().numpy(), mel0.cpu().numpy()) <= 1e-6
