"""GPU parity of the generic kernels (pow2 FFT, generic STFT, dense mel, stand-alone spectral
statistics / contrast, sosfiltfilt, Welch) against the oracle and against golden vectors that
were produced by the reference's own functions (tests/golden/ref_*.npz)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")
from oracle import cpu_ref as O
from tests.gpu_util import assert_parity, peak_rel

TOL = 1e-5
G = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def ops():
    from sygnals_amd import ops
    ops.require_gpu()
    return ops


def c2n(t):
    a = t.cpu().numpy()
    return a[..., 0] + 1j * a[..., 1]


@pytest.mark.parametrize("n", [2, 4, 8, 16, 32, 64, 128, 256, 512, 1024, 2048, 4096, 8192])
def test_fft_pow2_forward_inverse(ops, n):
    rng = np.random.default_rng(n)
    x = (rng.normal(size=(5, n)) + 1j * rng.normal(size=(5, n))).astype(np.complex64)
    xd = torch.from_numpy(np.stack([x.real, x.imag], -1).copy()).cuda()
    X = c2n(ops.fft_pow2(xd))
    ref = np.fft.fft(x.astype(np.complex128), axis=1)
    assert peak_rel(X, ref) <= TOL
    xi = c2n(ops.fft_pow2(xd, inverse=True))
    assert peak_rel(xi, np.fft.ifft(x.astype(np.complex128), axis=1)) <= TOL
    back = c2n(ops.fft_pow2(ops.fft_pow2(xd), inverse=True))
    assert peak_rel(back, x) <= TOL


@pytest.mark.parametrize("n_fft,hop,win_length,center", [(64, 16, None, True), (256, 64, None, True),
                                                         (512, 128, 400, True), (1024, 256, None, True),
                                                         (1024, 256, None, False), (4096, 1024, None, True),
                                                         (8192, 4096, None, True), (2048, 512, None, True)])
def test_stft_pow2_matches_oracle(ops, n_fft, hop, win_length, center):
    Y = O.synth_clips(3, 20000, 16000, seed=n_fft)
    if n_fft == 2048:
        X = c2n(ops.stft_pow2(ops.to_device_f32(Y), n_fft, hop, center, "hann", win_length))
        X2 = c2n(ops.stft2048_c2c(ops.to_device_f32(Y), hop, center))
        assert peak_rel(X2, X) <= 2e-6          # the two device kernels agree
    else:
        X = c2n(ops.stft_any(ops.to_device_f32(Y), n_fft, hop, center, "hann", win_length))
    for i in range(3):
        ref = O.stft(Y[i].astype(np.float64), n_fft, hop, win_length, "hann", center).T
        assert X[i].shape == ref.shape
        assert peak_rel(X[i], ref) <= TOL


def test_reference_shape_cases(ops):
    """Frame counts pinned by reference tests/test_features_manager.py:183-220."""
    assert ops.stft_any(ops.to_device_f32(np.ones((1, 512), np.float32)), 1024, 256).shape[1] == 3
    assert ops.stft_any(ops.to_device_f32(np.ones((1, 100), np.float32)), 2048, 512).shape[1] == 1


def test_mel_dense_and_cabs(ops):
    from sygnals_amd import _tables as T
    Y = O.synth_clips(2, 16000, 16000, seed=9)
    X = ops.stft_any(ops.to_device_f32(Y), 1024, 256)
    P = ops.cabs_pow(X, 2)
    Mg = ops.cabs_pow(X, 1)
    basis = T.mel_filterbank(16000, 1024, 64)
    mel = ops.mel_dense(P, torch.from_numpy(basis).cuda()).cpu().numpy()
    for i in range(2):
        S = np.abs(O.stft(Y[i].astype(np.float64), 1024, 256))
        assert_parity(Mg[i].cpu().numpy().T, S, TOL, "magnitude")
        assert_parity(mel[i], O.melspectrogram(S ** 2, 16000, 1024, 64), TOL, "dense mel")


def test_spectral_stats_vs_reference_golden(ops):
    """Stand-alone statistics kernel against the REFERENCE's own per-frame outputs."""
    g = np.load(os.path.join(G, "ref_freq.npz"))
    S, fr = g["spectra"], g["freqs"]
    st = ops.spectral_stats(torch.from_numpy(S.astype(np.float32)).cuda(),
                            torch.from_numpy(fr.astype(np.float32)).cuda()).cpu().numpy()
    assert_parity(st[0], g["centroid"], TOL, "centroid")
    assert_parity(st[1], g["bandwidth"], TOL, "bandwidth")
    assert_parity(st[2], g["flatness"], TOL, "flatness")
    ref = O.spectral_stats_frames(S.astype(np.float32).astype(np.float64).T, fr)
    sure = ref["rolloff_margin"] > 1e-6
    assert (fr[st[3].astype(int)][sure] == g["rolloff85"][sure]).all()
    assert (np.abs(st[3].astype(int) - ref["rolloff_bin"]) <= 1).all()
    sure = ref["dominant_margin"] > 1e-6
    assert (fr[st[4].astype(int)][sure] == g["dominant"][sure]).all()
    for rp, key in ((0.5, "rolloff50"), (0.0, "rolloff0")):
        s2 = ops.spectral_stats(torch.from_numpy(S.astype(np.float32)).cuda(),
                                torch.from_numpy(fr.astype(np.float32)).cuda(), roll_percent=rp).cpu().numpy()
        r2 = O.spectral_stats_frames(S.astype(np.float32).astype(np.float64).T, fr, roll_percent=rp)
        sure = r2["rolloff_margin"] > 1e-6
        assert (fr[s2[3].astype(int)][sure] == g[key][sure]).all()
    s3 = ops.spectral_stats(torch.from_numpy(S.astype(np.float32)).cuda(),
                            torch.from_numpy(fr.astype(np.float32)).cuda(), bw_p=1.0).cpu().numpy()
    assert_parity(s3[1], g["bandwidth_p1"], TOL, "bandwidth p=1")


def test_contrast_standalone(ops):
    from sygnals_amd import _tables as T
    rng = np.random.default_rng(4)
    S = np.abs(rng.normal(size=(37, 513))).astype(np.float32)
    fr = O.fft_frequencies(22050, 1024)
    plan = T.contrast_plan(fr, 22050)
    pv = ops.contrast_pv(torch.from_numpy(S).cuda(), plan).cpu().numpy()
    for k, (bins, kk) in enumerate(O.contrast_bands(fr, 22050)):
        srt = np.sort(S.astype(np.float64)[:, bins], axis=1)
        assert_parity(pv[1, k], srt[:, :kk].mean(axis=1), TOL, f"valley {k}")
        assert_parity(pv[0, k], srt[:, -kk:].mean(axis=1), TOL, f"peak {k}")
    # ties and constant rows
    S2 = np.ones((3, 513), np.float32); S2[1, 100:200] = 5.0; S2[2] = 0.0
    pv = ops.contrast_pv(torch.from_numpy(S2).cuda(), plan).cpu().numpy()
    for k, (bins, kk) in enumerate(O.contrast_bands(fr, 22050)):
        srt = np.sort(S2.astype(np.float64)[:, bins], axis=1)
        assert_parity(pv[1, k], srt[:, :kk].mean(axis=1), TOL, "valley ties")
        assert_parity(pv[0, k], srt[:, -kk:].mean(axis=1), TOL, "peak ties")


DESIGNS = ["bp4_48k", "lp5_1k", "hp5_1k", "bs5_1k", "lp8_1k", "bp2_16k"]


@pytest.mark.parametrize("name", DESIGNS)
def test_sosfiltfilt_vs_reference_golden(ops, name):
    """Device zero-phase filtering against outputs of the REFERENCE's apply_sos_filter."""
    g = np.load(os.path.join(G, "ref_filters.npz"))
    sos, x, yref = g[f"{name}_sos"], g[f"{name}_x"], g[f"{name}_y"]
    xb = np.stack([x, -0.5 * x, x[::-1]]).astype(np.float32)
    y = ops.sosfiltfilt(ops.to_device_f32(xb), sos, O.sosfilt_zi(sos), O.sosfiltfilt_padlen(sos)).cpu().numpy()
    assert_parity(y[0], yref, TOL, name)
    assert_parity(y[1], -0.5 * yref, TOL, name + " (linearity)")
    assert_parity(y[2], O.apply_sos_filter(sos, xb[2].astype(np.float64)), TOL, name + " (reversed)")


def test_sosfiltfilt_c3_clips(ops):
    sos = O.design_butterworth_sos((300.0, 3400.0), 48000.0, 4, "bandpass")
    Y = O.synth_clips(6, 48000, 48000, seed=21)
    y = ops.sosfiltfilt(ops.to_device_f32(Y), sos, O.sosfilt_zi(sos), 27).cpu().numpy()
    for i in range(6):
        assert_parity(y[i], O.apply_sos_filter(sos, Y[i].astype(np.float64)), TOL, f"C3 clip {i}")
    # ragged: shortest legal input (L = padlen + 1) and a non-multiple of the chunk size
    for L in (28, 257, 1000):
        x = Y[:2, :L].copy()
        y = ops.sosfiltfilt(ops.to_device_f32(x), sos, O.sosfilt_zi(sos), 27).cpu().numpy()
        assert_parity(y, np.stack([O.apply_sos_filter(sos, r.astype(np.float64)) for r in x]), TOL, f"L={L}")
    with pytest.raises(ValueError, match="greater than padlen, which is 27"):
        ops.sosfiltfilt(ops.to_device_f32(Y[:1, :27]), sos, O.sosfilt_zi(sos), 27)


@pytest.mark.parametrize("L", [28, 100, 16384 - 54, 16384 - 53, 30000, 32768 - 54, 40001, 48000, 49152 - 54, 49152 - 53, 65000,
                               65536 - 54, 65536 - 53])
def test_sosfiltfilt_clip_resident_lengths(ops, L, monkeypatch):
    """Clips that fit a workgroup's registers take both sweeps in one launch (sosfilt_clip.hip: 64 / 128 / 192 / 256
    samples per lane): every chunk length and the lengths on either side of each boundary, against the oracle, and
    against the chunked path (same float64 recurrences, other chunking: agreement far inside the tolerance)."""
    sos = O.design_butterworth_sos((300.0, 3400.0), 48000.0, 4, "bandpass")
    rng = np.random.default_rng(L)
    x = np.stack([rng.normal(0, 0.3, L) + 0.7, np.sin(np.arange(L) * 0.05) * (1 + rng.normal(0, 0.01, L)),
                  rng.normal(0, 1.0, L)]).astype(np.float32)
    y = ops.sosfiltfilt(ops.to_device_f32(x), sos, O.sosfilt_zi(sos), 27).cpu().numpy()
    with ops.override(sos_clip=0):
        yc = ops.sosfiltfilt(ops.to_device_f32(x), sos, O.sosfilt_zi(sos), 27).cpu().numpy()
    for i in range(3):
        ref = O.apply_sos_filter(sos, x[i].astype(np.float64))
        assert_parity(y[i], ref, TOL, f"L={L} row {i}")
        assert np.max(np.abs(y[i] - yc[i])) <= 2e-6 * max(1.0, np.max(np.abs(ref)))


@pytest.mark.parametrize("order,kind,cut", [(2, "lowpass", 1000.0), (4, "highpass", 500.0), (5, "lowpass", 4000.0),
                                            (8, "lowpass", 2000.0), (3, "bandpass", (200.0, 2000.0)), (10, "lowpass", 3000.0)])
def test_sosfiltfilt_section_counts(ops, order, kind, cut):
    """1 ... 4 sections go through the clip-resident kernel, 5 through the chunked one."""
    sos = O.design_butterworth_sos(cut, 48000.0, order, kind)
    pad = O.sosfiltfilt_padlen(sos)
    rng = np.random.default_rng(order)
    x = (rng.normal(0, 0.5, (2, 24000)) + np.linspace(-1, 1, 24000)).astype(np.float32)
    y = ops.sosfiltfilt(ops.to_device_f32(x), sos, O.sosfilt_zi(sos), pad).cpu().numpy()
    for i in range(2):
        assert_parity(y[i], O.apply_sos_filter(sos, x[i].astype(np.float64)), TOL, f"order {order} {kind} row {i}")


@pytest.mark.parametrize("kind", ["ellip", "cheby2", "scaled"])
def test_sosfiltfilt_general_numerators(ops, kind):
    """The clip-resident kernel has a form for cascades whose sections behind the first have b0 = b2 = 1 (zeros on the unit
    circle with the gain in the first section, as zpk2sos lays out Butterworth, elliptic and Chebyshev-II designs: b1 is
    free) and a general one, which a cascade with the gain moved between its sections takes: the same oracle on both sides
    of the switch."""
    import scipy.signal
    if kind == "ellip":
        sos = scipy.signal.ellip(6, 0.5, 60.0, [300.0, 3400.0], btype="bandpass", fs=48000.0, output="sos")[:4]
    elif kind == "cheby2":
        sos = scipy.signal.cheby2(6, 50.0, 2500.0, btype="lowpass", fs=48000.0, output="sos")
    else:
        sos = O.design_butterworth_sos((300.0, 3400.0), 48000.0, 4, "bandpass").copy()
        sos[0, :3] *= 4.0; sos[2, :3] *= 0.25
    sos = np.ascontiguousarray(sos, dtype=np.float64)
    pad = O.sosfiltfilt_padlen(sos)
    rng = np.random.default_rng(len(kind))
    x = (rng.normal(0, 0.5, (3, 30000)) + np.sin(np.arange(30000) * 0.02)).astype(np.float32)
    y = ops.sosfiltfilt(ops.to_device_f32(x), sos, O.sosfilt_zi(sos), pad).cpu().numpy()
    for i in range(3):
        assert_parity(y[i], O.apply_sos_filter(sos, x[i].astype(np.float64)), TOL, f"{kind} row {i}")


@pytest.mark.parametrize("L", [65536 - 54, 65537, 200001, 1 << 20])
def test_sosfiltfilt_long_signals(ops, L):
    """More than 256 chunks per signal: every thread of the chunk-state prefix owns several consecutive chunks
    (M^q by squaring in LDS); lengths around the 256-chunk boundary and a ragged last thread."""
    sos = O.design_butterworth_sos((300.0, 3400.0), 48000.0, 4, "bandpass")
    rng = np.random.default_rng(L)
    t = np.arange(L) / 48000.0
    x = np.stack([rng.normal(0, 0.3, L) + np.sin(2 * np.pi * 1000.0 * t), rng.normal(0, 1.0, L) + 0.5]).astype(np.float32)
    y = ops.sosfiltfilt(ops.to_device_f32(x), sos, O.sosfilt_zi(sos), 27).cpu().numpy()
    for i in range(2):
        assert_parity(y[i], O.apply_sos_filter(sos, x[i].astype(np.float64)), TOL, f"L={L} row {i}")
    sos8 = O.design_butterworth_sos(1000.0, 48000.0, 8, "lowpass")          # four sections, other poles
    y8 = ops.sosfiltfilt(ops.to_device_f32(x[:1]), sos8, O.sosfilt_zi(sos8), O.sosfiltfilt_padlen(sos8)).cpu().numpy()
    assert_parity(y8[0], O.apply_sos_filter(sos8, x[0].astype(np.float64)), TOL, f"L={L} lowpass 8")


WELCH = [("w4096", dict(nperseg=4096)), ("w256", dict(nperseg=256)), ("w1024o768", dict(nperseg=1024, noverlap=768)),
         ("w512nfft1024", dict(nperseg=512, nfft=1024)), ("w1024spec", dict(nperseg=1024, scaling="spectrum")),
         ("w1024nodet", dict(nperseg=1024, detrend=False)), ("w1024hamming", dict(nperseg=1024, window="hamming"))]


@pytest.mark.parametrize("tag,kw", WELCH)
def test_welch_vs_reference_golden(ops, tag, kw):
    import scipy.signal
    g = np.load(os.path.join(G, "ref_dsp.npz"))
    x = g["x20000"]
    nperseg = kw["nperseg"]; nfft = kw.get("nfft", nperseg); nov = kw.get("noverlap", nperseg // 2)
    w = scipy.signal.get_window(kw.get("window", "hann"), nperseg)
    scale = 1.0 / (48000.0 * (w * w).sum()) if kw.get("scaling", "density") == "density" else 1.0 / w.sum() ** 2
    xb = np.stack([x, 2 * x]).astype(np.float32)
    p = ops.welch(ops.to_device_f32(xb), nperseg, nov, nfft, w, kw.get("detrend", "constant") == "constant",
                  scale).cpu().numpy()
    assert_parity(p[0], g[f"welch_{tag}_p"], TOL, tag)
    assert_parity(p[1], 4 * g[f"welch_{tag}_p"], TOL, tag + " (x2 -> x4)")


@pytest.mark.parametrize("n_fft,n_mels,L", [(256, 13, 4000), (1024, 128, 16000), (4096, 40, 30001), (500, 70, 5003)])
def test_mel_dense_mfma_shapes(ops, n_fft, n_mels, L):
    """The dense filterbank on the matrix cores (frame lengths the fused kernel does not take): row counts that do not
    fill a 16-row tile, more than one group of four tiles (128 mels), F = n_fft/2 + 1 with and without a partial last
    group of 16 bins, frame counts that are not multiples of 16 -- against the float64 product."""
    from sygnals_amd import _tables as T
    rng = np.random.default_rng(n_fft + n_mels)
    F = n_fft // 2 + 1
    Tn = 1 + L // (n_fft // 4)
    P = (rng.random((3, Tn, F)) ** 4 * 100.0).astype(np.float32)          # wide dynamic range, non-negative
    basis = T.mel_filterbank(16000, n_fft, n_mels)
    mel = ops.mel_dense(torch.from_numpy(P).cuda(), torch.from_numpy(basis).cuda()).cpu().numpy()
    want = np.einsum("mf,btf->bmt", basis.astype(np.float64), P.astype(np.float64))
    assert mel.shape == (3, n_mels, Tn)
    for b in range(3):
        assert_parity(mel[b], want[b], TOL, f"dense mel n_fft={n_fft} n_mels={n_mels}")


# ---- fused kernel for the other power-of-two frame lengths (stft_mel_pow2.hip): the reference's own tests use 1024 / 256
@pytest.mark.parametrize("sr,hop,n_mels,L,center", [(48000, 256, 40, 48000, True), (16000, 256, 40, 16000, True),
                                                     (44100, 333, 64, 9000, True), (48000, 512, 80, 20000, False),
                                                     (22050, 256, 40, 700, True), (16000, 256, 26, 5000, True)])
def test_frame_length_1024_segment_sum_kernel(ops, sr, hop, n_mels, L, center):
    """syg_stft_mel_w1024_seg_f32 (free-running waves, two frames per wave transform, mel by segment sums with a two-row
    table) against the float64 oracle and the dense-matrix kernel: odd frame counts (a pair with one frame), odd hops,
    center=False, clips shorter than a frame."""
    Y = O.synth_clips(9, L, sr, seed=hop + n_mels)
    Y[2] *= 1e-3
    Y[5][:] = 0.0
    y = ops.to_device_f32(Y)
    assert ops.w1024_segtab(sr, n_mels) is not None
    mel = ops.stft_mel_w1024_seg(y, sr, hop, center, "hann", None, n_mels).cpu().numpy()
    dense = ops.stft_mel_pow2(y, sr, 1024, hop, center, "hann", None, n_mels).cpu().numpy()
    assert mel.shape == dense.shape
    for b in range(len(Y)):
        P = np.abs(O.stft(Y[b].astype(np.float64), 1024, hop, 1024, "hann", center)) ** 2
        want = O.melspectrogram(P, sr, 1024, n_mels)
        assert_parity(mel[b], want, TOL, f"mel n_fft=1024 sr={sr} clip {b}")
        assert_parity(mel[b], dense[b], TOL, f"segment sums vs dense matrix, clip {b}")
    again = ops.stft_mel_w1024_seg(y, sr, hop, center, "hann", None, n_mels).cpu().numpy()
    assert np.array_equal(mel, again)


@pytest.mark.parametrize("M,T,K,lifter", [(40, 188, 13, 0.0), (40, 751, 13, 22.0), (128, 94, 20, 0.0), (40, 1203, 13, 0.0), (7, 5, 7, 0.0)])
def test_mel_mfcc_equals_logmel_dct_without_touching_hbm_db(ops, M, T, K, lifter):
    """syg_mel_mfcc_f32: the MFCCs of syg_logmel_dct_f32, bit for bit, without the dB matrix in HBM (clips that fit the LDS;
    the 40 x 1203 case takes the in-place form)."""
    rng = np.random.default_rng(M + T)
    mel = (rng.random((6, M, T)) ** 6 * 50.0).astype(np.float32)
    mel[3] = 0.0
    a = ops.to_device_f32(mel.reshape(6, -1)).reshape(6, M, T)
    b = a.clone()
    _, want = ops.logmel_dct(a, K, lifter=lifter)
    got = ops.mel_mfcc(b, K, lifter=lifter)
    assert torch.equal(got, want)
    ref = np.stack([O.mfcc(S=O.power_to_db(m.astype(np.float64), ref=np.max), n_mfcc=K, lifter=lifter) for m in mel])
    assert_parity(got.cpu().numpy(), ref, TOL, "mel_mfcc vs oracle")


@pytest.mark.parametrize("n_fft,sr,hop,n_mels,L,center", [(512, 48000, 128, 40, 48000, True), (256, 48000, 64, 40, 24000, True),
                                                           (512, 16000, 100, 40, 7000, True), (256, 16000, 64, 40, 5001, False),
                                                           (256, 22050, 33, 40, 3000, True), (512, 8000, 128, 26, 300, True),
                                                           (256, 8000, 64, 26, 100, True)])
def test_frame_lengths_512_256_segment_sum_kernel(ops, n_fft, sr, hop, n_mels, L, center):
    """syg_stft_mel_wseg_small_f32 (free-running waves, four / eight frames per wave transform, mel by segment sums with a
    four-row table) against the float64 oracle and the dense-matrix kernel: frame counts that are no multiple of the
    group, odd hops, center=False, clips shorter than a frame."""
    Y = O.synth_clips(9, L, sr, seed=hop + n_mels)
    Y[2] *= 1e-3
    Y[5][:] = 0.0
    y = ops.to_device_f32(Y)
    assert ops.wsmall_segtab(sr, n_fft, n_mels) is not None
    mel = ops.stft_mel_wseg_small(y, sr, n_fft, hop, center, "hann", None, n_mels).cpu().numpy()
    dense = ops.stft_mel_pow2(y, sr, n_fft, hop, center, "hann", None, n_mels).cpu().numpy()
    assert mel.shape == dense.shape
    for b in range(len(Y)):
        P = np.abs(O.stft(Y[b].astype(np.float64), n_fft, hop, n_fft, "hann", center)) ** 2
        want = O.melspectrogram(P, sr, n_fft, n_mels)
        assert_parity(mel[b], want, TOL, f"mel n_fft={n_fft} sr={sr} clip {b}")
        assert_parity(mel[b], dense[b], TOL, f"segment sums vs dense matrix, clip {b}")
    again = ops.stft_mel_wseg_small(y, sr, n_fft, hop, center, "hann", None, n_mels).cpu().numpy()
    assert np.array_equal(mel, again)


@pytest.mark.parametrize("sr,hop,n_mels,L,center", [(48000, 1024, 40, 48000, True), (16000, 1024, 40, 30001, True),
                                                     (44100, 410, 40, 20000, True), (48000, 2048, 64, 25000, False),
                                                     (22050, 1024, 128, 22050, True), (48000, 1024, 40, 3000, True)])
def test_frame_length_4096_fused_mel_and_mfcc(ops, sr, hop, n_mels, L, center):
    """frame_length 4096 (syg_stft_mel_w4096_f32: one wave per frame, mel by segment sums with a four-pass table) against
    the float64 oracle: mel power and the MFCCs mfcc_batch builds from it; aligned and unaligned hops, center=False,
    clips shorter than a frame, more bands."""
    Y = O.synth_clips(7, L, sr, seed=hop + n_mels)
    Y[2] *= 1e-3
    Y[5][:] = 0.0
    y = ops.to_device_f32(Y)
    assert ops.w4096_segtab(sr, n_mels) is not None
    mel = ops.stft_mel_w4096(y, sr, hop, center, "hann", None, n_mels).cpu().numpy()
    for b in range(len(Y)):
        P = np.abs(O.stft(Y[b].astype(np.float64), 4096, hop, 4096, "hann", center)) ** 2
        want = O.melspectrogram(P, sr, 4096, n_mels)
        assert mel[b].shape == want.shape
        assert_parity(mel[b], want, TOL, f"mel n_fft=4096 sr={sr} clip {b}")
    got = ops.mfcc_batch(y, sr, 4096, hop, n_mels, 13, center).cpu().numpy()
    want = O.mfcc_batch(Y, sr, 4096, hop, n_mels, 13, center)
    gen = ops.mfcc_batch(y, sr, 4096, hop, n_mels, 13, center, fused=False).cpu().numpy()       # the generic chain
    for b in range(len(Y)):
        assert_parity(got[b], want[b], TOL, f"mfcc n_fft=4096 clip {b}")
        assert_parity(got[b], gen[b], TOL, f"fused vs generic clip {b}")


@pytest.mark.parametrize("n_fft,hop,n_mels,L,center", [
    (1024, 256, 40, 48000, True),      # tests/test_features_manager.py:183-220 shape, 1 s @ 48 kHz
    (1024, 256, 128, 16000, True),     # librosa's default n_mels
    (512, 128, 40, 22051, True),       # odd length, T not a multiple of 16 / 32
    (512, 200, 64, 30000, False),      # center=False, hop that is not a multiple of 4, 64 mels
    (512, 128, 40, 700, True),         # a clip of 1.4 frames
    (256, 64, 13, 4000, True),         # the CLI's small frame
    (256, 64, 40, 48000, True),        # tests/test_features_manager.py's other frame length at 1 s @ 48 kHz
    (256, 100, 64, 12345, False),      # center=False, odd everything
    (1024, 300, 40, 9000, False),      # center=False, hop that does not divide anything
    (64, 16, 8, 1000, True),           # the smallest frame
    (1024, 512, 20, 600, True),        # clip shorter than a frame
])
def test_mfcc_batch_other_frame_lengths(ops, n_fft, hop, n_mels, L, center):
    """mfcc_batch for n_fft != 2048: ONE launch (clip-resident), against the float64 oracle at 1e-5; the mel matrix the
    same launch can emit, the tile form (mel only) and the generic four-launch chain agree with it."""
    sr = 48000 if L >= 40000 else 16000
    Y = O.synth_clips(5, L, sr, seed=n_fft + hop)
    Y[3] *= 1e-3
    y = ops.to_device_f32(Y)
    n_mfcc = min(13, n_mels)
    want = O.mfcc_batch(Y, sr, n_fft, hop, n_mels, n_mfcc, center)
    got = ops.mfcc_batch(y, sr, n_fft, hop, n_mels, n_mfcc, center, fused=True).cpu().numpy()      # one launch
    assert got.shape == want.shape
    dflt = ops.mfcc_batch(y, sr, n_fft, hop, n_mels, n_mfcc, center).cpu().numpy()                 # default: one or two launches
    assert peak_rel(dflt, got) <= 2e-6
    for b in range(len(Y)):
        assert_parity(got[b], want[b], TOL, f"mfcc n_fft={n_fft} clip {b}")
    assert ops.mfcc_pow2_fits(n_fft, n_mels, want.shape[2], n_mfcc)
    mf1, mel1 = ops.stft_mfcc_pow2(y, sr, n_fft, hop, center, "hann", None, n_mels, n_mfcc, keep_mel=True)
    assert np.array_equal(mf1.cpu().numpy(), got)
    mel_t = ops.stft_mel_pow2(y, sr, n_fft, hop, center, "hann", None, n_mels).cpu().numpy()
    if n_fft in (256, 512):        # (clip form: LDS Stockham transform; tile form: four / eight frames per wave transform)
        assert peak_rel(mel1.cpu().numpy(), mel_t) <= 2e-6
    else:
        assert np.array_equal(mel1.cpu().numpy(), mel_t)                     # clip form and tile form: same bits
    S = [np.abs(O.stft(Y[b].astype(np.float64), n_fft, hop, n_fft, "hann", center)) ** 2 for b in range(len(Y))]
    for b in range(len(Y)):
        assert_parity(mel_t[b], O.melspectrogram(S[b], sr, n_fft, n_mels), TOL, f"mel n_fft={n_fft} clip {b}")
    generic = ops.mfcc_batch(y, sr, n_fft, hop, n_mels, n_mfcc, center, fused=False).cpu().numpy()
    assert peak_rel(generic, got) <= 2e-6
    # magnitude mel (power = 1), the manager's melspec 'power' parameter
    mel_m = ops.stft_mel_pow2(y, sr, n_fft, hop, center, "hann", None, n_mels, power=1).cpu().numpy()
    for b in (0, 3):
        assert_parity(mel_m[b], O.mel_filterbank(sr, n_fft, n_mels).astype(np.float64) @ np.sqrt(S[b]), TOL, "magnitude mel")


def test_mfcc_other_frame_length_long_clip_two_launch(ops):
    """A clip whose mel matrix does not fit the LDS beside the transform buffers takes the tile kernel + logmel_dct."""
    sr, n_fft, hop, n_mels = 16000, 1024, 64, 128
    Y = O.synth_clips(2, 60000, sr, seed=5)
    assert not ops.mfcc_pow2_fits(n_fft, n_mels, 1 + 60000 // hop, 13)
    got = ops.mfcc_batch(ops.to_device_f32(Y), sr, n_fft, hop, n_mels, 13).cpu().numpy()
    want = O.mfcc_batch(Y, sr, n_fft, hop, n_mels, 13)
    for b in range(2):
        assert_parity(got[b], want[b], TOL, "long clip")


def test_segment_kernels_random_shapes_against_the_generic_chain(ops):
    """Forty seeded random shapes (frame length, rate, bands, hop -- also above the frame length --, clip length, center)
    through the segment-sum mel kernels of the other frame lengths against the generic chain (complex STFT -> |X|^2 ->
    dense filterbank); shapes whose filterbank has no piece table must say so (None), not fail."""
    rng = np.random.default_rng(2024)
    done = 0
    for _ in range(40):
        n_fft = int(rng.choice([256, 512, 1024, 4096]))
        sr = int(rng.choice([8000, 16000, 22050, 44100, 48000]))
        n_mels = int(rng.choice([20, 26, 40, 48]))
        hop = int(rng.integers(1, 2 * n_fft)) if rng.random() < 0.3 else int(rng.choice([n_fft // 4, n_fft // 2, n_fft // 8]))
        center = bool(rng.random() < 0.7)
        L = int(rng.integers(n_fft if not center else 1, 6 * n_fft))
        B = int(rng.integers(1, 7))
        Y = (rng.normal(0, 0.3, (B, L)) * rng.random((B, 1))).astype(np.float32)
        y = ops.to_device_f32(Y)
        mel = ops.stft_mel_segments(y, sr, n_fft, hop, center, "hann", None, n_mels, 0.0, None)
        if mel is None:
            continue
        P = ops.cabs_pow(ops.stft_any(y, n_fft, hop, center, "hann"), 2)
        want = ops.mel_dense(P, ops.mel_config(sr, n_fft, n_mels, 0.0, None).basis).cpu().numpy()
        assert mel.shape == want.shape, (n_fft, sr, n_mels, hop, center, L)
        assert_parity(mel.cpu().numpy(), want, TOL, f"n_fft={n_fft} sr={sr} n_mels={n_mels} hop={hop} center={center} L={L}")
        done += 1
    assert done >= 15


def test_segment_entry_points_refuse_bad_arguments(ops):
    """The C entry points of the segment-sum kernels answer bad arguments with an error code and a message, not a launch:
    a table of the wrong size, a frame count that does not follow from (L, hop, center), too many bands, null pointers."""
    import ctypes as C
    from sygnals_amd._lib import lib, SygnalsHipError, check
    y = ops.to_device_f32(np.zeros((2, 8000), np.float32))
    st = C.c_void_p(ops._stream_ptr())
    out = torch.empty((2, 40, 64), dtype=torch.float32, device=y.device)
    win1k, tw1k = ops.window_dev("hann", 1024, 1024), ops.twiddle_dev(1024)
    tab1k = ops.w1024_segtab(16000, 40)
    T1k = ops.num_frames(8000, 1024, 256, True)
    def call(fn, *a):
        rc = fn(*a)
        assert rc != 0
        with pytest.raises(SygnalsHipError):
            check(rc, "x")
    L = lib()
    call(L.syg_stft_mel_w1024_seg_f32, ops._ptr(y), 2, 8000, 8000, 256, 1, T1k, ops._ptr(win1k), ops._ptr(tw1k), ops._ptr(tab1k),
         int(tab1k.numel()) - 4, 40, ops._ptr(out), st)                                         # table size
    call(L.syg_stft_mel_w1024_seg_f32, ops._ptr(y), 2, 8000, 8000, 256, 1, T1k + 1, ops._ptr(win1k), ops._ptr(tw1k), ops._ptr(tab1k),
         int(tab1k.numel()), 40, ops._ptr(out), st)                                             # frame count
    call(L.syg_stft_mel_w1024_seg_f32, ops._ptr(y), 2, 8000, 8000, 256, 1, T1k, ops._ptr(win1k), ops._ptr(tw1k), None,
         int(tab1k.numel()), 40, ops._ptr(out), st)                                             # null table
    tab256 = ops.wsmall_segtab(16000, 256, 40)
    win256 = ops.window_dev("hann", 256, 256)
    T256 = ops.num_frames(8000, 256, 64, True)
    call(L.syg_stft_mel_wseg_small_f32, ops._ptr(y), 2, 8000, 8000, 128, 64, 1, T256, ops._ptr(win256), ops._ptr(tw1k), ops._ptr(tab256),
         int(tab256.numel()), 40, ops._ptr(out), st)                                            # frame length
    call(L.syg_stft_mel_wseg_small_f32, ops._ptr(y), 2, 8000, 8000, 256, 64, 1, T256, ops._ptr(win256), ops._ptr(tw1k), ops._ptr(tab256),
         int(tab256.numel()), 49, ops._ptr(out), st)                                            # more bands than the tile holds
    tab4k = ops.w4096_segtab(16000, 40)
    call(L.syg_stft_mel_w4096_f32, ops._ptr(y), 2, 8000, 8000, 1024, 1, 3, ops._ptr(ops.window_dev("hann", 4096, 4096)),
         ops._ptr(ops.twiddle_dev(4096)), ops._ptr(tab4k), int(tab4k.numel()), 40, ops._ptr(out), st)      # frame count
    assert ops.wsmall_segtab(16000, 256, 49) is None and ops.wsmall_segtab(16000, 128, 40) is None
    with pytest.raises(SygnalsHipError):
        ops.stft_mel_wseg_small(y, 16000, 256, 64, True, "hann", None, 64)
