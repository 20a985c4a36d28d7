"""Host-side logic of the product (tables, plans, validation) -- CPU only, no GPU calls."""
import os
import re

import numpy as np
import pytest
import scipy.fft
import scipy.signal
from numpy.testing import assert_allclose, assert_array_equal

from oracle import cpu_ref as O
from sygnals_amd import _tables as T

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("sr,n_fft,n_mels", [(48000, 2048, 40), (16000, 2048, 128), (22050, 2048, 128),
                                             (48000, 1024, 64), (8000, 512, 20)])
def test_mel_filterbank_bit_identical_to_oracle(sr, n_fft, n_mels):
    a, b = T.mel_filterbank(sr, n_fft, n_mels), O.mel_filterbank(sr, n_fft, n_mels)
    assert a.dtype == np.float32 and a.shape == b.shape
    assert_array_equal(a, b)
    assert_array_equal(T.mel_filterbank(sr, n_fft, n_mels, 100.0, sr / 4), O.mel_filterbank(sr, n_fft, n_mels, 100.0, sr / 4))


@pytest.mark.parametrize("waves", [8, 16])
@pytest.mark.parametrize("n_mels", [1, 13, 16, 40, 64, 100, 128, 256])
def test_mel_plan_reconstructs_dense_basis(n_mels, waves):
    """The block-sparse packing for v_mfma_f32_4x4x1_16b_f32: replaying the kernel's addressing (slot = lane >> 4, row =
    lane & 3, step i at row position p0 + i of the skewed power row) over every (wave, step, lane) rebuilds the dense
    filterbank exactly; pad positions and everything past a chunk carry zero weights."""
    W = T.mel_filterbank(48000, 2048, n_mels)
    if n_mels == 256 and waves == 8:
        # 64 groups of four rows need 64 slots = 16 waves: a clear error (callers then take the dense path), no hang
        with pytest.raises(ValueError, match="non-empty groups"):
            T.pack_mel_plan(W, waves)
        return
    wp, plan = T.pack_mel_plan(W, waves)
    assert plan.tolist()[:2] == [2, waves] and plan.shape == (5,) and wp.dtype == np.float32
    steps, ng, toff = int(plan[2]), int(plan[3]), int(plan[4])
    assert steps % 4 == 0 and steps >= T.MEL_MIN_STEPS and ng == (n_mels + 3) // 4
    assert toff == waves * steps * 64 + 8 * 64 * 4 and wp.size == toff + 256
    assert not wp[waves * steps * 64:toff].any()                       # the zero groups the kernel may pre-load
    tab = wp[toff:].view(np.int32)
    p0, sg, gf, gc = tab[:64], tab[64:128], tab[128:192], tab[192:256]
    A = wp[:waves * steps * 64].reshape(waves, steps // 4, 64, 4)       # [wave][group][lane][step in group]
    R = np.zeros((4 * ng, 1025), np.float64)
    for slot in range(4 * waves):
        w, s = divmod(slot, 4)
        for i in range(steps):
            for h in range(4):                                          # the four frame groups see the same weights
                a = A[w, i // 4, 16 * s + 4 * h: 16 * s + 4 * h + 4, i % 4]
                assert_array_equal(a, A[w, i // 4, 16 * s: 16 * s + 4, i % 4])
            a = A[w, i // 4, 16 * s: 16 * s + 4, i % 4].astype(np.float64)
            p = int(p0[slot]) + i
            if sg[slot] < 0 or p % 17 == 16 or p - p // 17 > 1024:
                assert not a.any()                                      # unused slot / pad word / past the last bin
                continue
            R[4 * sg[slot]: 4 * sg[slot] + 4, p - p // 17] += a
    assert_array_equal(R[:n_mels], W.astype(np.float64))
    assert not R[n_mels:].any()
    # the slots of a group are consecutive and in ascending position order
    for g in range(ng):
        sl = np.arange(gf[g], gf[g] + gc[g])
        assert (sg[sl] == g).all() and (np.diff(p0[sl]) == steps).all()
    assert gc[:ng].sum() == (sg >= 0).sum() <= 4 * waves


def test_mel_plan_never_loops_when_groups_outnumber_slots():
    """Round-2 advisor finding: with more non-empty groups than slots the step search could never succeed."""
    for n_mels in (132, 160, 200, 256):
        with pytest.raises(ValueError, match="non-empty groups"):
            T.pack_mel_plan(T.mel_filterbank(48000, 2048, n_mels), 8)
    T.pack_mel_plan(T.mel_filterbank(48000, 2048, 128), 8)          # 32 groups, 32 slots: fits


def test_mel_plan_rejects_too_many_mels():
    with pytest.raises(ValueError, match="max 64"):
        T.pack_mel_plan(np.ones((257, 1025), np.float32), 16)
    with pytest.raises(ValueError, match="waves must be 8 or 16"):
        T.pack_mel_plan(np.ones((16, 1025), np.float32), 4)


def test_dct_matrix_and_lifter():
    rng = np.random.default_rng(0)
    S = rng.normal(size=(40, 9))
    D = T.dct_matrix(13, 40)
    assert_allclose(D.astype(np.float64) @ S, scipy.fft.dct(S, axis=0, type=2, norm="ortho")[:13], atol=1e-5)
    assert_allclose(T.dct_matrix(20, 40, norm=None).astype(np.float64) @ S, scipy.fft.dct(S, axis=0, type=2)[:20],
                    rtol=1e-5, atol=1e-4)
    with pytest.raises(ValueError):
        T.dct_matrix(41, 40)
    assert T.lifter_weights(13, 0.0) is None
    assert_allclose(T.lifter_weights(13, 22.0), 1 + 11 * np.sin(np.pi * np.arange(1, 14) / 22), rtol=1e-6)
    with pytest.raises(ValueError, match="non-negative"):
        T.lifter_weights(13, -1.0)


@pytest.mark.parametrize("sr,n_fft", [(48000, 2048), (16000, 2048), (22050, 1024)])
def test_contrast_plan_matches_oracle_bands(sr, n_fft):
    fr = np.fft.rfftfreq(n_fft, 1 / sr)
    plan = T.contrast_plan(fr, sr)
    bands = O.contrast_bands(fr, sr)
    assert plan[0] == len(bands) == 7
    for b, (bins, kk) in enumerate(bands):
        assert plan[1 + b] == bins[0] and plan[1 + 16 + b] == bins[-1] + 1 and plan[1 + 32 + b] == kk
        assert_array_equal(bins, np.arange(bins[0], bins[-1] + 1))


def test_contrast_plan_errors():
    fr = np.fft.rfftfreq(512, 1 / 8000)
    with pytest.raises(ValueError, match="Nyquist"):
        T.contrast_plan(fr, 8000)
    with pytest.raises(ValueError, match="quantile"):
        T.contrast_plan(fr, 48000, quantile=1.5)
    with pytest.raises(ValueError, match="fmin"):
        T.contrast_plan(fr, 48000, fmin=0.0)


def test_window_and_twiddles():
    assert_array_equal(T.analysis_window("hann", 2048, 2048), scipy.signal.get_window("hann", 2048, fftbins=True))
    w = T.analysis_window("hamming", 400, 512)
    assert w.shape == (512,) and (w[:56] == 0).all() and (w[456:] == 0).all()
    assert_array_equal(w, O.fft_window("hamming", 400, 512))
    with pytest.raises(ValueError):
        T.analysis_window("hann", 600, 512)
    tw = T.twiddles(2048)
    k = np.arange(2048)
    assert_allclose(tw[:, 0] + 1j * tw[:, 1], np.exp(-2j * np.pi * k / 2048), atol=1e-7)


def test_filters_host_side_matches_reference_behaviour():
    from sygnals_amd.core import filters as F
    g = np.load(os.path.join(ROOT, "tests", "golden", "ref_filters.npz"))
    assert_array_equal(F.design_butterworth_sos((300.0, 3400.0), 48000.0, 4, "bandpass"), g["bp4_48k_sos"])
    assert_array_equal(F.design_butterworth_sos(100.0, 1000.0, 5, "lowpass"), g["lp5_1k_sos"])
    for name in ("bp4_48k", "lp5_1k", "hp5_1k", "bs5_1k", "lp8_1k", "bp2_16k"):
        sos = g[f"{name}_sos"]
        assert_allclose(F._steady_state(sos), scipy.signal.sosfilt_zi(sos), rtol=1e-9, atol=1e-12)
        assert T.butter_padlen(sos) == O.sosfiltfilt_padlen(sos)
    # error strings pinned by the reference's tests/test_filters.py:73-87
    with pytest.raises(ValueError, match="strictly between 0 and Nyquist"):
        F.design_butterworth_sos(500.0, 1000.0, 4, "lowpass")
    with pytest.raises(ValueError, match="strictly between 0 and Nyquist"):
        F.design_butterworth_sos(600.0, 1000.0, 4, "lowpass")
    with pytest.raises(ValueError, match="Low cutoff .* must be less than high cutoff"):
        F.design_butterworth_sos((200.0, 100.0), 1000.0, 4, "bandpass")
    with pytest.raises(TypeError, match="cutoff must be a float .* or a tuple"):
        F.design_butterworth_sos([100.0], 1000.0, 4, "lowpass")
    with pytest.raises(ValueError, match="must be a 1D array"):
        F.apply_sos_filter(g["lp5_1k_sos"], np.zeros((2, 100)))
    with pytest.raises(ValueError, match=r"shape \(n_sections, 6\)"):
        F.apply_sos_filter(np.zeros((2, 5)), np.zeros(100))


def test_api_validation_without_gpu():
    """Argument errors are raised before any device work (same messages as the reference)."""
    from sygnals_amd.core import dsp as D
    from sygnals_amd.core.features import cepstral, frequency_domain as fd, manager
    with pytest.raises(ValueError, match="Input data must be a 1D array."):
        D.compute_fft(np.zeros((2, 8)))
    with pytest.raises(ValueError, match="Input spectrum must be a 1D array."):
        D.compute_ifft(np.zeros((2, 8), complex))
    with pytest.raises(ValueError, match="Input data must be a 1D array."):
        D.compute_stft(np.zeros((2, 8)))
    with pytest.raises(ValueError, match="Input data must be a 1D array."):
        D.compute_psd_welch(np.zeros((2, 8)))
    with pytest.raises(ValueError, match="Invalid window type 'nope'"):
        D.apply_window(np.zeros(8), "nope")
    with pytest.raises(ValueError, match="Invalid window type 'nope'"):
        D.compute_fft(np.zeros(8), window="nope")
    with pytest.raises(ValueError, match="Either audio time series 'y' or Mel spectrogram 'S' must be provided."):
        cepstral.mfcc()
    with pytest.raises(ValueError, match="Sampling rate 'sr' must be provided"):
        cepstral.mfcc(y=np.zeros(100))
    with pytest.raises(ValueError, match="must match"):
        fd.spectral_centroid(np.zeros(4), np.zeros(5))
    with pytest.raises(ValueError, match="roll_percent must be between 0.0 and 1.0."):
        fd.spectral_rolloff(np.ones(4), np.ones(4), roll_percent=1.5)
    with pytest.raises(ValueError, match="must be positive"):
        fd.spectral_bandwidth(np.ones(4), np.ones(4), p=0)
    with pytest.raises(ValueError, match="Input S must be a 2D spectrogram"):
        fd.spectral_contrast(np.ones(5), 48000)
    e = np.array([], dtype=np.float64)     # empty-input constants, reference tests/test_features_freq.py
    assert fd.spectral_centroid(e, e) == 0.0 and fd.spectral_bandwidth(e, e) == 0.0
    assert fd.spectral_flatness(e) == 0.0 and fd.spectral_rolloff(e, e) == 0.0 and fd.dominant_frequency(e, e) == 0.0
    with pytest.raises(ValueError, match=r"Unknown feature\(s\) requested: \['nope'\]"):
        manager.extract_features(np.zeros(4096), 16000, ["nope"])
    with pytest.raises(ValueError, match="must be a 1D array"):
        manager.extract_features(np.zeros((2, 4096)), 16000, ["mfcc"])
    out = manager.extract_features(np.zeros(100), 16000, ["mfcc"], center=False, output_format="dict_of_arrays")
    assert set(out) == {"time"} and out["time"].size == 0
    assert manager.extract_features(np.zeros(100), 16000, ["mfcc"], center=False).empty


def test_frame_count_rule():
    from sygnals_amd import ops
    for L, n, h, c in [(48000, 2048, 512, True), (160000, 2048, 512, True), (512, 1024, 256, True), (100, 2048, 512, True),
                       (100, 2048, 512, False), (4096, 1024, 256, False)]:
        assert ops.num_frames(L, n, h, c) == O.num_frames(L, n, h, c)


def test_statistics_route_condition_mirrors_the_c_side():
    """manager.py picks syg_stft2048_stats_f32 only where that entry point accepts the call (hop <= 512, L < 2^28, fewer
    than 2^24 frames): a very long clip falls through to the mel route instead of raising."""
    from sygnals_amd import ops
    assert ops.stft2048_stats_fits(512, 48000) and ops.stft2048_stats_fits(1, 48000)
    assert not ops.stft2048_stats_fits(513, 48000)
    assert not ops.stft2048_stats_fits(512, 1 << 28) and ops.stft2048_stats_fits(512, (1 << 28) - 1)
    assert not ops.stft2048_stats_fits(1, (1 << 24) + 5)


def test_library_options_and_settings_do_not_read_the_environment(monkeypatch):
    """The library's switches are syg_set_option values, the Python side's are ops.settings attributes: no getenv in the
    product sources, and an environment variable of the old names changes nothing."""
    from sygnals_amd import ops
    for dirpath, _, files in os.walk(os.path.join(ROOT, "sygnals_amd", "csrc")):
        for f in files:
            assert "getenv" not in open(os.path.join(dirpath, f)).read(), f
    monkeypatch.setenv("SYGNALS_AMD_RESERVE_CUS", "32")
    monkeypatch.setenv("SYGNALS_AMD_WAVES", "8")
    assert ops.get_option("reserved_cus") == 0 and ops.fused_waves() == 16
    with ops.override(reserved_cus=16, stft_load=1, sos_clip=0, cqt_staged=2, cqt_mode="fft", waves=8):
        assert [ops.get_option(k) for k in ("reserved_cus", "stft_load", "sos_clip", "cqt_staged")] == [16, 1, 0, 2]
        assert ops.settings.cqt_mode == "fft" and ops.fused_waves() == 8
    assert [ops.get_option(k) for k in ("reserved_cus", "stft_load", "sos_clip", "cqt_staged")] == [0, -1, 1, -1]
    assert ops.settings.cqt_mode == "bf16x3" and ops.fused_waves() == 16
    with pytest.raises(ops.SygnalsHipError):
        ops.set_option("stft_load", 3)
    with pytest.raises(KeyError):
        with ops.override(no_such_switch=1):
            pass


def test_product_never_imports_the_oracle():
    """The product path must not route through oracle/ (or any CPU fallback)."""
    pat = re.compile(r"^\s*(from|import)\s+oracle\b|cpu_ref", re.M)
    for dirpath, _, files in os.walk(os.path.join(ROOT, "sygnals_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                assert not pat.search(open(os.path.join(dirpath, f)).read()), f"{f} references the oracle"


def test_ops_fail_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from sygnals_amd import ops
    from sygnals_amd._lib import SygnalsHipError
    with pytest.raises(SygnalsHipError, match="no CPU fallback"):
        ops.require_gpu()
    from sygnals_amd.core import dsp as D
    with pytest.raises(SygnalsHipError):
        D.compute_fft(np.zeros(64))


@pytest.mark.parametrize("sr,hop,n_bins", [(48000, 512, 84), (22050, 512, 60), (16000, 256, 84), (44100, 1024, 40)])
def test_cqt_plan_matches_oracle(sr, hop, n_bins):
    from sygnals_amd._cqt import CqtPlan, decimation_taps
    if sr == 16000 and n_bins == 84:
        n_bins = 72
    p, q = CqtPlan(sr, hop, n_bins=n_bins), O.cqt_plan(sr, hop, n_bins=n_bins)
    assert p.early == q["early"] and len(p.octaves) == len(q["octaves"])
    end = n_bins
    for a, b in zip(p.octaves, q["octaves"]):
        nb = b["basis"].shape[0]
        rows = np.arange(max(end - nb, 0), end)
        ref = (b["basis"][-len(rows):] if end < nb else b["basis"]) * q["scale"][rows][:, None]
        assert a["n_fft"] == b["n_fft"] and a["hop"] == b["hop"] and a["row0"] == rows[0] and a["n"] == len(rows)
        assert_allclose(a["basis"], ref, rtol=1e-12, atol=1e-15)
        end -= nb
    # the same filter as the oracle's, except that the 20 vanishing taps of the half-band design (~1e-18 out of
    # firwin) are exact zeros on the device side, so that the kernel can skip them
    taps, ref_taps = decimation_taps(), O.cqt_decimation_taps()
    assert_allclose(taps, ref_taps, rtol=0, atol=1e-16)
    assert (taps == 0).sum() == 20 and np.array_equal(taps[taps != 0], ref_taps[taps != 0])
    with pytest.raises(ValueError, match="Nyquist"):
        CqtPlan(8000)
    with pytest.raises(ValueError, match="multiple of 2"):
        CqtPlan(48000, hop_length=48)


def test_wave_fft_index_model_is_exact_and_conflict_free():
    """tools/wave_fft_model_v4.py is the lane/register model the kernel's exchange maps were derived from: it must
    reproduce numpy's rfft and keep every exchange access conflict-free under the per-instruction LDS banking."""
    import importlib.util
    import numpy as np
    spec = importlib.util.spec_from_file_location("wfm", os.path.join(ROOT, "tools", "wave_fft_model_v4.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    rng = np.random.default_rng(1)
    x = rng.normal(size=2048)
    win = 0.5 - 0.5 * np.cos(2 * np.pi * np.arange(2048) / 2048)
    X = m.model(x, win)
    assert not np.isnan(X).any() and np.abs(X - np.fft.rfft(x * win)).max() < 1e-10
    lane = np.arange(64)
    cl, bp, hi = lane >> 2, lane & 3, lane >> 5
    G16 = [list(range(i, i + 16)) for i in range(0, 64, 16)]
    G32 = [list(range(32)), list(range(32, 64))]
    assert all(m.conf(8 * m.x1(c, lane), 8, G16, 128) == 1 for c in range(16))
    for h in (0, 1):
        row = (cl & 7) + 8 * h
        assert all(m.conf(8 * m.x1(row, 4 * (8 * hi + i) + bp), 8, G32, 256) == 1 for i in range(8))
    assert all(m.conf(8 * m.x2(cl, cp, bp), 8, G16, 128) == 1 for cp in range(16))


# ---------------------------------------------------------------- segment-sum projection (stft_mel.hip MODE 6): host side
@pytest.mark.parametrize("sr,n_mels,fmin,fmax", [(48000, 40, 0.0, None), (22050, 40, 0.0, None), (16000, 40, 0.0, None),
                                                 (44100, 40, 0.0, None), (48000, 40, 300.0, 8000.0), (8000, 26, 50.0, None),
                                                 (48000, 24, 0.0, None)])
def test_mel_segment_table_reproduces_the_filterbank(sr, n_mels, fmin, fmax):
    """pack_mel_segments: the affine pieces stand for exactly the float32 filterbank the oracle multiplies by, the float32
    model of the kernel's sums (segments_project) agrees with the float64 product, and the windows of the 32 lanes of an
    LDS access start in 32 different banks."""
    from sygnals_amd import _tables as T
    W = T.mel_filterbank(sr, 2048, n_mels, fmin, fmax)
    tab = T.pack_mel_segments(sr, 2048, n_mels, fmin, fmax, basis=W)
    assert tab.shape == (2, 2, 64, 4) and tab.dtype == np.float32
    Wr = T.segments_weights(tab, n_mels, 1025)
    assert np.abs(Wr - W).max() <= 2e-7 * W.max()
    rng = np.random.default_rng(sr + n_mels)
    for P in (rng.random(1025) ** 8 * 100.0, np.ones(1025), np.eye(1025)[640] + 1e-6):
        ref = W.astype(np.float64) @ P
        got = T.segments_project(tab, P.astype(np.float32), n_mels)
        assert np.abs(got - ref).max() <= 1e-6 * ref.max()
        nz = ref > 1e-9 * ref.max()
        assert (np.abs(got - ref)[nz] / ref[nz]).max() <= 2e-6
    assert T.segments_read_cycles(tab) == 68
    ti = tab.view(np.int32)
    w0 = ti[:, 0, :, 0]
    hi, lead = (w0 >> 16) & 0xFF, (w0 >> 24) & 7
    live = hi > 0
    assert (lead[live] <= T.SEG_LEAD_MAX).all() and (hi[live] <= 17).all() and (hi[live] >= T.SEG_LEAD_MAX).all()
    assert (lead[~live] == 7).all()
    bands = ti[:, 0, :, 1].ravel()
    assert sorted(bands[bands >= 0].tolist()) == list(range(n_mels))      # every band is stored by exactly one lane


def test_mel_segment_table_refuses_what_it_cannot_hold():
    from sygnals_amd import _tables as T
    with pytest.raises(ValueError, match="lane slots"):
        T.pack_mel_segments(48000, 2048, 128)
    with pytest.raises(ValueError, match="more than 16 blocks"):
        T.pack_mel_segments(48000, 2048, 4)
    W = T.mel_filterbank(48000, 2048, 40)
    W2 = W.copy(); W2[7, 200] += 1e-4                      # not triangular any more
    with pytest.raises(ValueError, match="not reproduced"):
        T.pack_mel_segments(48000, 2048, 40, basis=W2)


@pytest.mark.parametrize("n_fft,kw,window,row_base", [
    (1024, dict(rows=2, row_words=568, n_pass=2), 17, 4),                 # stft_mel_w1024_seg.hip: two rows, two passes each
    (512, dict(rows=4, row_words=296, n_pass=1), 17, 4),                  # stft_mel_wseg_small.hip <4>
    (256, dict(rows=4, row_words=160, n_pass=1, block=8), 9, 4),          # <8>: 8-bin pieces, 9-word windows
])
@pytest.mark.parametrize("sr,n_mels", [(48000, 40), (16000, 40), (22050, 26)])
def test_mel_segment_row_tables_of_the_other_frame_lengths(n_fft, kw, window, row_base, sr, n_mels):
    """pack_mel_segments_rows: every row's block stands for the float32 filterbank, its windows stay inside the row and
    start in distinct banks, the band words carry the row tag, and no row's last lane stores a band."""
    from sygnals_amd import _tables as T
    W = T.mel_filterbank(sr, n_fft, n_mels)
    tab = T.pack_mel_segments_rows(sr, n_fft, n_mels, basis=W, **kw)
    rows, n_pass = kw["rows"], kw["n_pass"]
    assert tab.shape == (rows * n_pass, 2, 64, 4)
    ti = tab.view(np.int32)
    F = n_fft // 2 + 1
    rng = np.random.default_rng(n_fft + sr)
    P = (rng.random(F) ** 6 * 10.0).astype(np.float32)
    ref = W.astype(np.float64) @ P
    for r in range(rows):
        blk = tab[r * n_pass:(r + 1) * n_pass].copy()
        bi = blk.view(np.int32)
        bw = bi[:, 0, :, 1]
        assert ((bw[bw >= 0] >> 8) == r).all()
        bi[:, 0, :, 1] = np.where(bw >= 0, bw & 255, -1)                    # take the row tag and the row offset back
        w0 = bi[:, 0, :, 0]
        bi[:, 0, :, 0] = (w0 & ~0xFFFF) | ((w0 & 0xFFFF) - 4 * r * kw["row_words"])
        assert np.abs(T.segments_weights(blk, n_mels, F, row_base) - W).max() <= 2e-7 * W.max()
        got = T.segments_project(blk, P, n_mels, row_base)
        assert np.abs(got - ref).max() <= 1e-6 * ref.max()
        nz = ref > 1e-4 * ref.max()      # (a band whose filter is nearly empty -- narrower than a bin -- is right to 1e-7 of the
        assert (np.abs(got - ref)[nz] / ref[nz]).max() <= 3e-6          # row's scale, not of its own value)
        live = ((bi[:, 0, :, 0] >> 16) & 0xFF) > 0
        off = (bi[:, 0, :, 0] & 0xFFFF) // 4
        assert (off[live] >= 0).all() and (off[live] + window <= kw["row_words"]).all()
        assert T.segments_read_cycles(blk, window) <= (2 * n_pass + 1) * window      # at most one group with one conflict
        assert int(ti[(r + 1) * n_pass - 1, 0, 63, 1]) < 0


def test_mel_segment_table_four_passes_for_frame_length_4096():
    from sygnals_amd import _tables as T
    W = T.mel_filterbank(48000, 4096, 40)
    tab = T.pack_mel_segments(48000, 4096, 40, basis=W, n_pass=4)
    assert tab.shape == (4, 2, 64, 4)
    assert np.abs(T.segments_weights(tab, 40, 2049) - W).max() <= 2e-7 * W.max()
    assert T.segments_read_cycles(tab) == 4 * 34
    with pytest.raises(ValueError):
        T.pack_mel_segments(48000, 4096, 128, n_pass=4)


@pytest.mark.parametrize("sr,n_mels,fmin,fmax", [(48000, 128, 0.0, None), (16000, 128, 0.0, None), (22050, 128, 0.0, None),
                                                 (44100, 64, 0.0, None), (48000, 64, 0.0, None), (48000, 96, 100.0, 12000.0),
                                                 (8000, 128, 0.0, None), (48000, 40, 0.0, None), (32000, 160, 0.0, None)])
def test_mel_segment_table_four_passes_for_the_default_filterbank(sr, n_mels, fmin, fmax):
    """The tile form of the 2048 kernel (MODE 8 / 9, syg_stft2048_mel_tri_f32) reads a FOUR-pass table built with
    row_base = 4: the reference's default of 128 bands (manager.py:214) and the other filterbanks without a two-pass table
    are reproduced to 2e-7 of the largest weight; the float32 model of the kernel's sums agrees with the float64 product;
    a window starts at most 4 words in front of bin 0 and ends at most 16 words behind the row's 1090 (masked words: they
    lie in the next wave's row, or in the 16 slack words behind the last row -- inside the allocation either way)."""
    from sygnals_amd import _tables as T
    W = T.mel_filterbank(sr, 2048, n_mels, fmin, fmax)
    tab = T.pack_mel_segments(sr, 2048, n_mels, fmin, fmax, basis=W, n_pass=4, row_base=4)
    assert tab.shape == (4, 2, 64, 4) and tab.dtype == np.float32
    assert np.abs(T.segments_weights(tab, n_mels, 1025, 4) - W).max() <= 2e-7 * W.max()
    rng = np.random.default_rng(sr + n_mels)
    for P in (rng.random(1025) ** 8 * 100.0, np.ones(1025)):
        ref = W.astype(np.float64) @ P
        got = T.segments_project(tab, P.astype(np.float32), n_mels, 4)
        assert np.abs(got - ref).max() <= 1e-6 * ref.max()
        # a band is right to 1e-7 of the powers under its filter times the filter's peak -- the row's local scale -- not to
        # 1e-7 of its own value: a T0 + b T1 cancels where the power sits at the zero end of an edge (narrow filters, a
        # spectrum of 8 decades as here: 2e-5 of a band 3 decades down, i.e. 1e-4 dB; the matrix form is there for callers
        # who need such bands to the last bit)
        nz = ref > 1e-4 * ref.max()
        assert (np.abs(got - ref)[nz] / ref[nz]).max() <= 5e-5
    ti = tab.view(np.int32)
    w0 = ti[:, 0, :, 0]
    live = ((w0 >> 16) & 0xFF) > 0
    off = (w0 & 0xFFFF) // 4
    assert (off[live] >= 0).all() and (off[live] + 17 <= 4 + 1090 + 16).all()
    bands = ti[:, 0, :, 1].ravel()
    assert sorted(bands[bands >= 0].tolist()) == list(range(n_mels))
    assert int(ti[3, 0, 63, 1]) < 0                                        # the last lane never stores a band


def test_cqt_one_launch_shape():
    """Which plans syg_cqt_fused_f32 takes (ops.cqt asks CqtPlan.one_launch_shape): compute_cqt's usual call at 48 / 44.1 kHz
    and its variants with fewer octaves; not a plan without (22.05 kHz) or with more (hop 1024, 60 bins) early
    decimations, another frame length (32 kHz) or a clipped last octave."""
    from sygnals_amd._cqt import CqtPlan
    assert CqtPlan(48000, 512, None, 84).one_launch_shape()
    assert CqtPlan(44100, 512, None, 84).one_launch_shape()
    assert CqtPlan(48000, 512, 523.25, 36).one_launch_shape()
    assert CqtPlan(48000, 512, 261.63, 48).one_launch_shape()
    assert not CqtPlan(22050, 512, None, 84).one_launch_shape()
    assert not CqtPlan(48000, 1024, None, 84).one_launch_shape()
    assert not CqtPlan(48000, 512, None, 60).one_launch_shape()
    assert not CqtPlan(32000, 512, None, 72).one_launch_shape()
    assert not CqtPlan(48000, 512, None, 80).one_launch_shape()      # 80 bins: the lowest octave is clipped to 8 filters


def test_cqt_decimator_is_what_the_one_launch_kernel_assumes():
    """syg_cqt_fused_f32 reads the centre tap and the ten odd-offset taps of one side (cqt_fused.hip: a symmetric half-band):
    the filter the plan hands it is exactly that -- zero at every even offset from the centre, symmetric to the last bit
    after the float32 rounding the device applies."""
    from sygnals_amd._cqt import decimation_taps
    h = decimation_taps()
    assert h.shape == (41,)
    assert all(h[20 + d] == 0.0 and h[20 - d] == 0.0 for d in range(2, 21, 2))
    h32 = h.astype(np.float32)
    assert np.array_equal(h32, h32[::-1])
    assert abs(float(h.sum()) - 1.0) < 1e-12 and abs(float(h[20]) - 0.5) < 1e-3
