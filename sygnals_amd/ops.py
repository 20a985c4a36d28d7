"""Batched device operators: thin Python over the C ABI.

torch is used only as the device-array container (allocation, H2D/D2H, streams); every
arithmetic step runs in libsygnals_hip.so.  All functions take/return CUDA(ROCm) float32
tensors with a leading batch axis; there is no CPU path.
"""
from __future__ import annotations

import ctypes as C
import os
import functools
from typing import Optional

import numpy as np
import torch

from . import _tables as T
from ._lib import SygnalsHipError, check, lib


# ------------------------------------------------------------------ device plumbing
def require_gpu() -> torch.device:
    if not torch.cuda.is_available():
        raise SygnalsHipError("sygnals_amd needs an AMD GPU (torch.cuda.is_available() is False); "
                              "there is no CPU fallback")
    return torch.device("cuda", torch.cuda.current_device())


def _stream_ptr() -> int:
    return torch.cuda.current_stream().cuda_stream


def _ld(t: torch.Tensor) -> int:
    """Row stride in elements (a size-1 leading axis may carry an arbitrary stride)."""
    return int(t.stride(0)) if t.shape[0] > 1 else int(t.shape[1])


def _ptr(t: Optional[torch.Tensor]):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def to_device_f32(x, device=None) -> torch.Tensor:
    """Host array / tensor -> contiguous float32 device tensor (2-D: [B, L])."""
    device = device or require_gpu()
    if isinstance(x, torch.Tensor):
        t = x.to(device=device, dtype=torch.float32)
    else:
        t = torch.from_numpy(np.ascontiguousarray(np.asarray(x, dtype=np.float32))).to(device)
    return t.contiguous()


_dev_cache: dict = {}


def _cached(key, builder):
    dev = torch.cuda.current_device()
    k = (dev,) + key
    v = _dev_cache.get(k)
    if v is None:
        v = builder()
        _dev_cache[k] = v
    return v


def _dev(arr: np.ndarray) -> torch.Tensor:
    return torch.from_numpy(np.ascontiguousarray(arr)).to(require_gpu())


def _window_key(window, win_length, n_fft):
    if isinstance(window, (np.ndarray, list)):
        a = np.asarray(window, dtype=np.float64)
        return ("arr", a.tobytes(), win_length, n_fft)
    return (window, win_length, n_fft)


def window_dev(window, win_length, n_fft) -> torch.Tensor:
    return _cached(("win",) + _window_key(window, win_length, n_fft),
                   lambda: _dev(T.analysis_window(window, win_length, n_fft).astype(np.float32)))


def twiddle_dev(n) -> torch.Tensor:
    return _cached(("tw", n), lambda: _dev(T.twiddles(n)))


def num_frames(L: int, n_fft: int, hop: int, center: bool) -> int:
    """Frame-count rule of sygnals/core/features/manager.py:149-157."""
    if center:
        return 1 + L // hop
    return 1 + (L - n_fft) // hop if L >= n_fft else 0


# ------------------------------------------------------------------ fused 2048 path
class _Settings:
    """Process-wide switches between kernels that the tests hold to the same results (plain attributes: nothing is read
    from the environment; tests and tools set them with `ops.override(...)`):
      waves                 16 (one workgroup per CU) | 8: waves per workgroup of the matrix-form fused 2048 kernels
      cqt_mode              "bf16x3" (default) | "gemm" | "fft": octave kernel of compute_cqt
      cqt_streams           1 | 2: octave products on a side stream (measured: no gain)
      cqt_chain             True: up to three decimation levels per pass; False: one launch per level
      cqt_fused             True: the one-launch form (syg_cqt_fused_f32) where the plan has its shape; False: level by level
      one_launch_features   True: extract_features routes MFCC + statistics / contrast requests to the one-launch kernels
    Options that live in the library (syg_set_option): reserved_cus, stft_load, sos_clip, cqt_staged."""
    waves = T.WAVES
    cqt_mode = "bf16x3"
    cqt_streams = 1
    cqt_chain = True
    cqt_fused = True
    one_launch_features = True


settings = _Settings()
_LIB_OPTIONS = {"reserved_cus": 0, "stft_load": 1, "sos_clip": 2, "cqt_staged": 3}      # SYG_OPT_* of include/sygnals_hip.h


def set_option(name: str, value: int) -> None:
    """syg_set_option by name (reserved_cus | stft_load | sos_clip | cqt_staged)."""
    check(lib().syg_set_option(_LIB_OPTIONS[name], int(value)), "syg_set_option")


def get_option(name: str) -> int:
    return int(lib().syg_get_option(_LIB_OPTIONS[name]))


def set_reserved_cus(n: int) -> None:
    """Leave `n` CUs out of the persistent grids of the fused 2048 kernels (room for a collective's workgroups)."""
    set_option("reserved_cus", n)


class override:
    """Context manager: `with ops.override(cqt_mode="fft", cqt_staged=0): ...` sets Python-side settings and library
    options for the block and puts the previous values back."""

    def __init__(self, **kw):
        self.kw = kw
        self.old = {}

    def __enter__(self):
        for k, v in self.kw.items():
            if k in _LIB_OPTIONS:
                self.old[k] = get_option(k)
                set_option(k, v)
            else:
                if not hasattr(_Settings, k):
                    raise KeyError(k)
                self.old[k] = getattr(settings, k)
                setattr(settings, k, v)
        return self

    def __exit__(self, *exc):
        for k, v in self.old.items():
            if k in _LIB_OPTIONS:
                set_option(k, v)
            else:
                setattr(settings, k, v)
        return False


def fused_waves() -> int:
    """Waves per workgroup of the fused kernel: 8 (two workgroups per CU) or 16 (one)."""
    return int(settings.waves)


class MelConfig:
    """Device tables for one (sr, n_fft, n_mels, fmin, fmax) mel front end."""

    def __init__(self, sr, n_fft, n_mels, fmin, fmax, waves=None):
        basis = T.mel_filterbank(sr, n_fft, n_mels, fmin, fmax)
        self.basis_host = basis
        self.n_mels = n_mels
        waves = fused_waves() if waves is None else waves
        self.wpacked = None
        self.plan = None
        if n_fft == 2048 and n_mels <= 256:
            try:
                wp, plan = T.pack_mel_plan(basis, waves)
            except ValueError:
                pass        # more groups of four rows than the plan has slots (8-wave mode, n_mels > 128): dense path
            else:
                self.wpacked = _dev(wp)
                self.plan = np.ascontiguousarray(plan, dtype=np.int32)
        self.basis = _dev(basis)
        # piece table of the per-wave projection by segment sums (MODE 6 of the fused kernel): triangular filterbanks
        # whose pieces fit 128 lane slots (n_mels = 40 at any usual rate; 64 and more do not)
        self.segtab = None
        # ... and the four-pass table of the tile form (MODE 8 / 9: up to 256 pieces -- the reference's default of 128
        # bands, 64 ... 128 bands at the usual rates; row_base 4: a short first piece needs room for its lead)
        self.segtab4 = None
        if n_fft == 2048:
            try:
                self.segtab = _dev(T.pack_mel_segments(sr, n_fft, n_mels, fmin, fmax, basis=basis).reshape(-1))
            except ValueError:
                pass
            if n_mels <= 255:
                try:
                    self.segtab4 = _dev(T.pack_mel_segments(sr, n_fft, n_mels, fmin, fmax, basis=basis, n_pass=4,
                                                            row_base=4).reshape(-1))
                except ValueError:
                    pass


def mel_config(sr, n_fft, n_mels, fmin=0.0, fmax=None, waves=None) -> MelConfig:
    fmax = sr / 2.0 if fmax is None else fmax
    waves = fused_waves() if waves is None else waves
    return _cached(("mel", float(sr), n_fft, n_mels, float(fmin), float(fmax), waves),
                   lambda: MelConfig(sr, n_fft, n_mels, fmin, fmax, waves))


def fused_mel_ok(sr, n_fft, n_mels, fmin=0.0, fmax=None) -> bool:
    """True when the fused frame-length-2048 kernel has a block-sparse plan or a four-pass piece table for this filterbank
    (n_mels <= 256, and <= 128 in the 8-wave development mode); otherwise callers take the generic chain
    (stft_any -> mel_dense)."""
    if n_fft != 2048 or not 1 <= n_mels <= 256:
        return False
    cfg = mel_config(sr, n_fft, n_mels, fmin, fmax)
    return cfg.wpacked is not None or cfg.segtab4 is not None


def fused_pow2_ok(n_fft, n_mels) -> bool:
    """True when the fused kernel for the other power-of-two frame lengths (stft_mel_pow2.hip) takes this shape."""
    return is_pow2(n_fft) and 64 <= n_fft <= 1024 and 1 <= n_mels <= 256


def _basis_padded(cfg: "MelConfig", n_fft: int):
    """[16 ceil(M / 16), Fp] zero-padded dense filterbank (Fp = F rounded up to a multiple of 16) on the device."""
    if getattr(cfg, "basis_padded", None) is None:
        M, F = cfg.basis_host.shape
        Mp, Fp = -(-M // 16) * 16, -(-F // 16) * 16
        bp = np.zeros((Mp, Fp), np.float32)
        bp[:M, :F] = cfg.basis_host
        cfg.basis_padded, cfg.Fp = _dev(bp), Fp
    return cfg.basis_padded, cfg.Fp


def _pow2_front(y, sr, n_fft, hop, center, window, win_length, n_mels, fmin, fmax):
    require_gpu()
    if y.dim() != 2 or y.dtype != torch.float32 or not y.is_cuda:
        raise ValueError("y must be a float32 CUDA tensor of shape [B, L]")
    if not fused_pow2_ok(n_fft, n_mels):
        raise SygnalsHipError(f"no fused kernel for n_fft={n_fft}, n_mels={n_mels} (powers of two 64 ... 1024; 2048: stft2048_mel)")
    if y.stride(1) != 1:
        y = y.contiguous()
    B, L = y.shape
    Tn = num_frames(L, n_fft, hop, center)
    if Tn <= 0:
        raise ValueError("signal too short for one frame")
    cfg = mel_config(sr, n_fft, n_mels, fmin, fmax)
    bp, Fp = _basis_padded(cfg, n_fft)
    win = window_dev(window, n_fft if win_length is None else win_length, n_fft)
    # (n_fft 512 / 256: the twiddle block is followed by W_1024^k -- four / eight frames share one 1024-point wave transform)
    tw = twiddle_rfft_dev(n_fft) if n_fft not in (256, 512) else _cached(("twr+1024", n_fft), lambda: _dev(np.concatenate(
        [T.twiddles(n_fft), T.twiddles(n_fft // 2), T.twiddles(1024)], axis=0)))
    return y, B, L, Tn, bp, Fp, win, tw


def stft_mel_pow2(y: torch.Tensor, sr: float, n_fft: int, hop: int, center: bool = True, window="hann", win_length=None,
                  n_mels: int = 128, fmin: float = 0.0, fmax=None, power: int = 2) -> torch.Tensor:
    """Fused STFT(n_fft = 64 ... 1024) -> |X|^power -> mel [B, n_mels, T]: one launch, no spectrogram in HBM."""
    y, B, L, Tn, bp, Fp, win, tw = _pow2_front(y, sr, n_fft, hop, center, window, win_length, n_mels, fmin, fmax)
    mel = torch.empty((B, n_mels, Tn), dtype=torch.float32, device=y.device)
    rc = lib().syg_stft_mel_pow2_f32(_ptr(y), B, L, y.stride(0), n_fft, hop, int(center), Tn, _ptr(win), _ptr(tw), _ptr(bp),
                                     Fp, n_mels, int(power), _ptr(mel), C.c_void_p(_stream_ptr()))
    check(rc, "syg_stft_mel_pow2_f32")
    return mel


def mfcc_pow2_fits(n_fft, n_mels, n_frames, n_mfcc) -> bool:
    return bool(lib().syg_stft_mfcc_pow2_fits(int(n_fft), int(n_mels), int(n_frames), int(n_mfcc)))


def stft_mfcc_pow2(y: torch.Tensor, sr: float, n_fft: int, hop: int, center: bool = True, window="hann", win_length=None,
                   n_mels: int = 128, n_mfcc: int = 13, fmin: float = 0.0, fmax=None, lifter: float = 0.0,
                   amin: float = 1e-10, top_db: Optional[float] = 80.0, keep_mel: bool = False):
    """One launch for the other power-of-two frame lengths: [B, L] clips -> MFCC [B, n_mfcc, T] (a workgroup owns a
    clip; its mel matrix stays in LDS).  Returns (mfcc, mel_power | None)."""
    y, B, L, Tn, bp, Fp, win, tw = _pow2_front(y, sr, n_fft, hop, center, window, win_length, n_mels, fmin, fmax)
    if not mfcc_pow2_fits(n_fft, n_mels, Tn, n_mfcc):
        raise SygnalsHipError("stft_mfcc_pow2: the clip's mel matrix does not fit the LDS; use stft_mel_pow2 + logmel_dct")
    dct = _cached(("dct", n_mfcc, n_mels, 2, "ortho"), lambda: _dev(T.dct_matrix(n_mfcc, n_mels, 2, "ortho")))
    lw = T.lifter_weights(n_mfcc, lifter)
    lif = None if lw is None else _dev(lw)
    mf = torch.empty((B, n_mfcc, Tn), dtype=torch.float32, device=y.device)
    mel = torch.empty((B, n_mels, Tn), dtype=torch.float32, device=y.device) if keep_mel else None
    rc = lib().syg_stft_mfcc_pow2_f32(_ptr(y), B, L, y.stride(0), n_fft, hop, int(center), Tn, _ptr(win), _ptr(tw), _ptr(bp),
                                      Fp, n_mels, _ptr(dct), n_mfcc, None if lif is None else _ptr(lif), float(amin),
                                      -1.0 if top_db is None else float(top_db), 1, 1.0,
                                      None if mel is None else _ptr(mel), _ptr(mf), C.c_void_p(_stream_ptr()))
    check(rc, "syg_stft_mfcc_pow2_f32")
    return mf, mel


def w4096_segtab(sr, n_mels, fmin=0.0, fmax=None):
    """Four-pass piece table of the frame-length-4096 kernel for this filterbank on the device, or None."""
    fmax = sr / 2.0 if fmax is None else fmax

    def build():
        try:
            basis = T.mel_filterbank(sr, 4096, n_mels, fmin, fmax)
            return _dev(T.pack_mel_segments(sr, 4096, n_mels, fmin, fmax, basis=basis, n_pass=4).reshape(-1))
        except ValueError:
            return False
    tab = _cached(("seg4096", float(sr), n_mels, float(fmin), float(fmax)), build)
    return None if tab is False else tab


def w1024_segtab(sr, n_mels, fmin=0.0, fmax=None):
    """Two-row piece table of the frame-length-1024 segment-sum kernel for this filterbank on the device, or None."""
    fmax = sr / 2.0 if fmax is None else fmax

    def build():
        try:
            basis = T.mel_filterbank(sr, 1024, n_mels, fmin, fmax)
            return _dev(T.pack_mel_segments_rows(sr, 1024, n_mels, fmin, fmax, basis=basis).reshape(-1))
        except ValueError:
            return False
    tab = _cached(("seg1024", float(sr), n_mels, float(fmin), float(fmax)), build)
    return None if tab is False else tab


def stft_mel_w1024_seg(y: torch.Tensor, sr: float, hop: int = 256, center: bool = True, window="hann", win_length=None,
                       n_mels: int = 128, fmin: float = 0.0, fmax=None) -> torch.Tensor:
    """frame_length 1024, power 2: [B, L] clips -> mel power [B, n_mels, T] in one launch, free-running waves (two frames
    per wave transform, mel by segment sums).  Raises SygnalsHipError when the filterbank has no piece table."""
    require_gpu()
    if y.dim() != 2 or y.dtype != torch.float32 or not y.is_cuda:
        raise ValueError("y must be a float32 CUDA tensor of shape [B, L]")
    if y.stride(1) != 1:
        y = y.contiguous()
    tab = w1024_segtab(sr, n_mels, fmin, fmax)
    if tab is None:
        raise SygnalsHipError("stft_mel_w1024_seg: no piece table for this filterbank (use stft_mel_pow2)")
    B, L = y.shape
    Tn = num_frames(L, 1024, hop, center)
    if Tn <= 0:
        raise ValueError("signal too short for one frame")
    out = torch.empty((B, n_mels, Tn), dtype=torch.float32, device=y.device)
    rc = lib().syg_stft_mel_w1024_seg_f32(_ptr(y), B, L, _ld(y), hop, int(center), Tn,
                                          _ptr(window_dev(window, win_length or 1024, 1024)), _ptr(twiddle_dev(1024)), _ptr(tab),
                                          int(tab.numel()), n_mels, _ptr(out), C.c_void_p(_stream_ptr()))
    check(rc, "syg_stft_mel_w1024_seg_f32")
    return out


def stft_rows_w1024(y: torch.Tensor, sr: float, hop: int = 256, center: bool = True, window="hann", win_length=None,
                    n_mels: Optional[int] = None, fmin: float = 0.0, fmax=None, want_stats=False, roll_percent: float = 0.85,
                    bw_p: float = 2.0, contrast: Optional[np.ndarray] = None):
    """frame_length 1024: the statistics / contrast rows of stft2048_mel from the segment-sum kernel's launch
    (syg_stft_rows_w1024_f32), with the mel power block when n_mels is given (needs a piece table: w1024_segtab).
    Returns (mel [B, M, T] | None, stats [B, 8, T] | None, contrast_pv [B, 2, R, T] | None)."""
    smask = 31 if want_stats is True else int(want_stats or 0)
    require_gpu()
    if y.dim() != 2 or y.dtype != torch.float32 or not y.is_cuda:
        raise ValueError("y must be a float32 CUDA tensor of shape [B, L]")
    if not smask and contrast is None:
        raise ValueError("stft_rows_w1024: no statistics requested (stft_mel_w1024_seg gives the mel block alone)")
    if y.stride(1) != 1:
        y = y.contiguous()
    tab = None
    if n_mels is not None:
        tab = w1024_segtab(sr, n_mels, fmin, fmax)
        if tab is None:
            raise SygnalsHipError("stft_rows_w1024: no piece table for this filterbank")
    B, L = y.shape
    Tn = num_frames(L, 1024, hop, center)
    if Tn <= 0:
        raise ValueError("signal too short for one frame")
    mel = torch.empty((B, n_mels, Tn), dtype=torch.float32, device=y.device) if tab is not None else None
    stats = torch.zeros((B, 8, Tn), dtype=torch.float32, device=y.device) if smask else None
    cpv = cplan_p = None
    if contrast is not None:
        cplan = np.ascontiguousarray(contrast, dtype=np.int32)
        cpv = torch.empty((B, 2, int(cplan[0]), Tn), dtype=torch.float32, device=y.device)
        cplan_p = cplan.ctypes.data_as(C.c_void_p)
    rc = lib().syg_stft_rows_w1024_f32(_ptr(y), B, L, _ld(y), hop, int(center), Tn,
                                       _ptr(window_dev(window, win_length or 1024, 1024)), _ptr(twiddle_dev(1024)), _ptr(tab),
                                       int(tab.numel()) if tab is not None else 0, int(n_mels or 0), _ptr(mel), float(sr),
                                       float(roll_percent), float(bw_p), smask, _ptr(stats), cplan_p, _ptr(cpv),
                                       C.c_void_p(_stream_ptr()))
    check(rc, "syg_stft_rows_w1024_f32")
    return mel, stats, cpv


_SMALL_ROW_WORDS = {512: 296, 256: 160}


def wsmall_segtab(sr, n_fft, n_mels, fmin=0.0, fmax=None):
    """Four-row piece table of the frame-length-512 / 256 segment-sum kernel for this filterbank on the device, or None."""
    if n_fft not in _SMALL_ROW_WORDS or n_mels > 48:        # (the kernel's [band][16 frames] tile holds 48 bands)
        return None
    fmax = sr / 2.0 if fmax is None else fmax

    def build():
        try:
            basis = T.mel_filterbank(sr, n_fft, n_mels, fmin, fmax)
            return _dev(T.pack_mel_segments_rows(sr, n_fft, n_mels, fmin, fmax, basis=basis, rows=4,
                                                 row_words=_SMALL_ROW_WORDS[n_fft], n_pass=1,
                                                 block=(8 if n_fft == 256 else 16)).reshape(-1))
        except ValueError:
            return False
    tab = _cached(("segsmall", float(sr), n_fft, n_mels, float(fmin), float(fmax)), build)
    return None if tab is False else tab


def stft_mel_wseg_small(y: torch.Tensor, sr: float, n_fft: int, hop: int, center: bool = True, window="hann", win_length=None,
                        n_mels: int = 128, fmin: float = 0.0, fmax=None) -> torch.Tensor:
    """frame_length 512 / 256, power 2: [B, L] clips -> mel power [B, n_mels, T] in one launch, free-running waves (four /
    eight frames per wave transform, mel by segment sums).  Raises SygnalsHipError without a piece table."""
    require_gpu()
    if y.dim() != 2 or y.dtype != torch.float32 or not y.is_cuda:
        raise ValueError("y must be a float32 CUDA tensor of shape [B, L]")
    if y.stride(1) != 1:
        y = y.contiguous()
    tab = wsmall_segtab(sr, n_fft, n_mels, fmin, fmax)
    if tab is None:
        raise SygnalsHipError("stft_mel_wseg_small: no piece table for this frame length / filterbank (use stft_mel_pow2)")
    B, L = y.shape
    Tn = num_frames(L, n_fft, hop, center)
    if Tn <= 0:
        raise ValueError("signal too short for one frame")
    out = torch.empty((B, n_mels, Tn), dtype=torch.float32, device=y.device)
    rc = lib().syg_stft_mel_wseg_small_f32(_ptr(y), B, L, _ld(y), n_fft, hop, int(center), Tn,
                                           _ptr(window_dev(window, win_length or n_fft, n_fft)), _ptr(twiddle_dev(1024)),
                                           _ptr(tab), int(tab.numel()), n_mels, _ptr(out), C.c_void_p(_stream_ptr()))
    check(rc, "syg_stft_mel_wseg_small_f32")
    return out


def stft_rows_wsmall(y: torch.Tensor, sr: float, n_fft: int, hop: int, center: bool = True, window="hann", win_length=None,
                     want_stats=False, roll_percent: float = 0.85, bw_p: float = 2.0, contrast: Optional[np.ndarray] = None):
    """frame_length 512 / 256: the statistics / contrast rows of stft2048_mel from the segment-sum kernel's transform
    (syg_stft_rows_wsmall_f32; nothing is projected).  Returns (stats [B, 8, T] | None, contrast_pv [B, 2, R, T] | None)."""
    smask = 31 if want_stats is True else int(want_stats or 0)
    require_gpu()
    if y.dim() != 2 or y.dtype != torch.float32 or not y.is_cuda:
        raise ValueError("y must be a float32 CUDA tensor of shape [B, L]")
    if n_fft not in (512, 256):
        raise SygnalsHipError("stft_rows_wsmall: frame length must be 512 or 256")
    if not smask and contrast is None:
        raise ValueError("stft_rows_wsmall: no statistics requested")
    if y.stride(1) != 1:
        y = y.contiguous()
    B, L = y.shape
    Tn = num_frames(L, n_fft, hop, center)
    if Tn <= 0:
        raise ValueError("signal too short for one frame")
    stats = torch.zeros((B, 8, Tn), dtype=torch.float32, device=y.device) if smask else None
    cpv = cplan_p = None
    if contrast is not None:
        cplan = np.ascontiguousarray(contrast, dtype=np.int32)
        cpv = torch.empty((B, 2, int(cplan[0]), Tn), dtype=torch.float32, device=y.device)
        cplan_p = cplan.ctypes.data_as(C.c_void_p)
    rc = lib().syg_stft_rows_wsmall_f32(_ptr(y), B, L, _ld(y), n_fft, hop, int(center), Tn,
                                        _ptr(window_dev(window, win_length or n_fft, n_fft)), _ptr(twiddle_dev(1024)), float(sr),
                                        float(roll_percent), float(bw_p), smask, _ptr(stats), cplan_p, _ptr(cpv),
                                        C.c_void_p(_stream_ptr()))
    check(rc, "syg_stft_rows_wsmall_f32")
    return stats, cpv


def stft_rows_w4096(y: torch.Tensor, sr: float, hop: int = 1024, center: bool = True, window="hann", win_length=None,
                    n_mels: Optional[int] = None, fmin: float = 0.0, fmax=None, want_stats=False, roll_percent: float = 0.85,
                    bw_p: float = 2.0, contrast: Optional[np.ndarray] = None):
    """frame_length 4096: the statistics / contrast rows of stft2048_mel from the one-wave-per-frame kernel's launch
    (syg_stft_rows_w4096_f32), with the mel power block when n_mels is given (needs a piece table: w4096_segtab).
    Returns (mel [B, M, T] | None, stats [B, 8, T] | None, contrast_pv [B, 2, R, T] | None)."""
    smask = 31 if want_stats is True else int(want_stats or 0)
    require_gpu()
    if y.dim() != 2 or y.dtype != torch.float32 or not y.is_cuda:
        raise ValueError("y must be a float32 CUDA tensor of shape [B, L]")
    if not smask and contrast is None:
        raise ValueError("stft_rows_w4096: no statistics requested (stft_mel_w4096 gives the mel block alone)")
    if y.stride(1) != 1:
        y = y.contiguous()
    tab = None
    if n_mels is not None:
        tab = w4096_segtab(sr, n_mels, fmin, fmax)
        if tab is None:
            raise SygnalsHipError("stft_rows_w4096: no piece table for this filterbank")
    B, L = y.shape
    Tn = num_frames(L, 4096, hop, center)
    if Tn <= 0:
        raise ValueError("signal too short for one frame")
    mel = torch.empty((B, n_mels, Tn), dtype=torch.float32, device=y.device) if tab is not None else None
    stats = torch.zeros((B, 8, Tn), dtype=torch.float32, device=y.device) if smask else None
    cpv = cplan_p = None
    if contrast is not None:
        cplan = np.ascontiguousarray(contrast, dtype=np.int32)
        cpv = torch.empty((B, 2, int(cplan[0]), Tn), dtype=torch.float32, device=y.device)
        cplan_p = cplan.ctypes.data_as(C.c_void_p)
    rc = lib().syg_stft_rows_w4096_f32(_ptr(y), B, L, _ld(y), hop, int(center), Tn,
                                       _ptr(window_dev(window, win_length or 4096, 4096)), _ptr(twiddle_dev(4096)), _ptr(tab),
                                       int(tab.numel()) if tab is not None else 0, int(n_mels or 0), _ptr(mel), float(sr),
                                       float(roll_percent), float(bw_p), smask, _ptr(stats), cplan_p, _ptr(cpv),
                                       C.c_void_p(_stream_ptr()))
    check(rc, "syg_stft_rows_w4096_f32")
    return mel, stats, cpv


def stft_rows_seg(y, sr, n_fft, hop, center=True, window="hann", win_length=None, n_mels=None, fmin=0.0, fmax=None,
                  want_stats=False, roll_percent=0.85, bw_p=2.0, contrast=None):
    """Statistics / contrast rows (+ the mel power block when n_mels is given) from the segment-sum kernels of frame lengths
    1024 / 4096 (one launch) and 512 / 256 (the rows from one launch, the mel block -- power 2 -- from the projection form's).
    Returns (mel | None, stats | None, contrast_pv | None); raises SygnalsHipError for other frame lengths."""
    if n_fft == 1024:
        return stft_rows_w1024(y, sr, hop, center, window, win_length, n_mels, fmin, fmax, want_stats, roll_percent, bw_p, contrast)
    if n_fft == 4096:
        return stft_rows_w4096(y, sr, hop, center, window, win_length, n_mels, fmin, fmax, want_stats, roll_percent, bw_p, contrast)
    if n_fft in (512, 256):
        stats, cpv = stft_rows_wsmall(y, sr, n_fft, hop, center, window, win_length, want_stats, roll_percent, bw_p, contrast)
        mel = None
        if n_mels is not None:
            mel = stft_mel_segments(y, sr, n_fft, hop, center, window, win_length, n_mels, fmin, fmax)
            if mel is None:
                mel = stft_mel_pow2(y, sr, n_fft, hop, center, window, win_length, n_mels, fmin, fmax)
        return mel, stats, cpv
    raise SygnalsHipError(f"stft_rows_seg: no row functions in the fused kernel of frame length {n_fft}")


def stft_mel_segments(y, sr, n_fft, hop, center, window, win_length, n_mels, fmin, fmax):
    """Mel power by one of the segment-sum kernels of the other frame lengths (1024, 512, 256, 4096), or None when the
    frame length / filterbank has none."""
    if n_fft == 1024 and w1024_segtab(sr, n_mels, fmin, fmax) is not None:
        return stft_mel_w1024_seg(y, sr, hop, center, window, win_length, n_mels, fmin, fmax)
    if n_fft in _SMALL_ROW_WORDS and wsmall_segtab(sr, n_fft, n_mels, fmin, fmax) is not None:
        return stft_mel_wseg_small(y, sr, n_fft, hop, center, window, win_length, n_mels, fmin, fmax)
    if n_fft == 4096 and w4096_segtab(sr, n_mels, fmin, fmax) is not None:
        return stft_mel_w4096(y, sr, hop, center, window, win_length, n_mels, fmin, fmax)
    return None


def stft_mel_w4096(y: torch.Tensor, sr: float, hop: int = 1024, center: bool = True, window="hann", win_length=None,
                   n_mels: int = 128, fmin: float = 0.0, fmax=None) -> torch.Tensor:
    """frame_length 4096: [B, L] clips -> mel power [B, n_mels, T] in one launch (one wave per frame, mel by segment
    sums).  Raises SygnalsHipError when the filterbank has no four-pass piece table."""
    require_gpu()
    if y.dim() != 2 or y.dtype != torch.float32 or not y.is_cuda:
        raise ValueError("y must be a float32 CUDA tensor of shape [B, L]")
    if y.stride(1) != 1:
        y = y.contiguous()
    tab = w4096_segtab(sr, n_mels, fmin, fmax)
    if tab is None:
        raise SygnalsHipError("stft_mel_w4096: no piece table for this filterbank (use the generic chain)")
    B, L = y.shape
    Tn = num_frames(L, 4096, hop, center)
    if Tn <= 0:
        raise ValueError("signal too short for one frame")
    out = torch.empty((B, n_mels, Tn), dtype=torch.float32, device=y.device)
    rc = lib().syg_stft_mel_w4096_f32(_ptr(y), B, L, _ld(y), hop, int(center), Tn, _ptr(window_dev(window, win_length or 4096, 4096)),
                                      _ptr(twiddle_dev(4096)), _ptr(tab), int(tab.numel()), n_mels, _ptr(out),
                                      C.c_void_p(_stream_ptr()))
    check(rc, "syg_stft_mel_w4096_f32")
    return out


def stft2048_mel(y: torch.Tensor, sr: float, hop: int = 512, center: bool = True, window="hann",
                 win_length: int = 2048, n_mels: int = 128, fmin: float = 0.0, fmax=None,
                 want_stats=False, roll_percent: float = 0.85, bw_p: float = 2.0,
                 contrast: Optional[np.ndarray] = None, projection: str = "auto", tri_waves: int = 16):
    """Fused STFT(2048) -> power -> mel.  Returns (mel [B, M, T], stats [B, 8, T] | None,
    contrast_pv [B, 2, R, T] | None).  `want_stats`: False, True (all rows) or a bit mask (1 centroid,
    2 bandwidth, 4 flatness, 8 rolloff, 16 dominant): only the selected rows are computed / written.
    projection: "segments" (syg_stft2048_mel_tri_f32: per-wave segment sums -- the two-pass table where the filterbank has
    one, else the four-pass table -- no barriers in the projection; tri_waves = 16 | 8 waves per workgroup), "matrix"
    (syg_stft2048_mel_f32: block-sparse weights on the matrix cores), "auto": the matrix form wherever it has a plan."""
    smask = 31 if want_stats is True else int(want_stats or 0)
    require_gpu()
    if y.dim() != 2 or y.dtype != torch.float32 or not y.is_cuda:
        raise ValueError("y must be a float32 CUDA tensor of shape [B, L]")
    if y.stride(1) != 1:
        y = y.contiguous()
    B, L = y.shape
    Tn = num_frames(L, 2048, hop, center)
    if Tn <= 0:
        raise ValueError("signal too short for one frame")
    cfg = mel_config(sr, 2048, n_mels, fmin, fmax)
    if projection not in ("auto", "segments", "matrix"):
        raise ValueError("projection must be 'auto', 'segments' or 'matrix'")
    tab = cfg.segtab if cfg.segtab is not None else cfg.segtab4
    tri = tab is not None and fused_waves() == 16 and Tn * n_mels < (1 << 29) and projection != "matrix"
    if projection == "auto" and cfg.wpacked is not None:
        # measured (profiles/r04_rows_start.json vs r04b): the matrix form wins wherever it has a plan -- 128 bands 0.206 ms per
        # 1024 clips (mel + dB / DCT launch) against 0.241 for four passes of masked segment sums (about 380 vector
        # instructions per frame), 64 bands 0.178 against 0.214 -- so "auto" keeps it; the segment form serves the plans the
        # matrix form cannot hold (8-wave development mode beyond 128 bands) and stays selectable
        tri = False
    if projection == "segments" and not tri:
        raise SygnalsHipError("stft2048_mel: no piece table for this filterbank (projection='segments')")
    if not tri and cfg.wpacked is None:
        raise SygnalsHipError("fused path: no block-sparse plan for this filterbank (n_mels <= 256; <= 128 with 8 waves)")
    win = window_dev(window, win_length, 2048)
    tw = twiddle_dev(2048)
    mel = torch.empty((B, n_mels, Tn), dtype=torch.float32, device=y.device)
    stats = torch.zeros((B, 8, Tn), dtype=torch.float32, device=y.device) if smask else None
    cpv = None
    cplan_p = None
    if contrast is not None:
        cplan = np.ascontiguousarray(contrast, dtype=np.int32)
        cpv = torch.empty((B, 2, int(cplan[0]), Tn), dtype=torch.float32, device=y.device)
        cplan_p = cplan.ctypes.data_as(C.c_void_p)
    if tri:
        rc = lib().syg_stft2048_mel_tri_f32(
            _ptr(y), B, L, _ld(y), hop, int(center), Tn, _ptr(win), _ptr(tw), _ptr(tab), int(tab.numel()),
            n_mels, _ptr(mel), float(sr), float(roll_percent), float(bw_p), smask, _ptr(stats), cplan_p, _ptr(cpv),
            int(tri_waves), C.c_void_p(_stream_ptr()))
        check(rc, "syg_stft2048_mel_tri_f32")
        return mel, stats, cpv
    rc = lib().syg_stft2048_mel_f32(
        _ptr(y), B, L, _ld(y), hop, int(center), Tn, _ptr(win), _ptr(tw), _ptr(cfg.wpacked),
        cfg.plan.ctypes.data_as(C.c_void_p), n_mels, _ptr(mel), float(sr), float(roll_percent), float(bw_p),
        smask, _ptr(stats), cplan_p, _ptr(cpv), C.c_void_p(_stream_ptr()))
    check(rc, "syg_stft2048_mel_f32")
    return mel, stats, cpv


def stft2048_stats_fits(hop: int, L: int = 0) -> bool:
    """Whether stft2048_stats takes this call: the C side's conditions (syg_stft2048_stats_f32: staged tiles need
    hop <= 512 and 32-bit byte offsets, L < 2^28; frame count below 2^24) and the 16-wave kernel."""
    return hop <= 512 and L < (1 << 28) and 1 + L // max(hop, 1) < (1 << 24) and fused_waves() == 16


def stft2048_stats(y: torch.Tensor, sr: float, hop: int = 512, center: bool = True, window="hann", win_length: int = 2048,
                   want_stats=True, roll_percent: float = 0.85, bw_p: float = 2.0, contrast: Optional[np.ndarray] = None):
    """The statistics / contrast rows of stft2048_mel WITHOUT the mel spectrogram (syg_stft2048_stats_f32: transform + row
    functions, nothing projected) -- what spectral_centroid / bandwidth / flatness / rolloff / contrast need
    (manager.py:289-343).  Returns (stats [B, 8, T] | None, contrast_pv [B, 2, R, T] | None), bit-identical to
    stft2048_mel's."""
    smask = 31 if want_stats is True else int(want_stats or 0)
    require_gpu()
    if y.dim() != 2 or y.dtype != torch.float32 or not y.is_cuda:
        raise ValueError("y must be a float32 CUDA tensor of shape [B, L]")
    if not smask and contrast is None:
        raise ValueError("stft2048_stats: nothing requested (want_stats and contrast are both empty)")
    if y.stride(1) != 1:
        y = y.contiguous()
    B, L = y.shape
    Tn = num_frames(L, 2048, hop, center)
    if Tn <= 0:
        raise ValueError("signal too short for one frame")
    win = window_dev(window, win_length, 2048)
    tw = twiddle_dev(2048)
    stats = torch.zeros((B, 8, Tn), dtype=torch.float32, device=y.device) if smask else None
    cpv = None
    cplan_p = None
    if contrast is not None:
        cplan = np.ascontiguousarray(contrast, dtype=np.int32)
        cpv = torch.empty((B, 2, int(cplan[0]), Tn), dtype=torch.float32, device=y.device)
        cplan_p = cplan.ctypes.data_as(C.c_void_p)
    rc = lib().syg_stft2048_stats_f32(_ptr(y), B, L, _ld(y), hop, int(center), Tn, _ptr(win), _ptr(tw), float(sr),
                                      float(roll_percent), float(bw_p), smask or 1, _ptr(stats), cplan_p, _ptr(cpv),
                                      C.c_void_p(_stream_ptr()))
    check(rc, "syg_stft2048_stats_f32")
    return stats, cpv


def stft2048_c2c(y: torch.Tensor, hop: int = 512, center: bool = True, window="hann", win_length: int = 2048):
    """Complex STFT, frame-major [B, T, 1025, 2] float32."""
    require_gpu()
    if y.stride(1) != 1:
        y = y.contiguous()
    B, L = y.shape
    Tn = num_frames(L, 2048, hop, center)
    if Tn <= 0:
        raise ValueError("signal too short for one frame")
    win = window_dev(window, win_length, 2048)
    tw = twiddle_dev(2048)
    out = torch.empty((B, Tn, 1025, 2), dtype=torch.float32, device=y.device)
    rc = lib().syg_stft2048_c2c_f32(_ptr(y), B, L, _ld(y), hop, int(center), Tn, _ptr(win), _ptr(tw),
                                    _ptr(out), C.c_void_p(_stream_ptr()))
    check(rc, "syg_stft2048_c2c_f32")
    return out


def logmel_dct(mel: torch.Tensor, n_mfcc: Optional[int] = 13, dct_type: int = 2, norm="ortho", lifter: float = 0.0,
               amin: float = 1e-10, top_db: Optional[float] = 80.0, ref="max", keep_mel: bool = False):
    """power_to_db (ref = per-clip max or a scalar) then DCT.  Returns (logmel [B,M,T], mfcc [B,K,T] | None).

    `mel` is converted to dB in place unless keep_mel=True.
    """
    require_gpu()
    B, M, Tn = mel.shape
    if amin <= 0:
        raise ValueError("amin must be strictly positive")
    if top_db is not None and top_db < 0:
        raise ValueError("top_db must be non-negative")
    logmel = torch.empty_like(mel) if keep_mel else None
    mf = dct = lif = None
    K = 0
    if n_mfcc is not None:
        K = int(n_mfcc)
        dct = _cached(("dct", K, M, dct_type, norm), lambda: _dev(T.dct_matrix(K, M, dct_type, norm)))
        lw = T.lifter_weights(K, float(lifter))
        lif = _dev(lw) if lw is not None else None
        mf = torch.empty((B, K, Tn), dtype=torch.float32, device=mel.device)
    if isinstance(ref, str) and ref == "db":
        ref_is_max, ref_value = 2, 1.0          # input already in dB: DCT only
    elif (isinstance(ref, str) and ref == "max") or ref is np.max:
        ref_is_max, ref_value = 1, 1.0
    else:
        ref_is_max, ref_value = 0, float(ref)
    rc = lib().syg_logmel_dct_f32(_ptr(mel), B, M, Tn, _ptr(dct), K, _ptr(lif), float(amin),
                                  float(top_db) if top_db is not None else -1.0, ref_is_max, ref_value,
                                  _ptr(logmel), _ptr(mf), C.c_void_p(_stream_ptr()))
    check(rc, "syg_logmel_dct_f32")
    return (logmel if keep_mel else mel), mf


def mel_mfcc(mel: torch.Tensor, n_mfcc: int = 13, dct_type: int = 2, norm="ortho", lifter: float = 0.0, amin: float = 1e-10,
             top_db: Optional[float] = 80.0, ref="max") -> torch.Tensor:
    """MFCC [B, K, T] from a mel POWER matrix [B, M, T] that the caller no longer needs (it is scratch for the call): dB +
    DCT as logmel_dct, without writing the dB matrix to HBM."""
    require_gpu()
    B, M, Tn = mel.shape
    if amin <= 0:
        raise ValueError("amin must be strictly positive")
    if top_db is not None and top_db < 0:
        raise ValueError("top_db must be non-negative")
    K = int(n_mfcc)
    dct = _cached(("dct", K, M, dct_type, norm), lambda: _dev(T.dct_matrix(K, M, dct_type, norm)))
    lw = T.lifter_weights(K, float(lifter))
    lif = _dev(lw) if lw is not None else None
    mf = torch.empty((B, K, Tn), dtype=torch.float32, device=mel.device)
    if (isinstance(ref, str) and ref == "max") or ref is np.max:
        ref_is_max, ref_value = 1, 1.0
    else:
        ref_is_max, ref_value = 0, float(ref)
    rc = lib().syg_mel_mfcc_f32(_ptr(mel), B, M, Tn, _ptr(dct), K, _ptr(lif), float(amin),
                                float(top_db) if top_db is not None else -1.0, ref_is_max, ref_value, _ptr(mf),
                                C.c_void_p(_stream_ptr()))
    check(rc, "syg_mel_mfcc_f32")
    return mf


def mfcc_fused_fits(n_mels: int, n_frames: int, n_mfcc: int = 13) -> bool:
    """True when the clip's mel matrix + DCT rows fit the LDS left beside the fused kernel's buffers (the library
    owns the formula: syg_stft2048_mfcc_fits)."""
    return bool(lib().syg_stft2048_mfcc_fits(int(n_mels), int(n_frames), int(n_mfcc)))


def mfcc_fused_pays(n_mels: int, n_frames: int, n_mfcc: int = 13) -> bool:
    """True when mfcc_batch's default should take the one-launch form: the clip's mel matrix fits beside the stage buffer,
    or it fits in the buffer's place (frames then come straight from global memory) and is wide enough for the saved
    second launch to outweigh that (128 bands: 186 against 196 us per 1024 clips; 64 bands: 176 against 166)."""
    f = int(lib().syg_stft2048_mfcc_fits(int(n_mels), int(n_frames), int(n_mfcc)))
    return f == 2 or (f == 1 and n_mels >= 96)


class _MfccCall:
    """Prepared argument list of syg_stft2048_mfcc_f32 for one (shape, parameters) combination: a steady-state
    call then costs two allocations and one ctypes call (the kernel runs ~0.2 ms; rebuilding keys, tables and
    22 ctypes arguments per call costs the host about as much)."""

    def __init__(self, B, L, ld, device, sr, hop, center, window, n_mels, n_mfcc, fmin, fmax, lifter, amin, top_db,
                 ref, dct_type, norm, keep_mel, projection="auto"):
        if amin <= 0:
            raise ValueError("amin must be strictly positive")
        if top_db is not None and top_db < 0:
            raise ValueError("top_db must be non-negative")
        Tn = num_frames(L, 2048, hop, center)
        if Tn <= 0:
            raise ValueError("signal too short for one frame")
        self.cfg = mel_config(sr, 2048, n_mels, fmin, fmax, waves=16)
        K = int(n_mfcc)
        self.dct = _cached(("dct", K, n_mels, dct_type, norm), lambda: _dev(T.dct_matrix(K, n_mels, dct_type, norm)))
        lw = T.lifter_weights(K, float(lifter))
        self.lif = _dev(lw) if lw is not None else None
        if (isinstance(ref, str) and ref == "max") or ref is np.max:
            ref_is_max, ref_value = 1, 1.0
        else:
            ref_is_max, ref_value = 0, float(ref)
        self.win = window_dev(window, 2048, 2048)
        self.tw = twiddle_dev(2048)
        self.shape = (B, L, ld)
        self.device = device
        self.mel_shape = (B, n_mels, Tn) if keep_mel else None
        self.out_shape = (B, K, Tn)
        tail = (n_mels, _ptr(self.dct), K, _ptr(self.lif), float(amin), float(top_db) if top_db is not None else -1.0,
                ref_is_max, ref_value)
        # projection: "segments" = each wave projects its own power row by segment sums (no weight matrix, no workgroup
        # barrier in the projection: 133 against 149 us at config C2), "matrix" = block-sparse weights on the matrix
        # cores; "auto" takes the segment form when the filterbank has a piece table, the staged loads apply (hop <= 512),
        # no copy of the mel matrix is asked for and the two mel matrices fit the LDS
        if projection not in ("auto", "matrix", "segments"):
            raise ValueError("projection must be 'auto', 'matrix' or 'segments'")
        tri_ok = (self.cfg.segtab is not None and not keep_mel and hop <= 512
                  and bool(lib().syg_stft2048_mfcc_tri_fits(int(n_mels), int(Tn), K)))
        if projection == "segments" and not tri_ok:
            raise SygnalsHipError("stft2048_mfcc: no segment-sum projection for this call (filterbank without a piece table, "
                                  "keep_mel, hop > 512, or the clip's two mel matrices do not fit the LDS)")
        self.tri = tri_ok and projection != "matrix"
        if self.tri:
            self.fn, self.name = lib().syg_stft2048_mfcc_tri_f32, "syg_stft2048_mfcc_tri_f32"
            self.head = (B, L, ld, hop, int(center), Tn, _ptr(self.win), _ptr(self.tw), _ptr(self.cfg.segtab),
                         int(self.cfg.segtab.numel())) + tail
        else:
            self.fn, self.name = lib().syg_stft2048_mfcc_f32, "syg_stft2048_mfcc_f32"
            self.head = (B, L, ld, hop, int(center), Tn, _ptr(self.win), _ptr(self.tw), _ptr(self.cfg.wpacked),
                         self.cfg.plan.ctypes.data_as(C.c_void_p)) + tail

    def __call__(self, y):
        mel = torch.empty(self.mel_shape, dtype=torch.float32, device=self.device) if self.mel_shape else None
        mf = torch.empty(self.out_shape, dtype=torch.float32, device=self.device)
        if self.tri:
            rc = self.fn(y.data_ptr(), *self.head, mf.data_ptr(), _stream_ptr())
        else:
            rc = self.fn(y.data_ptr(), *self.head, mel.data_ptr() if mel is not None else None, mf.data_ptr(),
                         _stream_ptr())
        if rc:
            check(rc, self.name)
        return mf, mel


_mfcc_calls: dict = {}


def stft2048_mfcc(y: torch.Tensor, sr: float, hop: int = 512, center: bool = True, window="hann", n_mels: int = 128,
                  n_mfcc: int = 13, fmin: float = 0.0, fmax=None, lifter: float = 0.0, amin: float = 1e-10,
                  top_db: Optional[float] = 80.0, ref="max", dct_type: int = 2, norm="ortho", keep_mel: bool = False,
                  projection: str = "auto"):
    """One launch: [B, L] clips -> MFCC [B, n_mfcc, T]; the mel matrix of a clip never leaves LDS.
    projection: "auto" | "segments" | "matrix" -- how a power row becomes mel bands (see _MfccCall).

    Returns (mfcc, mel_power | None).  Raises SygnalsHipError when the clip's mel matrix does not fit
    (mfcc_fused_fits) -- mfcc_batch() falls back to the two-launch form by itself.
    """
    if y.dim() != 2 or y.dtype != torch.float32 or not y.is_cuda:
        require_gpu()
        raise ValueError("y must be a float32 [B, L] device tensor")
    if y.stride(1) != 1:
        y = y.contiguous()
    wkey = window if isinstance(window, str) else _window_key(window, 2048, 2048)
    rkey = "max" if (ref is np.max or ref == "max") else float(ref)
    key = (y.device.index, y.shape[0], y.shape[1], _ld(y), float(sr), hop, bool(center), wkey, n_mels, n_mfcc,
           float(fmin), fmax, float(lifter), amin, top_db, rkey, dct_type, norm, keep_mel, projection)
    call = _mfcc_calls.get(key)
    if call is None:
        require_gpu()
        call = _MfccCall(y.shape[0], y.shape[1], _ld(y), y.device, sr, hop, center, window, n_mels, n_mfcc, fmin,
                         fmax, lifter, amin, top_db, ref, dct_type, norm, keep_mel, projection)
        if len(_mfcc_calls) > 64:
            _mfcc_calls.clear()
        _mfcc_calls[key] = call
    return call(y)


def mfcc_batch(y: torch.Tensor, sr: float, n_fft: int = 2048, hop: int = 512, n_mels: int = 128, n_mfcc: int = 13,
               center: bool = True, window="hann", fmin: float = 0.0, fmax=None, lifter: float = 0.0, fused=None):
    """Config C2: [B, L] clips -> MFCC [B, n_mfcc, T] (manager path a1..a5), all on device.

    fused=None picks the one-launch clip-resident form when the clip's mel matrix fits in LDS and there are
    enough clips to fill the chip (a workgroup owns whole clips); True / False force either form.
    """
    if n_fft != 2048 and fused_pow2_ok(n_fft, n_mels) and fused is not False:
        # the other power-of-two frame lengths.  fused=True: ONE launch (a workgroup owns a clip, needs the clip's mel
        # matrix to fit the LDS).  Default: the tile kernel + logmel_dct -- two launches, but more workgroups per CU
        # (n_fft 1024: 266 us against 318 us per 1024 clips x 1 s; n_fft 512: 709 against 973 us)
        fits = mfcc_pow2_fits(n_fft, n_mels, num_frames(y.shape[1], n_fft, hop, center), n_mfcc)
        if fused and not fits:
            raise SygnalsHipError("mfcc_batch: the clip's mel matrix does not fit the LDS of the one-launch form")
        if fits and fused:
            return stft_mfcc_pow2(y, sr, n_fft, hop, center, window, None, n_mels, n_mfcc, fmin, fmax, lifter)[0]
        if n_fft == 1024 and w1024_segtab(sr, n_mels, fmin, fmax) is not None:
            # free-running waves, mel by segment sums: 0.162 against 0.235 ms per 1024 clips x 1 s for the mel launch
            mel = stft_mel_w1024_seg(y, sr, hop, center, window, None, n_mels, fmin, fmax)
        elif wsmall_segtab(sr, n_fft, n_mels, fmin, fmax) is not None:
            # the same for 512 / 256: 0.265 against 0.334 ms, 0.330 against 0.356 ms
            mel = stft_mel_wseg_small(y, sr, n_fft, hop, center, window, None, n_mels, fmin, fmax)
        else:
            mel = stft_mel_pow2(y, sr, n_fft, hop, center, window, None, n_mels, fmin, fmax)
        return mel_mfcc(mel, n_mfcc, lifter=lifter)
    if n_fft == 4096 and fused is not False and w4096_segtab(sr, n_mels, fmin, fmax) is not None:
        # frame length 4096: one launch samples -> mel (one wave per frame, mel by segment sums), then dB + DCT
        return mel_mfcc(stft_mel_w4096(y, sr, hop, center, window, None, n_mels, fmin, fmax), n_mfcc, lifter=lifter)
    if not fused_mel_ok(sr, n_fft, n_mels, fmin, fmax):
        # no fused kernel for this shape: complex STFT (any frame length) -> |X|^2 -> dense mel -> dB + DCT
        if fused:
            raise SygnalsHipError(f"mfcc_batch: no fused kernel for n_fft={n_fft}, n_mels={n_mels}")
        P = cabs_pow(stft_any(y, n_fft, hop, center, window), 2)
        mel = mel_dense(P, mel_config(sr, n_fft, n_mels, fmin, fmax).basis)
        return mel_mfcc(mel, n_mfcc, lifter=lifter)
    if fused is None:
        fused = mfcc_fused_pays(n_mels, num_frames(y.shape[1], 2048, hop, center), n_mfcc) and y.shape[0] >= 128
    if fused:
        return stft2048_mfcc(y, sr, hop, center, window, n_mels, n_mfcc, fmin, fmax, lifter)[0]
    mel, _, _ = stft2048_mel(y, sr, hop, center, window, 2048, n_mels, fmin, fmax)
    return mel_mfcc(mel, n_mfcc, lifter=lifter)


# ------------------------------------------------------------------ generic pow2 kernels
def is_pow2(n: int) -> bool:
    return n >= 2 and (n & (n - 1)) == 0


def twiddle_rfft_dev(n_fft) -> torch.Tensor:
    """[n_fft + n_fft/2, 2]: W_nfft^k followed by W_{nfft/2}^k (layout of stft_pow2 / welch)."""
    return _cached(("twr", n_fft), lambda: _dev(np.concatenate([T.twiddles(n_fft), T.twiddles(n_fft // 2)], axis=0)))


def fft_pow2(x: torch.Tensor, inverse: bool = False) -> torch.Tensor:
    """Batched complex FFT of x [batch, n, 2] float32, n a power of two <= 8192."""
    require_gpu()
    if x.dim() != 3 or x.shape[2] != 2 or x.dtype != torch.float32:
        raise ValueError("x must be float32 [batch, n, 2]")
    x = x.contiguous()
    batch, n, _ = x.shape
    out = torch.empty_like(x)
    rc = lib().syg_fft_pow2_c2c_f32(_ptr(x), _ptr(out), batch, n, int(inverse), _ptr(twiddle_dev(n)),
                                    C.c_void_p(_stream_ptr()))
    check(rc, "syg_fft_pow2_c2c_f32")
    return out


def stft_pow2(y: torch.Tensor, n_fft: int, hop: int, center: bool = True, window="hann", win_length=None):
    """Generic framed STFT, frame-major complex [B, T, 1 + n_fft/2, 2]."""
    require_gpu()
    if y.stride(1) != 1:
        y = y.contiguous()
    B, L = y.shape
    win_length = n_fft if win_length is None else win_length
    Tn = num_frames(L, n_fft, hop, center)
    if Tn <= 0:
        raise ValueError("signal too short for one frame")
    win = window_dev(window, win_length, n_fft)
    out = torch.empty((B, Tn, n_fft // 2 + 1, 2), dtype=torch.float32, device=y.device)
    rc = lib().syg_stft_pow2_c2c_f32(_ptr(y), B, L, _ld(y), n_fft, hop, int(center), Tn, _ptr(win),
                                     _ptr(twiddle_rfft_dev(n_fft)), _ptr(out), C.c_void_p(_stream_ptr()))
    check(rc, "syg_stft_pow2_c2c_f32")
    return out


def stft_any(y, n_fft, hop, center=True, window="hann", win_length=None):
    """Complex STFT [B, T, F, 2]: the wave-FFT kernel for n_fft = 2048, the LDS Stockham kernel otherwise."""
    if n_fft == 2048:
        return stft2048_c2c(y, hop, center, window, 2048 if win_length is None else win_length)
    if is_pow2(n_fft) and 8 <= n_fft <= 16384:
        return stft_pow2(y, n_fft, hop, center, window, win_length)
    return stft_rows(y, n_fft, hop, center, window, win_length)


MAX_ROWS = 65535


def pack_frames(y: torch.Tensor, n_rows: int, length: int, step: int, first: int, n_out: int,
                window: Optional[torch.Tensor] = None, detrend=False, cplx: bool = True) -> torch.Tensor:
    """Frames first .. first + n_rows - 1 (length `length`, start (first + r) * step) of EVERY row of the contiguous
    y [B, Lp] -> [B * n_rows, n_out(, 2)] detrended / windowed / zero-padded rows, clip-major: one launch for the whole
    batch (syg_pack_frames_f32; B * n_rows <= 65535)."""
    B, Lp = y.shape
    rows = B * n_rows
    out = torch.empty((rows, n_out, 2) if cplx else (rows, n_out), dtype=torch.float32, device=y.device)
    dcode = detrend_code(detrend)
    work = torch.empty(lib().syg_pack_rows_work_bytes(rows) // 8, dtype=torch.float64, device=y.device) if dcode else None
    base = y.data_ptr() + 4 * first * step
    rc = lib().syg_pack_frames_f32(C.c_void_p(base), rows, length, step, n_rows, Lp, _ptr(window), dcode, 0, int(cplx),
                                   _ptr(out), n_out, _ptr(work), C.c_void_p(_stream_ptr()))
    check(rc, "syg_pack_frames_f32")
    return out


def stft_rows(y: torch.Tensor, n_fft: int, hop: int, center: bool = True, window="hann", win_length=None):
    """Framed STFT for ANY n_fft >= 1 (librosa.stft accepts any frame length): the frames of each clip are windowed
    and packed as overlapping rows, transformed with the arbitrary-length FFT (four-step / Bluestein) and cut to the
    1 + n_fft//2 non-negative bins.  Frame-major complex [B, T, F, 2] like the power-of-two kernels."""
    require_gpu()
    B, L = y.shape
    win_length = n_fft if win_length is None else win_length
    pad = n_fft // 2 if center else 0
    # librosa's count on the padded signal; equals num_frames() for even n_fft, one less for an odd n_fft when hop
    # divides L (the padding is n_fft // 2 on both sides, one sample short of n_fft)
    Tn = 1 + (L + 2 * pad - n_fft) // hop if L + 2 * pad >= n_fft else 0
    if Tn <= 0:
        raise ValueError("signal too short for one frame")
    yp = y
    if pad or y.stride(1) != 1 or not y.is_contiguous():
        yp = torch.zeros((B, L + 2 * pad), dtype=torch.float32, device=y.device)     # zero padding: data movement only
        yp[:, pad:pad + L] = y
    win = window_dev(window, win_length, n_fft)
    F = n_fft // 2 + 1
    out = torch.empty((B, Tn, F, 2), dtype=torch.float32, device=y.device)
    yp = yp.contiguous()
    if Tn <= MAX_ROWS:                                        # whole clips per launch: as many as fit 65535 rows
        per = max(1, MAX_ROWS // Tn)
        for b0 in range(0, B, per):
            bc = min(per, B - b0)
            X = fft_any(pack_frames(yp[b0:b0 + bc], Tn, n_fft, hop, 0, n_fft, window=win))
            out[b0:b0 + bc] = X.view(bc, Tn, n_fft, 2)[:, :, :F]
        return out
    for b in range(B):                                        # long clips: 65535 frames of one clip at a time
        for t0 in range(0, Tn, MAX_ROWS):
            tc = min(MAX_ROWS, Tn - t0)
            X = fft_any(pack_frames(yp[b:b + 1], tc, n_fft, hop, t0, n_fft, window=win))
            out[b, t0:t0 + tc] = X[:, :F]
    return out


def cabs_pow(x: torch.Tensor, power: int = 1) -> torch.Tensor:
    """|x|^power (power 1 or 2) of an interleaved complex tensor [..., 2] -> [...]."""
    require_gpu()
    x = x.contiguous()
    out = torch.empty(x.shape[:-1], dtype=torch.float32, device=x.device)
    rc = lib().syg_cabs_pow_f32(_ptr(x), out.numel(), int(power), _ptr(out), C.c_void_p(_stream_ptr()))
    check(rc, "syg_cabs_pow_f32")
    return out


def mel_dense(P: torch.Tensor, basis: torch.Tensor) -> torch.Tensor:
    """mel [B, M, T] from frame-major power P [B, T, F] and a dense basis [M, F]."""
    require_gpu()
    P = P.contiguous()
    B, Tn, F = P.shape
    M = basis.shape[0]
    out = torch.empty((B, M, Tn), dtype=torch.float32, device=P.device)
    rc = lib().syg_mel_dense_f32(_ptr(P), B, Tn, F, _ptr(basis), M, _ptr(out), C.c_void_p(_stream_ptr()))
    check(rc, "syg_mel_dense_f32")
    return out


def spectral_stats(mag: torch.Tensor, freqs: torch.Tensor, roll_percent: float = 0.85, bw_p: float = 2.0):
    """Per-frame statistics of frame-major magnitudes mag [N, F]; returns [8, N] (SYG_STAT_* rows)."""
    require_gpu()
    if not 0.0 <= roll_percent <= 1.0:
        raise ValueError("roll_percent must be between 0.0 and 1.0.")
    if bw_p <= 0:
        raise ValueError("Order 'p' for spectral bandwidth must be positive.")
    mag = mag.contiguous()
    N, F = mag.shape
    out = torch.empty((8, N), dtype=torch.float32, device=mag.device)
    rc = lib().syg_spectral_stats_f32(_ptr(mag), N, F, _ptr(freqs), float(roll_percent), float(bw_p), _ptr(out),
                                      C.c_void_p(_stream_ptr()))
    check(rc, "syg_spectral_stats_f32")
    return out


FS_ROWS = ("mean_amplitude", "std_dev_amplitude", "skewness", "kurtosis", "peak_amplitude", "crest_factor",
           "signal_entropy", "rms_energy", "zero_crossing_rate")      # row r <-> mask bit r of syg_frame_stats_f32


def frame_stats(y: torch.Tensor, frame_length: int = 2048, hop: int = 512, center: bool = True, num_bins: int = 10,
                mask: int = 0x1FF) -> torch.Tensor:
    """Time-domain frame features of clips y [B, L]: [B, 9, T] float32, rows as FS_ROWS (only the rows
    selected by `mask` are written)."""
    require_gpu()
    if y.dim() != 2 or y.dtype != torch.float32 or not y.is_cuda:
        raise ValueError("y must be a float32 CUDA tensor of shape [B, L]")
    if y.stride(1) != 1:
        y = y.contiguous()
    B, L = y.shape
    Tn = num_frames(L, frame_length, hop, center)
    if Tn <= 0:
        raise ValueError("signal too short for one frame")
    out = torch.zeros((B, 9, Tn), dtype=torch.float32, device=y.device)
    rc = lib().syg_frame_stats_f32(_ptr(y), B, L, _ld(y), int(frame_length), int(hop), int(center), Tn, int(num_bins),
                                   int(mask), _ptr(out), C.c_void_p(_stream_ptr()))
    check(rc, "syg_frame_stats_f32")
    return out


def rms_from_spec(S: torch.Tensor, frame_length: int) -> torch.Tensor:
    """librosa.feature.rms(S=...) for frame-major magnitudes S [N, F] -> [N]."""
    require_gpu()
    S = S.contiguous()
    N, F = S.shape
    out = torch.empty((N,), dtype=torch.float32, device=S.device)
    rc = lib().syg_rms_from_spec_f32(_ptr(S), N, F, int(frame_length), _ptr(out), C.c_void_p(_stream_ptr()))
    check(rc, "syg_rms_from_spec_f32")
    return out


def contrast_pv(mag: torch.Tensor, cplan: np.ndarray) -> torch.Tensor:
    """Peak / valley tail means [2, R, N] of frame-major magnitudes mag [N, F]."""
    require_gpu()
    mag = mag.contiguous()
    N, F = mag.shape
    cplan = np.ascontiguousarray(cplan, dtype=np.int32)
    out = torch.empty((2, int(cplan[0]), N), dtype=torch.float32, device=mag.device)
    rc = lib().syg_contrast_pv_f32(_ptr(mag), N, F, cplan.ctypes.data_as(C.c_void_p), _ptr(out),
                                   C.c_void_p(_stream_ptr()))
    check(rc, "syg_contrast_pv_f32")
    return out


def sosfiltfilt(x: torch.Tensor, sos: np.ndarray, zi: np.ndarray, padlen: int) -> torch.Tensor:
    """Batched zero-phase SOS filtering of x [B, L] (scipy.signal.sosfiltfilt semantics)."""
    require_gpu()
    if x.stride(1) != 1:
        x = x.contiguous()
    B, L = x.shape
    sos = np.ascontiguousarray(sos, dtype=np.float64)
    zi = np.ascontiguousarray(zi, dtype=np.float64)
    S = sos.shape[0]
    if L <= padlen:
        raise ValueError(f"The length of the input vector x must be greater than padlen, which is {padlen}.")
    nbytes = lib().syg_sosfiltfilt_work_bytes(B, L, padlen, S)
    if nbytes < 0:
        raise SygnalsHipError(f"sosfiltfilt: unsupported configuration (sections={S}, max 8)")
    # (0 bytes: the clip-resident form keeps the clip in registers and needs no workspace)
    work = torch.empty((nbytes + 7) // 8, dtype=torch.float64, device=x.device) if nbytes > 0 else None
    y = torch.empty((B, L), dtype=torch.float32, device=x.device)
    rc = lib().syg_sosfiltfilt_f32(_ptr(x), B, L, _ld(x), sos.ctypes.data_as(C.c_void_p),
                                   zi.ctypes.data_as(C.c_void_p), S, int(padlen), _ptr(y), _ld(y),
                                   None if work is None else _ptr(work),
                                   C.c_void_p(_stream_ptr()))
    check(rc, "syg_sosfiltfilt_f32")
    return y


def detrend_code(detrend) -> int:
    """scipy's detrend argument -> the C ABI's code: 0 none (False / None / 'none'), 1 'constant' (True), 2 'linear'."""
    if detrend in (False, None, "none", 0):
        return 0
    if detrend in (True, "constant", 1):
        return 1
    if detrend in ("linear", 2):
        return 2
    raise ValueError("Trend type must be 'linear' or 'constant'.")     # scipy.signal.detrend's message


def welch(x: torch.Tensor, nperseg: int, noverlap: int, nfft: int, window_host: np.ndarray, detrend,
          scale: float) -> torch.Tensor:
    """Welch PSD [B, 1 + nfft/2] of x [B, L]; detrend as scipy.signal.welch (False / 'constant' / 'linear')."""
    require_gpu()
    if x.stride(1) != 1:
        x = x.contiguous()
    B, L = x.shape
    if not (is_pow2(nfft) and 8 <= nfft <= 16384):
        return welch_rows(x, nperseg, noverlap, nfft, window_host, detrend, scale)
    win = _dev(np.asarray(window_host, dtype=np.float32))
    nbytes = lib().syg_welch_work_bytes(B, nfft)
    work = torch.empty((nbytes + 3) // 4, dtype=torch.float32, device=x.device)
    out = torch.empty((B, nfft // 2 + 1), dtype=torch.float32, device=x.device)
    rc = lib().syg_welch_f32(_ptr(x), B, L, _ld(x), nperseg, nperseg - noverlap, nfft, _ptr(win),
                             _ptr(twiddle_rfft_dev(nfft)), detrend_code(detrend), float(scale), _ptr(out), _ptr(work),
                             C.c_void_p(_stream_ptr()))
    check(rc, "syg_welch_f32")
    return out


def welch_rows(x: torch.Tensor, nperseg: int, noverlap: int, nfft: int, window_host: np.ndarray, detrend,
               scale: float) -> torch.Tensor:
    """Welch for ANY nperseg / nfft (scipy.signal.welch takes any): segments as overlapping rows -> detrend + window
    (syg_pack_rows_f32) -> arbitrary-length FFT -> one-sided |X|^2 -> float64 average in segment order."""
    B, L = x.shape
    step = nperseg - noverlap
    nseg = (L - noverlap) // step
    if nseg < 1:
        raise ValueError("welch: the signal is shorter than one segment")
    win = _dev(np.asarray(window_host, dtype=np.float32))
    F = nfft // 2 + 1
    out = torch.empty((B, F), dtype=torch.float32, device=x.device)
    acc = torch.empty(F, dtype=torch.float64, device=x.device)
    st = C.c_void_p(_stream_ptr())
    x = x.contiguous()
    for b in range(B):
        for s0 in range(0, nseg, MAX_ROWS):
            sc = min(MAX_ROWS, nseg - s0)
            X = fft_any(pack_frames(x[b:b + 1], sc, nperseg, step, s0, nfft, window=win, detrend=detrend))
            P = torch.empty((sc, F), dtype=torch.float32, device=x.device)
            check(lib().syg_psd_onesided_f32(_ptr(X), sc, nfft, float(scale), _ptr(P), st), "syg_psd_onesided_f32")
            check(lib().syg_col_mean_f32(_ptr(P), sc, F, _ptr(acc), int(s0 == 0), int(s0 + sc == nseg), float(nseg),
                                         _ptr(out[b]), st), "syg_col_mean_f32")
    return out


def contrast_db(pv: torch.Tensor, amin: float = 1e-10, top_db: Optional[float] = 80.0, linear: bool = False) -> torch.Tensor:
    """pv [B, 2, R, T] (peak, valley means) -> spectral contrast [B, R, T]: dB difference, or the plain difference
    of the means with linear=True (librosa's `linear` option)."""
    if linear:
        amin = 0.0
    require_gpu()
    pv = pv.contiguous()
    B, two, R, Tn = pv.shape
    out = torch.empty((B, R, Tn), dtype=torch.float32, device=pv.device)
    rc = lib().syg_contrast_db_f32(_ptr(pv), B, R, Tn, float(amin), float(top_db) if top_db is not None else -1.0,
                                   _ptr(out), C.c_void_p(_stream_ptr()))
    check(rc, "syg_contrast_db_f32")
    return out


# ------------------------------------------------------------------ arbitrary-length FFT
MAX_LDS_FFT = 8192


def _fft_strided(x, out, outer, batch, n, inverse, strides, bign=0, scale=1.0):
    in_os, in_bs, in_es, out_os, out_bs, out_es = strides
    rc = lib().syg_fft_pow2_strided_c2c_f32(_ptr(x), _ptr(out), outer, batch, n, int(inverse), _ptr(twiddle_dev(n)),
                                            in_os, in_bs, in_es, out_os, out_bs, out_es, bign, float(scale),
                                            C.c_void_p(_stream_ptr()))
    check(rc, "syg_fft_pow2_strided_c2c_f32")


def fft_pow2_any(x: torch.Tensor, inverse: bool = False) -> torch.Tensor:
    """Complex FFT of rows of x [rows, n, 2], n any power of two up to 2^26 (four-step above 8192)."""
    rows, n, _ = x.shape
    if n <= MAX_LDS_FFT:
        return fft_pow2(x, inverse)
    lg = n.bit_length() - 1
    n1 = 1 << (lg // 2)
    n2 = n // n1
    if n2 > MAX_LDS_FFT:
        raise SygnalsHipError(f"FFT length {n} exceeds the supported maximum 2^26")
    if rows > 65535:
        raise SygnalsHipError("too many rows for the four-step FFT")
    x = x.contiguous()
    tmp = torch.empty_like(x)
    out = torch.empty_like(x)
    # step A: N2 transforms of length N1 over n1 (input stride N2), twiddle W_N^(n2*k1), stored [n2][k1]
    _fft_strided(x, tmp, rows, n2, n1, inverse, (n, 1, n2, n, n1, 1), bign=n)
    # step B: N1 transforms of length N2 over n2 (input stride N1), output X[k1 + N1*k2]
    _fft_strided(tmp, out, rows, n1, n2, inverse, (n, 1, n1, n, 1, n1), scale=(1.0 / n if inverse else 1.0))
    return out


def cmul(a: torch.Tensor, b: torch.Tensor, conj_b: bool = False, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """a [.., na, 2] * b [nb, 2] broadcast over i mod nb."""
    a = a.contiguous(); b = b.contiguous()
    out = torch.empty_like(a) if out is None else out
    rc = lib().syg_cmul_c64(_ptr(a), _ptr(b), _ptr(out), a.numel() // 2, b.numel() // 2, int(conj_b),
                            C.c_void_p(_stream_ptr()))
    check(rc, "syg_cmul_c64")
    return out


def pack_real(x: torch.Tensor, n: int, window: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Real rows [rows, len] -> complex rows [rows, n, 2], windowed, zero-padded / truncated to n."""
    if x.stride(1) != 1:
        x = x.contiguous()
    rows, ln = x.shape
    out = torch.empty((rows, n, 2), dtype=torch.float32, device=x.device)
    rc = lib().syg_pack_real_c64(_ptr(x), rows, ln, _ld(x), _ptr(window), _ptr(out), n,
                                 C.c_void_p(_stream_ptr()))
    check(rc, "syg_pack_real_c64")
    return out


def _bluestein_tables(n: int):
    """Host float64 tables for the chirp-z form of an n-point DFT, forward and inverse.

    w[k] = exp(-i*pi*k^2/n); forward: X = w * IFFT_M(FFT_M(x*w) * FFT_M(wrap(conj w)));
    the inverse uses the conjugate chirp and folds the 1/n into the final multiply.
    """
    m = max(2, 2 * n - 1)                      # any length the FFT kernels take directly will do for the chirp
    while not (is_pow2(m) or (_is_smooth(m) and smooth_split(m) is not None)):   # convolution: the next 7-smooth one
        m += 1
    k = np.arange(n, dtype=np.int64)
    w = np.exp(-1j * np.pi * ((k * k) % (2 * n)).astype(np.float64) / n)

    def wrapped_fft(c):
        b = np.zeros(m, dtype=np.complex128)
        b[:n] = c
        b[m - n + 1:] = c[1:][::-1]
        return np.fft.fft(b)

    as2 = lambda z: _dev(np.stack([z.real, z.imag], axis=-1).astype(np.float32))
    fwd = (as2(w), as2(wrapped_fft(np.conj(w))), as2(w))
    inv = (as2(np.conj(w)), as2(wrapped_fft(w)), as2(np.conj(w) / n))
    return m, fwd, inv


MAX_MIXED_FFT = 8192


def smooth_split(n: int):
    """None when n has a prime factor other than 2, 3, 5, 7 (or is too long); (n, 1) when one mixed-radix launch
    takes it; else the most balanced (n1, n2), n1 * n2 = n, both <= 8192 -- the four-step factors."""
    if n < 2 or not _is_smooth(n):
        return None
    if n <= MAX_MIXED_FFT:
        return n, 1
    best = None
    d = 1
    while d * d <= n:
        if n % d == 0 and n // d <= MAX_MIXED_FFT:
            best = (d, n // d)                       # d <= sqrt(n): the largest such d is the most balanced split
        d += 1
    return best


def _is_smooth(n: int) -> bool:
    for p in (2, 3, 5, 7):
        while n % p == 0:
            n //= p
    return n == 1


def _fft_mixed_strided(x, out, outer, batch, n, inverse, strides, bign=0, scale=1.0):
    in_os, in_bs, in_es, out_os, out_bs, out_es = strides
    rc = lib().syg_fft_mixed_strided_c2c_f32(_ptr(x), _ptr(out), outer, batch, n, int(inverse), _ptr(twiddle_dev(n)),
                                             in_os, in_bs, in_es, out_os, out_bs, out_es, bign, float(scale),
                                             C.c_void_p(_stream_ptr()))
    check(rc, "syg_fft_mixed_strided_c2c_f32")


def fft_smooth(x: torch.Tensor, inverse: bool = False) -> torch.Tensor:
    """Complex FFT of rows of x [rows, n, 2] for n = 2^a 3^b 5^c 7^d: one mixed-radix launch up to 8192 points,
    four-step (two passes of mixed-radix transforms) above."""
    rows, n, _ = x.shape
    split = smooth_split(n)
    if split is None:
        raise SygnalsHipError(f"fft_smooth: n = {n} is not a product of 2, 3, 5, 7 that splits into factors <= 8192")
    x = x.contiguous()
    out = torch.empty_like(x)
    n1, n2 = split
    if n2 == 1:
        _fft_mixed_strided(x, out, 1, rows, n, inverse, (0, n, 1, 0, n, 1), scale=(1.0 / n if inverse else 1.0))
        return out
    if rows > MAX_ROWS:
        raise SygnalsHipError("too many rows for the four-step FFT")
    tmp = torch.empty_like(x)
    # step A: n2 transforms of length n1 over the slow index (stride n2), twiddle W_n^(i2 k1), stored [i2][k1];
    # step B: n1 transforms of length n2 over i2 (stride n1), output X[k1 + n1 k2]
    _fft_mixed_strided(x, tmp, rows, n2, n1, inverse, (n, 1, n2, n, n1, 1), bign=n)
    _fft_mixed_strided(tmp, out, rows, n1, n2, inverse, (n, 1, n1, n, 1, n1), scale=(1.0 / n if inverse else 1.0))
    return out


def fft_any(x: torch.Tensor, inverse: bool = False) -> torch.Tensor:
    """Complex FFT / IFFT of rows of x [rows, n, 2] for ANY n >= 1: LDS / four-step kernels for powers of two, the
    mixed-radix kernel for other products of 2, 3, 5, 7 (48000, 44100, 16000 ...), Bluestein for the rest."""
    require_gpu()
    rows, n, _ = x.shape
    if n == 1:
        return x.clone()
    if rows > MAX_ROWS and n > MAX_LDS_FFT:
        # the four-step passes put the rows on a grid axis of at most 65535: long transforms of very many rows go
        # through in blocks (without this a Bluestein length above 8192 would recurse on the same row count)
        out = torch.empty_like(x)
        for r0 in range(0, rows, MAX_ROWS):
            out[r0:r0 + MAX_ROWS] = fft_any(x[r0:r0 + MAX_ROWS], inverse)
        return out
    if is_pow2(n):
        return fft_pow2_any(x, inverse)
    if _is_smooth(n) and smooth_split(n) is not None:
        return fft_smooth(x, inverse)
    m, fwd, inv = _cached(("blue", n), lambda: _bluestein_tables(n))
    w_in, bf, w_out = inv if inverse else fwd
    a = torch.zeros((rows, m, 2), dtype=torch.float32, device=x.device)
    a[:, :n].copy_(cmul(x, w_in))                 # zero-padded copy (data movement)
    A = fft_any(a, False)
    cmul(A, bf, out=A)
    c = fft_any(A, True)
    return cmul(c[:, :n].contiguous(), w_out)


# ------------------------------------------------------------------ feature formatting for ML
def _mat32(x: torch.Tensor) -> torch.Tensor:
    if x.dim() != 2 or x.dtype != torch.float32 or not x.is_cuda:
        require_gpu()
        raise ValueError("x must be a float32 [n, F] device tensor")
    return x.contiguous()


def col_stats(x: torch.Tensor) -> torch.Tensor:
    """[5, F] float64: count of non-NaN, mean, population variance, min, max of every column of x [n, F]."""
    x = _mat32(x)
    n, F = x.shape
    out = torch.empty((5, F), dtype=torch.float64, device=x.device)
    check(lib().syg_col_stats_f32(_ptr(x), n, F, _ptr(out), C.c_void_p(_stream_ptr())), "syg_col_stats_f32")
    return out


def affine_cols(x: torch.Tensor, sub, mul, add) -> torch.Tensor:
    """(x - sub[c]) * mul[c] + add[c] per column, float64 arithmetic; sub / mul / add: length-F float64 (host or device)."""
    x = _mat32(x)
    n, F = x.shape
    vecs = [torch.as_tensor(np.asarray(v.cpu() if isinstance(v, torch.Tensor) else v, dtype=np.float64)).to(x.device)
            for v in (sub, mul, add)]
    if any(v.shape != (F,) for v in vecs):
        raise ValueError(f"sub / mul / add must have length {F}")
    out = torch.empty_like(x)
    check(lib().syg_affine_cols_f32(_ptr(x), n, F, _ptr(vecs[0]), _ptr(vecs[1]), _ptr(vecs[2]), _ptr(out),
                                    C.c_void_p(_stream_ptr())), "syg_affine_cols_f32")
    return out


def col_quantiles(x: torch.Tensor, q) -> torch.Tensor:
    """np.nanpercentile(x, 100 * q, axis=0) -> [len(q), F] float64 (q: fractions in [0, 1]); at most 32768 rows."""
    x = _mat32(x)
    n, F = x.shape
    qv = np.atleast_1d(np.asarray(q, dtype=np.float64))
    if qv.ndim != 1 or qv.size < 1 or (qv < 0).any() or (qv > 1).any():
        raise ValueError("q must be fractions in [0, 1]")
    qd = torch.from_numpy(qv).to(x.device)
    out = torch.empty((qv.size, F), dtype=torch.float64, device=x.device)
    check(lib().syg_col_quantiles_f32(_ptr(x), n, F, _ptr(qd), int(qv.size), _ptr(out), C.c_void_p(_stream_ptr())),
          "syg_col_quantiles_f32")
    return out


def zoom2d(img: torch.Tensor, out_shape, order: int = 1) -> torch.Tensor:
    """scipy.ndimage.zoom(img, out_shape / img.shape, order=0|1, mode='nearest') of a [H, W] float32 device tensor."""
    img = _mat32(img)
    H, W = img.shape
    H2, W2 = int(out_shape[0]), int(out_shape[1])
    out = torch.empty((H2, W2), dtype=torch.float32, device=img.device)
    check(lib().syg_zoom_f32(_ptr(img), H, W, H2, W2, int(order), _ptr(out), C.c_void_p(_stream_ptr())), "syg_zoom_f32")
    return out


# ------------------------------------------------------------------ batched ingest
_PCM_BITS = {torch.int16: 16, torch.int32: 32, torch.uint8: 8}


def pcm_to_f32(pcm: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Integer PCM [B, L] or [B, L, C] (int16 / int32 / uint8, interleaved channels, on the device) -> float32 mono
    clips [B, L]: samples / 2^(bits-1), channels averaged (librosa.load(mono=True) semantics)."""
    require_gpu()
    if pcm.dtype not in _PCM_BITS or pcm.dim() not in (2, 3) or not pcm.is_cuda:
        raise ValueError("pcm must be an int16 / int32 / uint8 device tensor of shape [B, L] or [B, L, C]")
    pcm = pcm.contiguous()
    B, L = pcm.shape[0], pcm.shape[1]
    Cn = pcm.shape[2] if pcm.dim() == 3 else 1
    if out is None:
        out = torch.empty((B, L), dtype=torch.float32, device=pcm.device)
    elif out.shape != (B, L) or out.dtype != torch.float32 or out.stride(1) != 1:
        raise ValueError("out must be a float32 [B, L] tensor with unit inner stride")
    rc = lib().syg_pcm_to_f32(_ptr(pcm), _PCM_BITS[pcm.dtype], B, L, Cn, L * Cn, _ptr(out), _ld(out),
                              C.c_void_p(_stream_ptr()))
    check(rc, "syg_pcm_to_f32")
    return out


# ------------------------------------------------------------------ FFT-backed 1-D operations
def pack_rows(x: torch.Tensor, n: int, window: Optional[torch.Tensor] = None, detrend=False,
              reverse: bool = False, cplx: bool = False) -> torch.Tensor:
    """Rows of x [rows, len] -> [rows, n] real (or [rows, n, 2] complex) rows: mean ('constant') or least-squares
    line ('linear') removed, windowed, optionally time-reversed, zero-padded / truncated to n."""
    require_gpu()
    if x.dim() != 2 or x.dtype != torch.float32 or not x.is_cuda:
        raise ValueError("x must be a float32 [rows, len] device tensor")
    if x.stride(1) != 1:
        x = x.contiguous()
    rows, ln = x.shape
    out = torch.empty((rows, n, 2) if cplx else (rows, n), dtype=torch.float32, device=x.device)
    work = None
    dcode = detrend_code(detrend)
    if dcode:
        work = torch.empty(lib().syg_pack_rows_work_bytes(rows) // 8, dtype=torch.float64, device=x.device)
    rc = lib().syg_pack_rows_f32(_ptr(x), rows, ln, _ld(x), _ptr(window), dcode, int(bool(reverse)),
                                 int(bool(cplx)), _ptr(out), n, _ptr(work), C.c_void_p(_stream_ptr()))
    check(rc, "syg_pack_rows_f32")
    return out


def conv_fft_len(n_out: int) -> int:
    """Transform length (in real samples, even) for a linear convolution with n_out output samples: the smallest
    M >= n_out, M >= 16, whose half M/2 is a product of 2, 3, 5, 7 that the FFT kernels take directly -- what
    scipy.fft.next_fast_len does for fftconvolve.  (A power of two can be up to twice the needed length.)"""
    m = max(16, n_out + (n_out & 1))
    while True:
        h = m // 2
        if is_pow2(h) or (_is_smooth(h) and smooth_split(h) is not None):
            return m
        m += 2


def rfft_conv(x: torch.Tensor, k: torch.Tensor, reverse_k: bool = False) -> torch.Tensor:
    """Full linear convolution of the rows of x [B, n] with k [1 or B, m] -> [B, n + m - 1] (a view of the
    transform buffer).  reverse_k convolves with the time-reversed k, i.e. cross-correlates."""
    B, n = x.shape
    Bk, m = k.shape
    if Bk not in (1, B):
        raise ValueError("k must have one row or one row per row of x")
    M = conv_fft_len(n + m - 1)
    H = M // 2
    za = fft_pair_rows(x, H)
    if za is None:
        za = fft_any(pack_rows(x, M).view(B, H, 2))
    zb = fft_any(pack_rows(k, M, reverse=reverse_k).view(Bk, H, 2))
    rc = lib().syg_rconv_spectrum_c64(_ptr(za), _ptr(zb), B, Bk, H, _ptr(za), C.c_void_p(_stream_ptr()))
    check(rc, "syg_rconv_spectrum_c64")
    return fft_any(za, True).view(B, M)[:, : n + m - 1]


FFT_REAL_IN, FFT_ABS_OUT, FFT_PAIR_IN = 1, 2, 4


def _two_kernel_plan(n: int):
    """(kind, n1, n2) of the one- or two-launch plan of a length-n complex transform, or None (Bluestein lengths)."""
    if n < 2:
        return None
    if is_pow2(n):
        if n <= MAX_LDS_FFT:
            return "pow2", n, 1
        n1 = 1 << ((n.bit_length() - 1) // 2)
        return ("pow2", n1, n // n1) if n // n1 <= MAX_LDS_FFT else None
    if _is_smooth(n) and smooth_split(n) is not None:
        n1, n2 = smooth_split(n)
        return "mixed", n1, n2
    return None


def fft_pair_rows(x: torch.Tensor, H: int):
    """Forward transform of length H of the rows of the REAL tensor x [B, n] read as the complex sequences
    (x[2 p], x[2 p + 1]), zero beyond n -- pack_rows(x, 2 H) folded into the first pass's load.  None without a plan."""
    B, n = x.shape
    plan = _two_kernel_plan(H)
    if plan is None or B > MAX_ROWS or x.dtype != torch.float32 or not x.is_cuda or n > 2 * H:
        return None
    if x.stride(1) != 1:
        x = x.contiguous()
    kind, n1, n2 = plan
    ld = x.stride(0)
    out = torch.empty((B, H, 2), dtype=torch.float32, device=x.device)
    if n2 == 1:
        _strided_ex(kind, x, out, B, 1, H, False, (ld, 0, 1, H, 0, 1), flags=FFT_PAIR_IN, in_valid=n)
        return out
    tmp = torch.empty_like(out)
    _strided_ex(kind, x, tmp, B, n2, n1, False, (ld, 1, n2, H, n1, 1), bign=H, flags=FFT_PAIR_IN, in_valid=n)
    _strided_ex(kind, tmp, out, B, n1, n2, False, (H, 1, n1, H, 1, n1))
    return out


def _strided_ex(kind, x, out, outer, batch, n, inverse, strides, bign=0, scale=1.0, flags=0, mask_n=0, in_valid=0):
    in_os, in_bs, in_es, out_os, out_bs, out_es = strides
    fn = lib().syg_fft_pow2_strided_ex_f32 if kind == "pow2" else lib().syg_fft_mixed_strided_ex_f32
    rc = fn(_ptr(x), _ptr(out), outer, batch, n, int(inverse), _ptr(twiddle_dev(n)), in_os, in_bs, in_es, out_os, out_bs,
            out_es, bign, float(scale), int(flags), int(mask_n), int(in_valid), C.c_void_p(_stream_ptr()))
    check(rc, "syg_fft_%s_strided_ex_f32" % kind)


def analytic_fused(x: torch.Tensor, magnitude: bool):
    """scipy.signal.hilbert of the rows of x [B, n] (float32, contiguous) with the packing, masking and |.| passes folded
    into the transforms' loads and stores: complex [B, n, 2], or the envelope [B, n] when magnitude.  Returns None where
    the length has no two-kernel plan (powers of two up to 2^26 and 7-smooth lengths that split into two factors <= 8192;
    other lengths go through Bluestein in analytic_signal)."""
    B, n = x.shape
    plan = _two_kernel_plan(n)
    if plan is None or B > MAX_ROWS:
        return None
    kind, n1, n2 = plan
    x = x.contiguous()
    X = torch.empty((B, n, 2), dtype=torch.float32, device=x.device)
    out = torch.empty((B, n) if magnitude else (B, n, 2), dtype=torch.float32, device=x.device)
    oflag = FFT_ABS_OUT if magnitude else 0
    if n2 == 1:
        # (one transform per row: the ROW is the outer index, so that an element's position inside the row is its bin)
        _strided_ex(kind, x, X, B, 1, n, False, (n, 0, 1, n, 0, 1), flags=FFT_REAL_IN)
        _strided_ex(kind, X, out, B, 1, n, True, (n, 0, 1, n, 0, 1), scale=1.0 / n, flags=oflag, mask_n=n)
        return out
    tmp = torch.empty_like(X)
    # forward: step A reads the REAL rows; inverse: step A weights the spectrum as it loads it, step B stores |.|
    _strided_ex(kind, x, tmp, B, n2, n1, False, (n, 1, n2, n, n1, 1), bign=n, flags=FFT_REAL_IN)
    _strided_ex(kind, tmp, X, B, n1, n2, False, (n, 1, n1, n, 1, n1))
    _strided_ex(kind, X, tmp, B, n2, n1, True, (n, 1, n2, n, n1, 1), bign=n, mask_n=n)
    _strided_ex(kind, tmp, out, B, n1, n2, True, (n, 1, n1, n, 1, n1), scale=1.0 / n, flags=oflag)
    return out


def analytic_signal(x: torch.Tensor) -> torch.Tensor:
    """scipy.signal.hilbert of the rows of x [B, n] -> complex [B, n, 2] (exact length-n transforms)."""
    B, n = x.shape
    if x.dtype == torch.float32 and x.is_cuda:
        fused = analytic_fused(x, False)
        if fused is not None:
            return fused
    X = fft_any(pack_rows(x, n, cplx=True))
    rc = lib().syg_analytic_mask_c64(_ptr(X), B, n, C.c_void_p(_stream_ptr()))
    check(rc, "syg_analytic_mask_c64")
    return fft_any(X, True)


def periodogram(x: torch.Tensor, nfft: int, window_host: Optional[np.ndarray], detrend, scale: float
                ) -> torch.Tensor:
    """One-sided periodogram [B, nfft//2 + 1] of the first min(len, nfft) samples of the rows of x."""
    B = x.shape[0]
    win = None if window_host is None else _dev(np.asarray(window_host, dtype=np.float32))
    X = fft_any(pack_rows(x, nfft, window=win, detrend=detrend, cplx=True))
    out = torch.empty((B, nfft // 2 + 1), dtype=torch.float32, device=x.device)
    rc = lib().syg_psd_onesided_f32(_ptr(X), B, nfft, float(scale), _ptr(out), C.c_void_p(_stream_ptr()))
    check(rc, "syg_psd_onesided_f32")
    return out


# ------------------------------------------------------------------ constant-Q transform
_side_streams: dict = {}


def _side_stream(device) -> "torch.cuda.Stream":
    """One extra HIP stream per device for work that runs beside the caller's stream (CQT octave products)."""
    k = torch.device(device).index if torch.device(device).index is not None else torch.cuda.current_device()
    st = _side_streams.get(k)
    if st is None:
        st = torch.cuda.Stream(device=k)
        _side_streams[k] = st
    return st


def cqt_pack_gemm(basis: np.ndarray, n_fft: int) -> np.ndarray:
    """A operands of syg_cqt_octave_gemm_f32 for one octave: the frequency-domain rows basis [n_filt, 1 + n_fft/2]
    (complex) act on rfft(frame); the same linear map in the time domain is out[f] = sum_n frame[n] g_f[n] with
    g_f[n] = sum_k basis[f, k] exp(-2 pi i k n / n_fft) (float64 here).  Rows 2f / 2f + 1 of G^T hold Re / Im of g_f;
    packed [row tile][n_fft/16][4][64] float32 with entry (mt, s, u, lane) = G[16 s + 4 (lane >> 4) + u][16 mt + lane % 16]."""
    basis = np.asarray(basis, dtype=np.complex128)
    nf, F = basis.shape
    k = np.arange(F)[:, None]
    n = np.arange(n_fft)[None, :]
    g = basis @ np.exp(-2j * np.pi * ((k * n) % n_fft) / n_fft)            # [nf, n_fft]
    ntile = (2 * nf + 15) // 16
    G = np.zeros((n_fft, 16 * ntile), dtype=np.float64)
    G[:, 0:2 * nf:2] = g.real.T
    G[:, 1:2 * nf:2] = g.imag.T
    lane = np.arange(64)
    S = n_fft // 16
    out = np.empty((ntile, S, 4, 64), dtype=np.float32)
    for mt in range(ntile):
        for s in range(S):
            for u in range(4):
                out[mt, s, u] = G[16 * s + 4 * (lane >> 4) + u, 16 * mt + (lane & 15)]
    return np.ascontiguousarray(out)


def _bf16_rne(x32: np.ndarray) -> np.ndarray:
    """float32 -> bfloat16 bit patterns (uint16), round to nearest even (what v_cvt_pk_bf16_f32 does)."""
    u = np.ascontiguousarray(x32, dtype=np.float32).view(np.uint32).astype(np.uint64)
    return ((u + 0x7FFF + ((u >> 16) & 1)) >> 16).astype(np.uint16)


def cqt_pack_bf16x3(basis: np.ndarray, n_fft: int) -> np.ndarray:
    """Operand table of syg_cqt_octave_bf16x3_f32: float32(G) (cqt_pack_gemm's matrix) split into three bfloat16 terms
    hi + mid + lo (hi = bf16(g), mid = bf16(g - hi), lo = bf16(g - hi - mid); the two differences are exact in
    float32), packed uint16 [3][row tile][n_fft / 32][64 lanes][8]: entry (p, mt, s, lane, j) = term p of
    G[32 s + 8 (lane >> 4) + j][16 mt + (lane & 15)]."""
    basis = np.asarray(basis, dtype=np.complex128)
    nf, F = basis.shape
    k = np.arange(F)[:, None]
    n = np.arange(n_fft)[None, :]
    g = basis @ np.exp(-2j * np.pi * ((k * n) % n_fft) / n_fft)
    ntile = (2 * nf + 15) // 16
    G = np.zeros((n_fft, 16 * ntile), dtype=np.float32)
    G[:, 0:2 * nf:2] = g.real.T
    G[:, 1:2 * nf:2] = g.imag.T

    def f32_of(b16):
        return (b16.astype(np.uint32) << 16).view(np.float32)
    hi = _bf16_rne(G)
    r1 = G - f32_of(hi)
    mid = _bf16_rne(r1)
    lo = _bf16_rne(r1 - f32_of(mid))
    lane = np.arange(64)
    S = n_fft // 32
    out = np.empty((3, ntile, S, 64, 8), dtype=np.uint16)
    for p, term in enumerate((hi, mid, lo)):
        for mt in range(ntile):
            for s_ in range(S):
                for j in range(8):
                    out[p, mt, s_, :, j] = term[32 * s_ + 8 * (lane >> 4) + j, 16 * mt + (lane & 15)]
    return np.ascontiguousarray(out)


def decimate2(x: torch.Tensor, taps: torch.Tensor, scale: float) -> torch.Tensor:
    """FIR decimation by two of x [B, L] -> [B, ceil(L/2)] (zero padded ends)."""
    require_gpu()
    if x.stride(1) != 1:
        x = x.contiguous()
    B, L = x.shape
    y = torch.empty((B, (L + 1) // 2), dtype=torch.float32, device=x.device)
    rc = lib().syg_decimate2_f32(_ptr(x), B, L, _ld(x), _ptr(taps), taps.numel(), float(scale), _ptr(y), _ld(y),
                                 C.c_void_p(_stream_ptr()))
    check(rc, "syg_decimate2_f32")
    return y


def decimate2_chain(x: torch.Tensor, taps: torch.Tensor, scale: float, levels: int, keep=None) -> list:
    """`levels` successive decimate2 steps; up to 4 at a time go through one pass of syg_decimate2_chain_f32 (identical
    bits, every level written once).  keep[s] False: level s is not wanted (returned as None; never the last one)."""
    require_gpu()
    if x.stride(1) != 1:
        x = x.contiguous()
    keep = [True] * levels if keep is None else list(keep)
    keep[-1] = True
    out = []
    cur = x
    s = 0
    while s < levels:
        n = levels - s if levels - s <= 4 else 3        # three levels per pass; a remainder of four goes in one
        B, L = cur.shape
        ys, lens = [], L
        for j in range(n):
            lens = (lens + 1) // 2
            # (a level that feeds the next pass is needed even when the caller does not want it)
            need = keep[s + j] or j == n - 1
            ys.append(torch.empty((B, lens), dtype=torch.float32, device=x.device) if need else None)
        yp = (C.c_void_p * n)(*[(_ptr(t) if t is not None else None) for t in ys])
        ld = (C.c_int64 * n)(*[(_ld(t) if t is not None else 0) for t in ys])
        rc = lib().syg_decimate2_chain_f32(_ptr(cur), B, L, _ld(cur), _ptr(taps), taps.numel(), float(scale), n, yp, ld,
                                           C.c_void_p(_stream_ptr()))
        check(rc, "syg_decimate2_chain_f32")
        out += [t if keep[s + j] else None for j, t in enumerate(ys)]
        cur = ys[-1]
        s += n
    return out


def cqt(y: torch.Tensor, sr: float, hop_length: int = 512, fmin=None, n_bins: int = 84, bins_per_octave: int = 12,
        tuning: float = 0.0, filter_scale: float = 1.0, sparsity: float = 0.01) -> torch.Tensor:
    """Constant-Q transform of y [B, L] -> complex [B, n_bins, T, 2] float32, T = 1 + L // hop_length."""
    from ._cqt import CqtPlan, decimation_taps
    require_gpu()
    key = ("cqt", float(sr), int(hop_length), None if fmin is None else float(fmin), int(n_bins), int(bins_per_octave),
           float(tuning), float(filter_scale), float(sparsity))

    def build():
        p = CqtPlan(sr, hop_length, fmin, n_bins, bins_per_octave, tuning, filter_scale, sparsity)
        for o in p.octaves:
            b = o["basis"]
            b32 = np.stack([b.real, b.imag], axis=-1).astype(np.float32)
            o["basis_dev"] = _dev(b32)
            nz = (b32[..., 0] != 0) | (b32[..., 1] != 0)          # librosa's sparsified basis: a short run per row
            k0 = np.array([int(np.argmax(r)) if r.any() else 0 for r in nz], dtype=np.int32)
            k1 = np.array([int(len(r) - np.argmax(r[::-1])) if r.any() else 0 for r in nz], dtype=np.int32)
            o["hull"] = np.ascontiguousarray(np.concatenate([k0, k1 - k0]).astype(np.int32))
            o["gpacked_dev"] = _dev(cqt_pack_gemm(b, o["n_fft"])) if o["n_fft"] in (128, 256, 512) and len(b) <= 64 else None
            o["gsplit_dev"] = (torch.from_numpy(cqt_pack_bf16x3(b, o["n_fft"]).view(np.int16)).to(require_gpu())
                               if o["n_fft"] in (128, 256) and len(b) <= 16 else None)
        p.taps_dev = _dev(decimation_taps().astype(np.float32))
        # the one-launch form (syg_cqt_fused_f32) where the plan has its shape (CqtPlan.one_launch_shape)
        oc = p.octaves
        p.fused_ok = p.one_launch_shape() and oc[0].get("gsplit_dev") is not None
        p.row0 = np.ascontiguousarray([o["row0"] for o in oc], dtype=np.int32)
        return p
    plan = _cached(key, build)
    if y.stride(1) != 1:
        y = y.contiguous()
    B, L = y.shape
    # output frames = the smallest centred frame count over the octaves (librosa's __trim_stack); this is
    # 1 + L // hop_length except when the rounded-up decimated lengths add a frame to every octave
    Lc = L
    for _ in range(plan.early):
        Lc = (Lc + 1) // 2
    Tn = None
    for o in plan.octaves:
        To = 1 + Lc // o["hop"]
        Tn = To if Tn is None else min(Tn, To)
        if o["decimate_after"]:
            Lc = (Lc + 1) // 2
    out = torch.empty((B, plan.n_bins, Tn, 2), dtype=torch.float32, device=y.device)      # every row is written
    s2 = float(np.sqrt(2.0))
    if settings.cqt_fused and settings.cqt_mode == "bf16x3" and plan.fused_ok and Tn >= 1:
        rc = lib().syg_cqt_fused_f32(_ptr(y), B, L, _ld(y), _ptr(plan.taps_dev), plan.taps_dev.numel(), s2,
                                     _ptr(plan.octaves[0]["gsplit_dev"]), plan.octaves[0]["n"], len(plan.octaves),
                                     plan.row0.ctypes.data_as(C.c_void_p), Tn, _ptr(out), plan.n_bins * Tn,
                                     C.c_void_p(_stream_ptr()))
        check(rc, "syg_cqt_fused_f32")
        return out
    # octave kernel: "bf16x3" (default: the framed product with bfloat16-split operands, fp32-equivalent), "gemm" (the
    # single-instruction fp32 MFMA form), "fft" (rfft x sparse rows; also what other frame lengths take)
    mode = settings.cqt_mode
    use_gemm = mode != "fft"
    # settings.cqt_streams = 2: the decimation chain (memory-bound) on the caller's stream, the octave products
    # (matrix-core bound, no LDS) on a side stream, so that octave i runs beside the decimation towards octave i + 1.
    # Measured on one 1-hour stream: 1.03 ms against 1.05 ms on one stream -- the two kernels slow each other down by
    # about what the overlap saves -- so one stream is the default.
    main = torch.cuda.current_stream()
    two = settings.cqt_streams == 2
    side = _side_stream(y.device) if two else main
    if two:
        side.wait_stream(main)                    # `out` and `y` are ready for the side stream
    # every decimation level up front (three levels per pass; settings.cqt_chain = False: one launch per level)
    n_dec = plan.early + sum(1 for o in plan.octaves[:-1] if o["decimate_after"])
    chain = settings.cqt_chain and not two and n_dec > 0
    levels = None
    if chain:
        keepl = [i >= plan.early - 1 for i in range(n_dec)]
        levels = [y] + decimate2_chain(y, plan.taps_dev, s2, n_dec, keepl)
    cur = y
    lvl = plan.early
    if chain:
        cur = levels[lvl]
    else:
        for _ in range(plan.early):
            cur = decimate2(cur, plan.taps_dev, s2)
    for oi, o in enumerate(plan.octaves):
        if o["n"] > 0:
            if two:
                ev = torch.cuda.Event()
                ev.record(main)                   # `cur` has been produced on the main stream
                side.wait_event(ev)
                cur.record_stream(side)
            sp = C.c_void_p(side.cuda_stream)
            if mode == "bf16x3" and o.get("gsplit_dev") is not None:
                rc = lib().syg_cqt_octave_bf16x3_f32(_ptr(cur), B, cur.shape[1], _ld(cur), o["n_fft"], o["hop"], Tn,
                                                     _ptr(o["gsplit_dev"]), o["n"], _ptr(out), plan.n_bins * Tn, o["row0"], sp)
                check(rc, "syg_cqt_octave_bf16x3_f32")
            elif use_gemm and o.get("gpacked_dev") is not None:
                rc = lib().syg_cqt_octave_gemm_f32(_ptr(cur), B, cur.shape[1], _ld(cur), o["n_fft"], o["hop"], Tn,
                                                   _ptr(o["gpacked_dev"]), o["n"], _ptr(out), plan.n_bins * Tn, o["row0"], sp)
                check(rc, "syg_cqt_octave_gemm_f32")
            else:
                rc = lib().syg_cqt_octave_f32(_ptr(cur), B, cur.shape[1], _ld(cur), o["n_fft"], o["hop"], Tn,
                                              _ptr(twiddle_rfft_dev(o["n_fft"])), _ptr(o["basis_dev"]), o["n"],
                                              o["hull"].ctypes.data_as(C.c_void_p), _ptr(out),
                                              plan.n_bins * Tn, o["row0"], sp)
                check(rc, "syg_cqt_octave_f32")
        if o["decimate_after"] and oi + 1 < len(plan.octaves):
            lvl += 1
            cur = levels[lvl] if chain else decimate2(cur, plan.taps_dev, s2)
    if two:
        main.wait_stream(side)
    return out
