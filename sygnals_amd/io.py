"""Minimal signal I/O either side of the hot path (host only).

Follows sygnals/core/data_handler.py: read_data :74 (audio -> (float64 data, sr); CSV with a
'value' column; NPZ with a 'data' key) and save_data :183-303 (CSV written with index=False --
which drops the 'time' index of a feature DataFrame, :248 -- NPZ via np.savez with the dict's own
keys, a single array under 'data').  Audio decoding uses scipy.io.wavfile (the reference goes
through librosa/soundfile, not installed here): PCM16/24/32 are scaled by 2^(bits-1).
"""
from __future__ import annotations

from pathlib import Path
from typing import Dict, Tuple, Union

import numpy as np
import pandas as pd

AUDIO_EXT = {".wav"}


def read_audio(path) -> Tuple[np.ndarray, int]:
    from scipy.io import wavfile
    sr, data = wavfile.read(str(path))
    if data.dtype.kind == "i":
        data = data.astype(np.float64) / float(2 ** (8 * data.dtype.itemsize - 1))
    elif data.dtype.kind == "u":
        data = (data.astype(np.float64) - 128.0) / 128.0
    else:
        data = data.astype(np.float64)
    if data.ndim == 2:
        data = data.T                       # (channels, samples) like librosa.load(mono=False)
        if data.shape[0] == 1:
            data = data[0]
    return data, int(sr)


def read_data(path) -> Union[pd.DataFrame, Dict[str, np.ndarray], Tuple[np.ndarray, int]]:
    p = Path(path)
    ext = p.suffix.lower()
    if not p.exists():
        raise FileNotFoundError(str(p))
    if ext in AUDIO_EXT:
        return read_audio(p)
    if ext == ".csv":
        return pd.read_csv(p)
    if ext == ".npz":
        with np.load(p, allow_pickle=False) as z:
            return {k: z[k] for k in z.files}
    raise ValueError(f"Unsupported input file format: '{ext}'.")


def signal_from(result, column: str = "value", key: str = "data") -> Tuple[np.ndarray, Union[int, None]]:
    """1-D float64 signal (+ sr if known) from whatever read_data returned."""
    if isinstance(result, tuple):
        return np.asarray(result[0], dtype=np.float64), int(result[1])
    if isinstance(result, pd.DataFrame):
        if column not in result.columns:
            raise ValueError(f"CSV input needs a '{column}' column.")
        return result[column].to_numpy(dtype=np.float64), None
    if isinstance(result, dict):
        if key not in result:
            raise ValueError(f"NPZ input needs a '{key}' array.")
        sr = int(result["sr"]) if "sr" in result else (int(result["fs"]) if "fs" in result else None)
        return np.asarray(result[key], dtype=np.float64), sr
    raise TypeError(f"Unsupported data type: {type(result)}")


def save_data(data, output_path, sr=None) -> None:
    p = Path(output_path).resolve()
    ext = p.suffix.lower()
    p.parent.mkdir(parents=True, exist_ok=True)
    if isinstance(data, pd.DataFrame):
        if ext == ".csv":
            data.to_csv(p, index=False)
        elif ext == ".npz":
            np.savez(p, **{c: data[c].values for c in data.columns})
        else:
            raise ValueError(f"Cannot save DataFrame to core format '{ext}'.")
    elif isinstance(data, np.ndarray):
        if ext == ".npz":
            np.savez(p, data=data)
        elif ext == ".csv":
            if data.ndim == 1:
                pd.DataFrame(data, columns=["value"]).to_csv(p, index=False)
            elif data.ndim == 2:
                pd.DataFrame(data).to_csv(p, index=False, header=False)
            else:
                raise ValueError("Cannot save NumPy array with >2 dimensions as CSV.")
        else:
            raise ValueError(f"Cannot save single NumPy array directly to core format '{ext}'. Use NPZ or CSV.")
    elif isinstance(data, dict) and all(isinstance(v, np.ndarray) for v in data.values()):
        if ext != ".npz":
            raise ValueError(f"Cannot save dictionary of NumPy arrays to core format '{ext}'. Use NPZ.")
        np.savez(p, **data)
    elif isinstance(data, tuple) and len(data) == 2 and ext == ".wav":
        from scipy.io import wavfile
        x = np.clip(np.asarray(data[0], dtype=np.float64), -1.0, 1.0)
        wavfile.write(str(p), int(sr or data[1]), np.round(x * 32767.0).astype(np.int16))
    else:
        raise TypeError(f"Unsupported data type for core saving handlers: {type(data)}.")


def read_clips(paths, length: Union[int, None] = None, mono: bool = True, pin: bool = False):
    """Batched ingest for the device path: decode WAV / NPZ / CSV clips into ONE float32 [B, L]
    host array (zero padded / cut to `length`, default = the longest clip), ready for a single host-to-device copy.

    Multi-channel audio is mixed down like `sygnals features extract` does (mean over channels,
    cli/features_cmd.py:66-68).  With pin=True the batch is returned as a pinned torch tensor so that
    `tensor.to(device, non_blocking=True)` overlaps with device work.  Returns (batch, sample_rates).
    """
    sigs, srs = [], []
    for p in paths:
        r = read_data(p)
        if isinstance(r, tuple) and np.ndim(r[0]) == 2:
            if not mono:
                raise ValueError(f"{p}: multi-channel audio needs mono=True (mix-down)")
            r = (np.mean(r[0], axis=0), r[1])
        y, sr = signal_from(r)
        if y.ndim != 1:
            raise ValueError(f"{p}: expected a 1-D signal, got shape {y.shape}")
        sigs.append(y)
        srs.append(sr)
    if not sigs:
        raise ValueError("read_clips: no input files")
    L = int(length) if length is not None else max(len(s) for s in sigs)
    if L < 1:
        raise ValueError("read_clips: length must be >= 1")
    batch = np.zeros((len(sigs), L), dtype=np.float32)
    for i, s in enumerate(sigs):
        n = min(len(s), L)
        batch[i, :n] = s[:n]
    if pin:
        import torch
        t = torch.from_numpy(batch)
        return (t.pin_memory() if torch.cuda.is_available() else t), srs
    return batch, srs


def read_wav_pcm(path):
    """(frames [n, channels] in the file's own integer type, sample rate) -- no conversion on the host.
    float WAV files come back as float32 (they need no scaling)."""
    from scipy.io import wavfile
    sr, data = wavfile.read(str(path))
    if data.ndim == 1:
        data = data[:, None]
    if data.dtype.kind == "f":
        data = data.astype(np.float32)
    elif data.dtype not in (np.int16, np.int32, np.uint8):
        raise ValueError(f"{path}: unsupported PCM sample type {data.dtype}")
    return data, int(sr)


def read_clips_pcm(paths, length: Union[int, None] = None, out=None):
    """Batched ingest WITHOUT host-side conversion: WAV clips -> ONE integer array [B, L, C]
    (the files' common sample type and channel count, zero padded / cut to `length`) for a single host-to-device
    copy; scaling and mono mix-down happen on the device (ops.pcm_to_f32).  `out` may be a preallocated (e.g. pinned)
    array / tensor view to fill.  Returns (batch, sample_rates)."""
    clips, srs = [], []
    for p in paths:
        d, sr = read_wav_pcm(p)
        clips.append(d)
        srs.append(sr)
    if not clips:
        raise ValueError("read_clips_pcm: no input files")
    dt, ch = clips[0].dtype, clips[0].shape[1]
    for p, d in zip(paths, clips):
        if d.dtype != dt or d.shape[1] != ch:
            raise ValueError(f"{p}: sample type / channel count differs from the first file ({d.dtype}, {d.shape[1]} "
                             f"channels vs {dt}, {ch}); batch such files separately")
    L = int(length) if length is not None else max(d.shape[0] for d in clips)
    if L < 1:
        raise ValueError("read_clips_pcm: length must be >= 1")
    if out is None:
        out = np.empty((len(clips), L, ch), dtype=dt)
    elif tuple(out.shape) != (len(clips), L, ch):
        raise ValueError(f"read_clips_pcm: out has shape {tuple(out.shape)}, need {(len(clips), L, ch)}")
    fill = 128 if dt == np.uint8 else 0                      # unsigned 8-bit PCM is offset binary: 128 is silence
    for i, d in enumerate(clips):
        n = min(d.shape[0], L)
        out[i, :n] = d[:n] if isinstance(out, np.ndarray) else _as_tensor(d[:n])
        if n < L:
            out[i, n:] = fill
    return out, srs


def _as_tensor(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a))
