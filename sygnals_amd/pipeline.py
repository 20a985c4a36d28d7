"""Host <-> device streaming around the hot path.

The device computes a 1024-clip batch in ~0.17 ms; a PCIe 5 x16 link needs ~2 ms to deliver it as 16-bit PCM and ~4 ms
as float32 -- end to end the copy is the limiter, so the job of this module is to keep the link busy and to move as
few bytes as possible over it:

* clips travel in the files' own integer PCM and are scaled / mixed down on the device (`ops.pcm_to_f32`, the work
  librosa.load does on the host for the reference: sygnals/core/audio/io.py:84-90, cli/features_cmd.py:66-68);
* `depth` pinned staging slots: while batch k computes, batch k+1 is in flight host-to-device on its own stream and
  the results of batch k-1 return device-to-host on a third one (HIP streams + events, no host synchronisation
  inside the loop except on a slot about to be reused);
* files are decoded by a small thread pool straight into the pinned slot.

Results come back in submission order.  There is no CPU path: every batch is computed by the HIP kernels.
"""
from __future__ import annotations

from concurrent.futures import ThreadPoolExecutor
from typing import Callable, Dict, Iterable, Iterator, List, Optional, Sequence, Union

import numpy as np
import torch

from . import ops

_TORCH_OF = {np.dtype(np.int16): torch.int16, np.dtype(np.int32): torch.int32, np.dtype(np.uint8): torch.uint8,
             np.dtype(np.float32): torch.float32}
Result = Union[torch.Tensor, Sequence[torch.Tensor], Dict[str, torch.Tensor]]


def _map(res, fn):
    if isinstance(res, torch.Tensor):
        return fn(res)
    if isinstance(res, dict):
        return {k: fn(v) for k, v in res.items()}
    return type(res)(fn(v) for v in res)


def _leaves(res) -> List[torch.Tensor]:
    if isinstance(res, torch.Tensor):
        return [res]
    return list(res.values()) if isinstance(res, dict) else list(res)


class _Slot:
    def __init__(self):
        self.pinned = None       # pinned host staging buffer (input)
        self.dev = None          # device copy of it
        self.ev_in = torch.cuda.Event()
        self.ev_done = torch.cuda.Event()
        self.ev_out = torch.cuda.Event()
        self.out = None          # pinned host result(s), same structure as compute()'s return value
        self.tag = None
        self.busy = False


class DevicePipeline:
    """`compute(x)` maps a float32 device batch [B, L] to a device tensor / tuple / dict of device tensors.

    `submit(fill, shape, dtype, tag)` queues one batch: `fill(buf)` writes the host data into the pinned staging
    buffer it is handed (a torch tensor of that shape / dtype; integer PCM [B, L] or [B, L, C], or float32 [B, L]).
    `results()` / iteration yields `(tag, host_result)` in submission order, `host_result` mirroring compute()'s
    structure with NumPy arrays."""

    def __init__(self, compute: Callable[[torch.Tensor], Result], depth: int = 2, device=None):
        if depth < 1:
            raise ValueError("depth must be >= 1")
        self.device = ops.require_gpu() if device is None else torch.device(device)
        self.compute = compute
        self.slots = [_Slot() for _ in range(depth)]
        self.s_in = torch.cuda.Stream(self.device)
        self.s_out = torch.cuda.Stream(self.device)
        self.k = 0               # batches submitted
        self.j = 0               # batches handed back
        self.bytes_in = 0

    # -- one batch in ------------------------------------------------------------------------------------------
    def submit(self, fill: Callable[[torch.Tensor], None], shape, dtype: torch.dtype, tag=None):
        slot = self.slots[self.k % len(self.slots)]
        if slot.busy:
            raise RuntimeError("DevicePipeline: take the pending result (next(results())) before submitting more")
        shape = tuple(int(v) for v in shape)
        if slot.pinned is None or tuple(slot.pinned.shape) != shape or slot.pinned.dtype != dtype:
            slot.ev_in.synchronize()
            slot.pinned = torch.empty(shape, dtype=dtype).pin_memory()
            slot.dev = torch.empty(shape, dtype=dtype, device=self.device)
        else:
            slot.ev_in.synchronize()                    # the previous upload out of this buffer has finished
        fill(slot.pinned)
        self.bytes_in += slot.pinned.numel() * slot.pinned.element_size()
        cur = torch.cuda.current_stream(self.device)
        self.s_in.wait_event(slot.ev_done)              # the kernels that read slot.dev last time are done
        with torch.cuda.stream(self.s_in):
            slot.dev.copy_(slot.pinned, non_blocking=True)
            slot.ev_in.record(self.s_in)
        cur.wait_event(slot.ev_in)
        x = slot.dev if dtype == torch.float32 else ops.pcm_to_f32(slot.dev)
        res = self.compute(x)
        slot.ev_done.record(cur)
        for t in _leaves(res):
            t.record_stream(self.s_out)
        self.s_out.wait_event(slot.ev_done)
        with torch.cuda.stream(self.s_out):
            if slot.out is None or [tuple(t.shape) for t in _leaves(slot.out)] != [tuple(t.shape) for t in _leaves(res)]:
                slot.out = _map(res, lambda t: torch.empty(t.shape, dtype=t.dtype).pin_memory())
            for dst, src in zip(_leaves(slot.out), _leaves(res)):
                dst.copy_(src, non_blocking=True)
            slot.ev_out.record(self.s_out)
        slot.tag, slot.busy = tag, True
        self.k += 1

    # -- results out -------------------------------------------------------------------------------------------
    def pending(self) -> int:
        return self.k - self.j

    def take(self):
        """Oldest pending result: waits for its device-to-host copy only."""
        if self.pending() == 0:
            raise RuntimeError("DevicePipeline: nothing pending")
        slot = self.slots[self.j % len(self.slots)]
        slot.ev_out.synchronize()
        out = _map(slot.out, lambda t: t.numpy().copy())
        slot.busy = False
        self.j += 1
        return slot.tag, out

    def run(self, batches: Iterable) -> Iterator:
        """batches: iterable of host arrays (NumPy or torch; integer PCM or float32) or of
        (fill, shape, dtype[, tag]) tuples.  Yields (tag, host_result) in order; tag defaults to the batch index."""
        for i, b in enumerate(batches):
            if self.pending() == len(self.slots):
                yield self.take()
            if isinstance(b, tuple) and callable(b[0]):
                fill, shape, dtype = b[:3]
                tag = b[3] if len(b) > 3 else i
            else:
                arr = b if isinstance(b, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(b))
                if arr.dtype not in _TORCH_OF.values():
                    raise ValueError(f"unsupported batch dtype {arr.dtype}: int16 / int32 / uint8 PCM or float32")
                fill, shape, dtype, tag = (lambda buf, a=arr: buf.copy_(a)), arr.shape, arr.dtype, i
            self.submit(fill, shape, dtype, tag)
        while self.pending():
            yield self.take()


def wav_batches(paths: Sequence, batch_clips: int, length: Optional[int] = None, workers: int = 8):
    """(fill, shape, dtype, tag) tuples for DevicePipeline.run: files are decoded by `workers` threads straight into
    the pinned staging buffer, `batch_clips` per batch, all clips cut / zero padded to `length` frames (default: the
    length of the first file).  Every file of a batch must share the sample type and channel count of the first."""
    from .io import read_wav_pcm
    paths = list(paths)
    if not paths:
        return
    first, sr0 = read_wav_pcm(paths[0])
    L = int(length) if length is not None else first.shape[0]
    ch, dt = first.shape[1], first.dtype
    tdt = _TORCH_OF[np.dtype(dt)]
    fill_value = 128 if dt == np.uint8 else 0
    pool = ThreadPoolExecutor(max_workers=max(1, workers))

    def load_into(buf_np, i, p):
        d, sr = read_wav_pcm(p)
        if d.dtype != dt or d.shape[1] != ch:
            raise ValueError(f"{p}: sample type / channel count differs from the first file")
        if sr != sr0:
            raise ValueError(f"{p}: sample rate {sr} differs from the first file's {sr0}")
        n = min(d.shape[0], L)
        buf_np[i, :n] = d[:n]
        buf_np[i, n:] = fill_value

    for lo in range(0, len(paths), batch_clips):
        chunk = paths[lo:lo + batch_clips]

        def fill(buf, chunk=chunk):
            view = buf.numpy()
            if dt == np.float32 and ch > 1:
                raise ValueError("float WAV files with several channels: mix down on the host first")
            view = view if view.ndim == 3 else view[:, :, None]
            list(pool.map(lambda ip: load_into(view, *ip), enumerate(chunk)))

        shape = (len(chunk), L) if (dt == np.float32) else (len(chunk), L, ch)
        yield fill, shape, tdt, (lo, lo + len(chunk))
    pool.shutdown()


def mfcc_from_files(paths: Sequence, sr: float, batch_clips: int = 1024, length: Optional[int] = None,
                    workers: int = 8, depth: int = 2, **mfcc_kw) -> np.ndarray:
    """MFCCs of a list of equally long WAV clips -> float32 [N, n_mfcc, T] (ops.mfcc_batch per batch; keyword
    arguments as ops.mfcc_batch: n_fft, hop, n_mels, n_mfcc, ...)."""
    pipe = DevicePipeline(lambda x: ops.mfcc_batch(x, sr, **mfcc_kw), depth=depth)
    out = [r for _, r in pipe.run(wav_batches(paths, batch_clips, length, workers))]
    if not out:
        raise ValueError("mfcc_from_files: no input files")
    return np.concatenate(out, axis=0)
