"""GPU tests of the ingest side of the path: integer PCM -> float32 on the device
(syg_pcm_to_f32) and the overlapped host -> device -> host pipeline, end to end from WAV files to MFCCs."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")
from oracle import cpu_ref as O
from tests.gpu_util import assert_parity

TOL = 1e-5


@pytest.fixture(scope="module")
def ops():
    from sygnals_amd import ops
    ops.require_gpu()
    return ops


@pytest.mark.parametrize("dtype,scale,off", [(np.int16, 32768.0, 0), (np.int32, 2147483648.0, 0), (np.uint8, 128.0, 128)])
@pytest.mark.parametrize("channels", [1, 2, 3])
def test_pcm_to_f32_matches_host_conversion(ops, dtype, scale, off, channels):
    """Bit-identical to the host conversion: float64 scale and channel mean, rounded to float32 once."""
    rng = np.random.default_rng(channels * 10 + np.dtype(dtype).itemsize)
    info = np.iinfo(dtype)
    for L in (1, 3, 4, 1000, 4097):
        pcm = rng.integers(info.min, info.max, (5, L, channels), dtype=dtype, endpoint=True)
        pcm[0, 0, :] = info.min; pcm[0, -1, :] = info.max
        arg = pcm if channels > 1 else pcm[:, :, 0]
        got = ops.pcm_to_f32(torch.from_numpy(np.ascontiguousarray(arg)).cuda()).cpu().numpy()
        want = ((pcm.astype(np.float64) - off) / scale).mean(axis=2)
        assert np.array_equal(got, want.astype(np.float32)), (dtype, channels, L)


def test_pcm_to_f32_argument_errors(ops):
    with pytest.raises(ValueError, match="int16 / int32 / uint8"):
        ops.pcm_to_f32(torch.zeros((2, 8), dtype=torch.float32, device="cuda"))
    with pytest.raises(ValueError, match="out must be"):
        ops.pcm_to_f32(torch.zeros((2, 8), dtype=torch.int16, device="cuda"), out=torch.zeros((2, 9), device="cuda"))


def _write_wavs(tmp_path, n, L, sr, channels, rng):
    from scipy.io import wavfile
    paths, clips = [], []
    for i in range(n):
        y = O.synth_clips(1, L, sr, seed=100 + i)[0]
        pcm = np.round(y * 20000).astype(np.int16)
        if channels == 2:
            pcm = np.stack([pcm, np.round(pcm * 0.5).astype(np.int16)], axis=1)
        p = tmp_path / f"c{i:03d}.wav"
        wavfile.write(str(p), sr, pcm)
        paths.append(p)
        clips.append(pcm)
    return paths, clips


@pytest.mark.parametrize("channels", [1, 2])
def test_mfcc_from_wav_files_matches_oracle(ops, tmp_path, channels):
    """23 PCM16 files, batches of 8 (last one ragged), two staging slots: MFCCs equal the oracle's on the samples
    librosa.load would hand the reference (int / 32768, channels averaged)."""
    from sygnals_amd.pipeline import mfcc_from_files
    sr, L = 16000, 16000
    paths, clips = _write_wavs(tmp_path, 23, L, sr, channels, np.random.default_rng(0))
    got = mfcc_from_files(paths, sr, batch_clips=8, workers=4, n_mels=40, n_mfcc=13)
    assert got.shape == (23, 13, 1 + L // 512)
    for i in (0, 7, 8, 15, 22):
        y = clips[i].astype(np.float64) / 32768.0
        y = y if channels == 1 else y.mean(axis=1)
        want = O.mfcc_manager(y, sr, n_mels=40, n_mfcc=13)
        assert_parity(got[i], want, TOL, f"clip {i}")


def test_pipeline_order_structure_and_slot_reuse(ops):
    """Results come back in submission order with the structure compute() returned (tensor / tuple / dict), for more
    batches than slots, mixed batch sizes and both integer and float32 input."""
    from sygnals_amd.pipeline import DevicePipeline
    rng = np.random.default_rng(5)
    batches = [rng.integers(-3000, 3000, (b, 4096), dtype=np.int16) for b in (4, 4, 2, 4, 1, 4, 4)]

    def compute(x):
        return {"sum": x.sum(dim=1), "first": x[:, :3].clone()}
    pipe = DevicePipeline(compute, depth=3)
    out = list(pipe.run(batches))
    assert [t for t, _ in out] == list(range(len(batches)))
    for (tag, r), b in zip(out, batches):
        f = b.astype(np.float32) / 32768.0
        np.testing.assert_allclose(r["sum"], f.sum(axis=1), rtol=1e-5, atol=1e-6)
        assert np.array_equal(r["first"], f[:, :3])
    fb = [rng.normal(0, 1, (3, 100)).astype(np.float32) for _ in range(5)]
    out = list(DevicePipeline(lambda x: (x * 2.0, x + 1.0), depth=2).run(fb))
    for (tag, r), b in zip(out, fb):
        assert np.array_equal(r[0], b * 2.0) and np.array_equal(r[1], b + 1.0)
    with pytest.raises(ValueError, match="unsupported batch dtype"):
        list(DevicePipeline(lambda x: x).run([np.zeros((2, 8), dtype=np.float64)]))
