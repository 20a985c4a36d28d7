"""Multi-rank path rehearsed on CPU: world_size-2 gloo process group, clip sharding + gather to rank 0.
The per-rank compute is supplied by the test (the float64 oracle) -- the sharding/gather logic is what is
under test; on the GPU box the same code runs with the HIP compute and the RCCL backend (bench.py)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, n_total, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import cpu_ref as O
    from sygnals_amd.distributed import run_sharded, shard_range
    allc = O.synth_clips(n_total, 4096, 16000, seed=3)

    def clips(lo, hi):
        return torch.from_numpy(allc[lo:hi])

    def compute(x):
        return torch.from_numpy(O.mfcc_batch(x.numpy(), 16000, 2048, 512, 40, 13))

    out = run_sharded(clips, compute, n_total, dst=0)
    lo, hi = shard_range(n_total, rank, world)
    q.put((rank, lo, hi, None if out is None else out.numpy()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_total", [4, 5])
def test_two_rank_shard_and_gather(n_total):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_total, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    res.sort(key=lambda t: t[0])
    assert res[0][1] == 0 and res[0][2] == res[1][1] and res[1][2] == n_total      # contiguous cover
    assert res[1][3] is None                                                       # only the root holds the result
    from oracle import cpu_ref as O
    ref = O.mfcc_batch(O.synth_clips(n_total, 4096, 16000, seed=3), 16000, 2048, 512, 40, 13)
    assert res[0][3].shape == ref.shape and np.array_equal(res[0][3], ref)


def test_shard_range_properties():
    from sygnals_amd.distributed import shard_range
    for n in (0, 1, 7, 1024, 16384):
        for w in (1, 2, 3, 4, 8):
            spans = [shard_range(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_range(4, 2, 2)


def _pipeline_worker(rank, world, port, n_total, steps, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from sygnals_amd.distributed import RootGather, shard_range
    lo, hi = shard_range(n_total, rank, world)
    gat = RootGather(n_total, (hi - lo, 3, 5), torch.float32, "cpu", dst=0)
    got = []
    for k in range(steps):                      # bench.py's loop shape: compute k, finish k-1, start k
        block = (torch.arange(lo, hi, dtype=torch.float32)[:, None, None] * 1000 + k) * torch.ones(1, 3, 5)
        prev = gat.finish()
        if prev is not None:
            got.append(prev.clone())
        gat.start(block)
    last = gat.finish()
    if last is not None:
        got.append(last.clone())
    q.put((rank, [g.numpy() for g in got]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_total", [6, 7])
def test_two_rank_pipelined_gather(n_total):
    """RootGather: one asynchronous gather in flight, double-buffered receive side, even and uneven shards."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    steps = 5
    procs = [ctx.Process(target=_pipeline_worker, args=(r, 2, port, n_total, steps, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[1] == []                                        # only the root assembles results
    assert len(res[0]) == steps
    for k, g in enumerate(res[0]):
        want = (np.arange(n_total, dtype=np.float32)[:, None, None] * 1000 + k) * np.ones((1, 3, 5), np.float32)
        assert g.shape == want.shape and np.array_equal(g, want), f"step {k}"


def test_root_gather_single_process_passthrough():
    from sygnals_amd.distributed import RootGather
    gat = RootGather(4, (4, 2), torch.float32, "cpu")
    x = torch.arange(8, dtype=torch.float32).reshape(4, 2)
    gat.start(x)
    assert gat.finish() is x and gat.finish() is None


# ---- config C4's payload: every rank's [b_r, 22, T] feature block (13 MFCC + centroid + rolloff + 7 contrast rows)
# gathered to rank 0 -- the per-rank compute is the float64 oracle here (CPU); on the GPU box bench.py --config c4 runs
# sygnals_amd.core.features.manager.feature_block under the same RootGather
C4_FEATS = ["mfcc", "spectral_centroid", "spectral_rolloff", "spectral_contrast"]


def _c4_block_oracle(Y, sr):
    from oracle import cpu_ref as O
    rows = []
    for y in Y:
        d = O.extract_features(y.astype(np.float64), sr, C4_FEATS, feature_params={"mfcc": {"n_mels": 40}})
        rows.append(np.stack([d[k] for k in d if k != "time"]))
    return np.stack(rows).astype(np.float32)


def _c4_worker(rank, world, port, n_total, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from sygnals_amd.distributed import RootGather, shard_range
    from sygnals_amd.synth import synth_clips
    lo, hi = shard_range(n_total, rank, world)
    allc = synth_clips(n_total, 8192, 48000, seed=9)
    T = 1 + 8192 // 512
    gat = RootGather(n_total, (hi - lo, 22, T), torch.float32, "cpu", dst=0)
    gat.start(torch.from_numpy(_c4_block_oracle(allc[lo:hi], 48000)))
    out = gat.finish()
    q.put((rank, None if out is None else out.numpy()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_total", [4, 5])
def test_two_rank_c4_feature_block_gather(n_total):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_c4_worker, args=(r, 2, port, n_total, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[1] is None
    from sygnals_amd.synth import synth_clips
    want = _c4_block_oracle(synth_clips(n_total, 8192, 48000, seed=9), 48000)
    assert res[0].shape == (n_total, 22, 17) and np.array_equal(res[0], want)


def test_synth_recipe_matches_the_oracles_statement():
    """bench.py's inputs come from sygnals_amd.synth (no oracle import on the GPU leg); the oracle states the same
    SURVEY 8(d) recipe: bit-identical clips."""
    from oracle import cpu_ref as O
    from sygnals_amd.synth import synth_clips
    assert np.array_equal(synth_clips(3, 4800, 48000, seed=5), O.synth_clips(3, 4800, 48000, seed=5))
    assert np.array_equal(synth_clips(2, 1600, 16000, seed=20250523), O.synth_clips(2, 1600, 16000, seed=20250523))
