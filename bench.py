#!/usr/bin/env python3
"""bench.py -- headline benchmark: Msamples/s of batched STFT -> mel -> MFCC on MI355X.

Workload (BASELINE.json configs[1], "C2"): per GPU 1024 synthetic clips of 1 s @ 48 kHz
(49 152 000 samples), STFT n_fft=2048 hop=512 hann center -> |X|^2 -> 40 Slaney mel bands ->
power_to_db(ref=max, top_db=80) -> DCT-II ortho, 13 coefficients.  Inputs are resident in HBM
when the timed region starts.  One step = one pass of the hot path over the batch: ONE kernel launch
(samples in, MFCCs out) on a single GPU.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

N > 1: one process per GPU, every rank owns its own 1024 clips (weak scaling), no data-path
collective; the only exchange is the RCCL gather of the [1024, 13, 94] result blocks to rank 0,
inside the timed region, asynchronous: the gather of step k runs beside the kernels of step k+1.
A workgroup of the fused kernels fills a whole CU, so a communication kernel that needs CUs at the
same time either waits for a full launch or makes that launch wait for it (measured with a stand-in:
tools/queue_bench.py -> profiles/r01_coresidency.json, 167 -> 217 us per step).  For N > 1 the step
therefore runs as the two-launch form (STFT->mel kernel + per-clip dB/DCT kernel, +3 us) on all but
8 / 16 / 32 compute units (N = 2 / 4 / 8: four RCCL channels per peer at the root, plus a few), and RCCL's point-to-point channels are capped so that its workgroups fit
the CUs set aside (NCCL_NCHANNELS_PER_PEER / NCCL_MAX_P2P_NCHANNELS; all three can be overridden from
the environment).  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np

SR = 48000
L = 48000
B_PER_GPU = 1024
P2P_CHANNELS_PER_PEER = 4     # N > 1: RCCL point-to-point channels (= workgroups) per peer; the CUs set aside follow
N_FFT, HOP, N_MELS, N_MFCC = 2048, 512, 40, 13
T_FRAMES = 1 + L // HOP
ALGO_BYTES_PER_CLIP = 4 * L + 4 * N_MFCC * T_FRAMES        # 196 888 B: read samples, write MFCCs
HBM_PEAK_GBS = 8000.0                                      # MI355X_MICROARCH.md: 8.0 TB/s spec


def _cpu_worker(args):
    seed, n = args
    for k in ("OMP_NUM_THREADS", "MKL_NUM_THREADS", "OPENBLAS_NUM_THREADS"):
        os.environ[k] = "1"
    from oracle import cpu_ref as O
    Y = O.synth_clips(n, L, SR, seed=seed)
    t0 = time.perf_counter()
    O.mfcc_batch(Y, SR, N_FFT, HOP, N_MELS, N_MFCC)
    return time.perf_counter() - t0


def cpu_baseline(target_seconds=10.0):
    """The oracle (float64 NumPy/SciPy port of the reference CPU path) on the host cores of this box.

    Runs BEFORE the GPU is initialised (worker processes are forked).  Sample: clips of the same
    synthetic recipe, one clip per call as the reference does, all host cores busy (one process
    per core, BLAS threads pinned to 1); sized from a single-core probe to about `target_seconds`.
    """
    import multiprocessing as mp
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    cores = max(1, min(avail, 16))          # a 1-GPU box's CPU share is 16 cores
    _cpu_worker((1, 4))                       # warm-up (imports, FFT plan caches)
    probe = _cpu_worker((2, 16))
    per_clip = probe / 16
    n_per = max(8, int(target_seconds / per_clip))
    n_per = min(n_per, 4096)
    t0 = time.perf_counter()
    with mp.get_context("fork").Pool(cores) as pool:
        pool.map(_cpu_worker, [(100 + i, n_per) for i in range(cores)])
    wall = time.perf_counter() - t0
    clips = cores * n_per
    return {"value": round(clips * L / wall / 1e6, 3), "unit": "Msamples/s", "cores": cores, "kind": "port",
            "sample": f"{clips} clips x 1 s @ 48 kHz (same recipe/config as the GPU workload), float64 oracle, "
                      f"one clip per call, {cores} processes x 1 thread, {wall:.1f} s wall",
            "single_core_value": round(L / per_clip / 1e6, 3)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # a step is ~0.2 ms and the GPU needs ~150 steps (30 ms) from idle to reach its steady clock
    # (tools/warm_curve.py), hence the long default warm-up; the whole default run is still ~0.2 s of GPU time
    ap.add_argument("--steps", type=int, default=500)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--clips", type=int, default=B_PER_GPU, help="clips per GPU (default: the C2 batch)")
    a = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        if world == 1 and a.gpus > 1:
            sys.exit("bench.py: --gpus N > 1 must be launched with torch.distributed.run (one process per GPU)")
        a.gpus = world

    cpu = None
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        cpu = cpu_baseline()                  # before any GPU initialisation (forks workers)

    import torch
    import torch.distributed as dist
    from oracle import cpu_ref as O           # synthetic-input recipe only
    from sygnals_amd import ops
    from sygnals_amd.distributed import RootGather

    # rehearsal on a 1-GPU box: SYG_BENCH_SAME_GPU=1 puts every rank on cuda:0 and exchanges through gloo
    # (RCCL refuses two ranks on one device); the real multi-GPU run uses one GPU per rank over RCCL
    same_gpu = os.environ.get("SYG_BENCH_SAME_GPU") == "1"
    dev_index = 0 if same_gpu else local_rank
    torch.cuda.set_device(dev_index)
    reserve = 0
    if world > 1:
        # CUs set aside for the collective + RCCL capped to fit them (see the module docstring); set before the
        # process group and the first launch read them
        nch = P2P_CHANNELS_PER_PEER * (world - 1)             # RCCL workgroups at the root while a gather runs
        os.environ.setdefault("SYGNALS_AMD_RESERVE_CUS", str(8 * ((nch + 4 + 7) // 8)))   # 8 / 16 / 32 for N = 2 / 4 / 8
        os.environ.setdefault("NCCL_NCHANNELS_PER_PEER", str(P2P_CHANNELS_PER_PEER))
        os.environ.setdefault("NCCL_MAX_P2P_NCHANNELS", str(nch))
        reserve = int(os.environ["SYGNALS_AMD_RESERVE_CUS"])
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if same_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))

    B = a.clips
    base = O.synth_clips(64, L, SR, seed=20250523 + rank)          # 64 distinct clips, tiled to the batch
    y = ops.to_device_f32(np.tile(base, (B // 64 + 1, 1))[:B])
    n_total = B * world

    # N > 1: every step's MFCC block is gathered to rank 0 (the only exchange of the path).  The gather
    # of step k is asynchronous and overlaps the kernel of step k+1; all gathers are finished inside the timed
    # region.
    T_frames = 1 + L // HOP
    gat = RootGather(n_total, (B, N_MFCC, T_frames), torch.float32, "cpu" if same_gpu else torch.device("cuda", dev_index))

    def step():
        # N > 1: two-launch form (tile-granular shares; robust to the CUs the gather holds), else one launch
        out = ops.mfcc_batch(y, SR, N_FFT, HOP, N_MELS, N_MFCC, fused=False if world > 1 else None)
        if world > 1:
            gat.finish()                               # (the previous step's gather; a stream-side wait)
            gat.start(out.cpu() if same_gpu else out)
        return out

    def sync():
        if world > 1:
            gat.finish()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(a.warmup):
        step()
    sync()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    sync()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if same_gpu else "cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # dominant kernel: the one-launch STFT -> mel -> dB -> DCT kernel (MODE 3 of stft2048_kernel; the whole step
    # at this configuration), HIP events on the stream it is launched on
    roof = None
    if rank == 0:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = max(10, min(a.steps, 100))
        # back-to-back launches between two events on the launch stream (steady clock: the timed steps above
        # have just run); the average includes the ~2 us dispatch gap, as rocprofv3's per-dispatch average does not
        one_launch = world == 1
        e0.record()
        for _ in range(reps):
            if one_launch:
                ops.stft2048_mfcc(y, SR, HOP, True, "hann", N_MELS, N_MFCC)
            else:                                   # N > 1 runs the two-launch form: its dominant kernel is the mel kernel
                ops.stft2048_mel(y, SR, HOP, True, "hann", 2048, N_MELS)
        e1.record()
        e1.synchronize()
        kdur = e0.elapsed_time(e1) * 1e-3 / reps
        # algorithmic bytes of the path this launch carries: every sample read once, MFCCs (N > 1: the mel matrix,
        # which the second launch turns into MFCCs) written once
        kbytes = B * ALGO_BYTES_PER_CLIP if one_launch else B * (4 * L + 4 * N_MELS * T_frames)
        achieved = kbytes / kdur / 1e9
        traffic = None
        tp = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tp):
            try:
                traffic = json.load(open(tp)).get("dominant_kernel_bytes_per_launch") if one_launch else None
            except Exception:
                traffic = None
        roof = {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                "kernel": "stft2048_kernel<16,2,3> (16 waves, staged tiles, clip-resident MFCC)" if one_launch else
                          f"stft2048_kernel<16,2,0> (16 waves, staged tiles, mel out) on all but {reserve} CUs",
                "kernel_avg_us": round(kdur * 1e6, 2),
                "algorithmic_bytes_per_launch": kbytes}

    if rank == 0:
        samples = n_total * L * a.steps
        value = samples / elapsed / 1e6
        job_bytes = n_total * ALGO_BYTES_PER_CLIP * a.steps
        line = {
            "metric": "Msamples/s STFT->MFCC (n_fft=2048, hop=512)", "value": round(value, 1), "unit": "Msamples/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(elapsed / a.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "C2: 1024 x 1 s @ 48 kHz clips per GPU, STFT n_fft=2048 hop=512 hann center -> "
                                   "40-band Slaney mel -> power_to_db(ref=max, top_db=80) -> 13 MFCC (DCT-II ortho)",
                       "clips_per_gpu": B, "clip_samples": L, "sr": SR, "n_fft": N_FFT, "hop": HOP,
                       "n_mels": N_MELS, "n_mfcc": N_MFCC, "parallelism": f"clip-sharded x{world}" +
                       (f", two-launch form on all but {reserve} CUs, asynchronous RCCL gather to rank 0 in the timed "
                        f"region" if world > 1 else ", one launch per step")},
            "hbm_roofline_frac_whole_step": round(job_bytes / elapsed / 1e9 / (HBM_PEAK_GBS * world), 5),
            "roofline": roof, "cpu_baseline": cpu,
        }
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
