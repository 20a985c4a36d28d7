#!/usr/bin/env python3
"""bench.py -- headline benchmark: Msamples/s of batched STFT -> mel -> MFCC on MI355X.

Default workload (BASELINE.json configs[1], "C2"): per GPU 1024 synthetic clips of 1 s @ 48 kHz
(49 152 000 samples), STFT n_fft=2048 hop=512 hann center -> |X|^2 -> 40 Slaney mel bands ->
power_to_db(ref=max, top_db=80) -> DCT-II ortho, 13 coefficients.  Inputs are resident in HBM
when the timed region starts.  One step = one pass of the hot path over the batch: ONE kernel launch
(samples in, MFCCs out) on every GPU.

    python bench.py --gpus N --steps K --warmup W [--config c2|c4] [--prewarm P]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Both shapes work for N > 1: started plainly (no WORLD_SIZE in the environment), the parent -- which never touches the
GPU -- times the CPU baseline, then starts `python -m torch.distributed.run --nproc-per-node N bench.py ...` as a fresh
child process, hands it the baseline through the environment, relays rank 0's JSON line and the child's return code.

`--config c4` (BASELINE.json configs[3]): per GPU 2048 clips, MFCC + spectral centroid + rolloff + contrast packed
to one [2048, 22, 94] block per step, gathered to rank 0.

N > 1: one process per GPU, every rank owns its own clips (weak scaling), no data-path collective; the only
exchange is the RCCL gather of the result blocks to rank 0, inside the timed region, asynchronous: the gather of
step k runs beside the kernels of step k+1.  Every rank runs exactly the kernels of the N = 1 run.  `--reserve-cus n`
(opt-in, unmeasured on a multi-GPU node) switches to the two-launch form on all but n CUs and caps RCCL's
point-to-point channels to fit them.

`value` / `ms_per_step` are the driver's protocol and nothing else: W warm-up steps, then exactly K timed steps between
two barrier + synchronize pairs.  On a GPU coming from idle the first ~150 launches (30 ms) run below the steady clock,
so a short run (K = 20) reads lower than a long one; the same K steps timed again behind `--prewarm` further launches
are reported BESIDE the value as `steady_clock` (never as `value`).  Rank 0 prints ONE JSON line.

N > 1 adds what the first hardware run needs to diagnose itself: `compute_only_ms` (the same K steps with the gather
off, timed first), `gather_wait_ms` (host time spent inside the waits for the gathers of the timed steps, per step),
`ranks_seen` (world size and every rank's device), and -- `--reserve-cus auto` -- one re-timing with 32 CUs set aside
for the collective when a step costs more than 1.15 x its compute.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# one BLAS / OpenMP thread per process, set BEFORE numpy loads its BLAS: the CPU baseline runs one process per core, and a
# forked worker whose BLAS pool was sized for every core of the box (256 here) turns 256 workers into 65 536 threads
for _k in ("OMP_NUM_THREADS", "MKL_NUM_THREADS", "OPENBLAS_NUM_THREADS"):
    os.environ.setdefault(_k, "1")

import numpy as np

SR = 48000
L = 48000
N_FFT, HOP, N_MELS, N_MFCC = 2048, 512, 40, 13
T_FRAMES = 1 + L // HOP
HBM_PEAK_GBS = 8000.0                                      # MI355X_MICROARCH.md: 8.0 TB/s spec
MFMA_F32_PEAK_TFLOPS = 157.3                               # MI355X_MICROARCH.md: fp32-input MFMA, dense
CONFIGS = {
    # clips per GPU, feature rows per clip, algorithmic bytes per clip (read samples, write the rows)
    "c2": {"clips": 1024, "rows": N_MFCC},
    "c4": {"clips": 2048, "rows": N_MFCC + 1 + 1 + 7},
}


def algo_bytes_per_clip(rows):
    return 4 * L + 4 * rows * T_FRAMES                     # C2: 196 888 B, C4: 200 272 B


def _cpu_worker(args):
    """n clips through the oracle's C2 chain, one clip per call; returns the seconds spent IN the chain (the synthesis of the
    worker's 32 distinct clips is not part of the path and not timed)."""
    seed, n = args
    try:
        from threadpoolctl import threadpool_limits
        threadpool_limits(1)
    except Exception:
        pass
    from oracle import cpu_ref as O
    Y = O.synth_clips(min(32, n), L, SR, seed=seed)
    el, done = 0.0, 0
    while done < n:
        m = min(len(Y), n - done)
        t0 = time.perf_counter()
        O.mfcc_batch(Y[:m], SR, N_FFT, HOP, N_MELS, N_MFCC)
        el += time.perf_counter() - t0
        done += m
    return el


def _cgroup_cpu_quota():
    """CPU quota of this process's cgroup in cores (None: unlimited / unknown)."""
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]          # cgroup v2
        return None if q == "max" else float(q) / float(per)
    except Exception:
        pass
    try:
        q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())         # cgroup v1
        per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        return None if q <= 0 else q / per
    except Exception:
        return None


def cpu_baseline(target_seconds=10.0, all_cores_seconds=4.0):
    """The oracle (float64 NumPy/SciPy port of the reference CPU path) on the host cores of this box.

    Runs BEFORE the GPU is initialised (worker processes are forked).  Sample: clips of the same
    synthetic recipe, one clip per call as the reference does, all workers busy at the same time (one process
    per core, BLAS threads pinned to 1); sized from a single-core probe to about `target_seconds`.  The rate is
    clips x samples / the slowest worker's time inside the chain.
    """
    import multiprocessing as mp
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    quota = _cgroup_cpu_quota()               # cores this process can actually get (a 1-GPU box: 16 of the host's 256)
    cores = max(1, min(avail, 16))          # a 1-GPU box's CPU share is 16 cores (the box's own count is reported too)
    _cpu_worker((1, 4))                       # warm-up (imports, FFT plan caches)
    probe = _cpu_worker((2, 16))
    per_clip = probe / 16
    n_per = max(8, int(target_seconds / per_clip))
    n_per = min(n_per, 8192)
    t0 = time.perf_counter()
    with mp.get_context("fork").Pool(cores) as pool:
        els = pool.map(_cpu_worker, [(100 + i, n_per) for i in range(cores)], chunksize=1)
    wall = time.perf_counter() - t0
    clips = cores * n_per
    out = {"value": round(clips * L / max(els) / 1e6, 3), "unit": "Msamples/s", "cores": cores, "kind": "port",
           "sample": f"{clips} clips x 1 s @ 48 kHz (same recipe/config as the C2 GPU workload), float64 oracle, "
                     f"one clip per call, {cores} processes x 1 thread side by side, slowest worker {max(els):.1f} s in the chain "
                     f"({wall:.1f} s wall with start-up and clip synthesis)",
           "single_core_value": round(L / per_clip / 1e6, 3), "cores_available": avail}
    out["cpu_quota_cores"] = quota
    usable = avail if quota is None else min(avail, int(quota + 0.5))
    if usable <= cores and avail > cores:
        out["all_cores_note"] = (f"the affinity mask shows {avail} cores but the cgroup CPU quota is {quota:.1f}: the {cores}-process "
                                 "figure IS the all-core figure of this box share (256 processes on it, measured once: 96 Msamples/s "
                                 "in 290 s -- oversubscribed; profiles/r04_cpu_all_cores_256proc.json)")
    if usable > cores:
        # SURVEY 8(d) "(ii) all host cores": one process per core the affinity mask shows (a box's cgroup share may be
        # smaller than its mask -- the figure is what this process can actually get), a shorter second run
        n_all = min(usable, 256)
        # sized so that the run stays near `all_cores_seconds` even if the extra cores are not really there (a mask of 256
        # on a 16-core share): the parallelism the first pool actually got decides
        got = clips * per_clip / max(els)
        n_each = max(4, min(n_per, int(all_cores_seconds * got / (n_all * per_clip))))
        t0 = time.perf_counter()
        with mp.get_context("fork").Pool(n_all) as pool:
            els = pool.map(_cpu_worker, [(1000 + i, n_each) for i in range(n_all)], chunksize=1)
        wall_all = time.perf_counter() - t0
        out["all_cores_value"] = round(n_all * n_each * L / max(els) / 1e6, 3)
        out["all_cores"] = n_all
        out["all_cores_sample"] = (f"{n_all * n_each} clips, {n_all} processes x 1 thread side by side, slowest worker "
                                   f"{max(els):.1f} s in the chain ({wall_all:.1f} s wall)")
    return out


CPU_ENV = "SYG_BENCH_CPU_BASELINE"


def launch_ranks(a):
    """`python bench.py --gpus N` with N > 1 and no rendezvous in the environment: this process (which makes no GPU call,
    so the children start on an untouched device) times the CPU baseline, then runs one rank per GPU under
    torch.distributed.run as a child process and relays its output and return code.  The rendezvous port is picked by
    binding port 0; should another process take it before the child binds (the child then dies in its rendezvous, before
    any GPU work), one more child is started on a fresh port."""
    import socket
    import subprocess
    env = dict(os.environ)
    if not a.no_cpu_baseline and not env.get(CPU_ENV):
        env[CPU_ENV] = json.dumps(cpu_baseline())
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    rc = 1
    for attempt in range(2):
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        p = subprocess.run(cmd, env=env, stderr=subprocess.PIPE, text=True)
        sys.stderr.write(p.stderr)
        rc = p.returncode
        if rc == 0 or "address already in use" not in p.stderr.lower():
            break
    return rc


def mfma_flops_per_launch(ops, B, segments):
    """fp32 MFMA flops one launch of the one-launch MFCC kernel issues (zero-weight steps included -- they are issued):
    the per-clip dB/DCT epilogue (v_mfma_f32_16x16x4_f32: 2048 flop) and, in the matrix form of the projection only, the
    block-sparse filterbank (v_mfma_f32_4x4x1_16b_f32: 16 blocks x 4 x 4 x 1 x 2 = 512 flop per instruction, `steps`
    instructions per wave and 16-frame tile).  The segment-sum projection (MODE 6) runs on the vector pipe."""
    cfg = ops.mel_config(SR, N_FFT, N_MELS, waves=16)
    waves, steps = int(cfg.plan[1]), int(cfg.plan[2])
    tiles = (T_FRAMES + 15) // 16
    dct_steps = ((N_MFCC + 15) // 16) * tiles * (4 * ((N_MELS + 15) // 16))
    return B * ((0 if segments else tiles * waves * steps * 512) + dct_steps * 2048)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=500)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--prewarm", type=int, default=300,
                    help="launches between the driver-protocol measurement (`value`) and the `steady_clock` re-timing; 0: skip it")
    ap.add_argument("--config", choices=sorted(CONFIGS), default="c2")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--clips", type=int, default=0, help="clips per GPU (default: the configuration's)")
    ap.add_argument("--reserve-cus", default="auto",
                    help="N > 1: leave this many CUs to RCCL (two-launch form, capped p2p channels); `auto` (default): time the "
                         "plain form -- that is `value` -- and, when a step costs more than 1.15 x its compute, once more with 32 CUs "
                         "set aside, reported beside it as `reserve_cus_retry`")
    a = ap.parse_args()
    reserve_auto = a.reserve_cus == "auto"
    a.reserve_cus = 0 if reserve_auto else int(a.reserve_cus)

    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        sys.exit(launch_ranks(a))             # plain `python bench.py --gpus N`: start the ranks ourselves

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    a.gpus = world

    # CPU baseline: on rank 0, before any GPU initialisation (forks workers); a launching parent has already timed it
    cpu = None
    if rank == 0 and not a.no_cpu_baseline:
        if os.environ.get(CPU_ENV):
            cpu = json.loads(os.environ[CPU_ENV])
        else:
            cpu = cpu_baseline()

    import torch
    import torch.distributed as dist
    from sygnals_amd import ops
    from sygnals_amd.distributed import RootGather
    from sygnals_amd.synth import synth_clips

    # rehearsal on a 1-GPU box: SYG_BENCH_SAME_GPU=1 puts every rank on cuda:0 and exchanges through gloo
    # (RCCL refuses two ranks on one device); the real multi-GPU run uses one GPU per rank over RCCL
    same_gpu = os.environ.get("SYG_BENCH_SAME_GPU") == "1"
    # SYG_BENCH_DRY=1 (tests/test_bench_launcher.py, no GPU): rehearses the launcher, the rendezvous, the gather and the
    # JSON line with an all-zero stand-in for the kernels; the line says so and carries no measurement
    dry = os.environ.get("SYG_BENCH_DRY") == "1"
    same_gpu = same_gpu or dry
    dev_index = 0 if same_gpu else local_rank
    if not dry:
        torch.cuda.set_device(dev_index)
    reserve = a.reserve_cus if world > 1 else 0
    if world > 1:
        if reserve > 0:
            # opt-in: CUs set aside for the collective + RCCL capped to fit them; set before the process group and
            # the first launch read them
            per_peer = max(1, (reserve - 4) // (world - 1))
            if not dry:
                ops.set_reserved_cus(reserve)
            os.environ.setdefault("NCCL_NCHANNELS_PER_PEER", str(min(4, per_peer)))
            os.environ.setdefault("NCCL_MAX_P2P_NCHANNELS", str(min(4, per_peer) * (world - 1)))
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if same_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))

    cfgc = CONFIGS[a.config]
    B = a.clips or cfgc["clips"]
    rows = cfgc["rows"]
    n_total = B * world
    one_launch = reserve == 0
    if not dry:
        base = synth_clips(64, L, SR, seed=20250523 + rank)          # 64 distinct clips, tiled to the batch
        y = ops.to_device_f32(np.tile(base, (B // 64 + 1, 1))[:B])

    if dry:
        zero = torch.zeros((B, rows, T_FRAMES), dtype=torch.float32)
        def compute():
            return zero
    elif a.config == "c2":
        def compute():
            return ops.mfcc_batch(y, SR, N_FFT, HOP, N_MELS, N_MFCC, fused=None if one_launch else False)
    else:
        from sygnals_amd.core.features.manager import feature_block
        def compute():
            return feature_block(y, SR, HOP, n_mels=N_MELS, n_mfcc=N_MFCC)

    # N > 1: every step's result block is gathered to rank 0 (the only exchange of the path).  The gather
    # of step k is asynchronous and overlaps the kernel of step k+1; all gathers are finished inside the timed region.
    gat = RootGather(n_total, (B, rows, T_FRAMES), torch.float32, "cpu" if same_gpu else torch.device("cuda", dev_index))
    backend = "gloo (same-GPU / dry rehearsal, through the host)" if same_gpu else "RCCL"
    wait_acc = [0.0]

    def finish_gather():
        t0 = time.perf_counter()
        gat.finish()                                   # (the previous step's gather; a stream-side wait under RCCL)
        wait_acc[0] += time.perf_counter() - t0

    def step(gather=True):
        out = compute()
        if world > 1 and gather:
            finish_gather()
            gat.start(out.cpu() if same_gpu else out)
        return out

    def dev_sync():
        if not dry:
            torch.cuda.synchronize()

    def sync():
        if world > 1:
            finish_gather()
        dev_sync()
        if world > 1:
            dist.barrier()
            dev_sync()

    def timed(steps, gather=True):
        sync()
        wait_acc[0] = 0.0
        t0 = time.perf_counter()
        for _ in range(steps):
            step(gather)
        sync()
        el = time.perf_counter() - t0
        wait = wait_acc[0]
        if world > 1:
            t = torch.tensor([el, wait], dtype=torch.float64, device="cpu" if same_gpu else "cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el, wait = float(t[0].item()), float(t[1].item())
        return el, wait

    # The driver's protocol exactly: W warm-up steps, K timed steps -> `value`.  (N > 1: the same K steps with the
    # gather off are timed first, so that the line shows what the collective costs.)
    for _ in range(a.warmup):
        step()
    compute_only = None
    if world > 1:
        compute_only, _ = timed(a.steps, gather=False)
    elapsed, gather_wait = timed(a.steps)
    # once more behind `--prewarm` further launches (a GPU coming from idle needs ~150 launches to reach its steady clock):
    # reported beside the value, never as the value
    steady = None
    if a.prewarm > 0:
        for _ in range(a.prewarm):
            compute()
        for _ in range(a.warmup):
            step()
        steady, _ = timed(a.steps)
    # --reserve-cus auto: a step that costs more than 1.15 x its compute is waiting for the collective (or the collective
    # for CUs: the persistent kernels fill every CU); one re-timing with 32 CUs left to RCCL, both printed
    retry = None
    if world > 1 and reserve_auto and not dry:
        ref_el = steady if steady is not None else elapsed
        if ref_el > 1.15 * compute_only:
            ops.set_reserved_cus(32)
            for _ in range(a.warmup):
                step()
            r_el, r_wait = timed(a.steps)
            ops.set_reserved_cus(0)
            retry = {"reserved_cus": 32, "ms_per_step": round(r_el / a.steps * 1e3, 4),
                     "gather_wait_ms": round(r_wait / a.steps * 1e3, 4),
                     "value": round(n_total * L * a.steps / r_el / 1e6, 1)}
    ranks_seen = None
    if world > 1:
        names = [None] * world
        dist.all_gather_object(names, "cpu (dry run)" if dry else f"cuda:{dev_index} {torch.cuda.get_device_name(dev_index)}")
        ranks_seen = {"world_size": dist.get_world_size(), "backend": backend, "devices": names}

    # dominant kernel: HIP events on the stream it is launched on, back-to-back launches (steady clock: the timed
    # steps above have just run); the average includes the ~2 us dispatch gap, rocprofv3's per-dispatch average does not
    roof = None
    if rank == 0 and not dry:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 200                                         # (>= 30 ms of back-to-back launches: the steady clock)
        if a.config == "c2" and one_launch:
            segments = ops.mel_config(SR, N_FFT, N_MELS, waves=16).segtab is not None
            kname = ("stft2048_kernel<16,2,6> (16 waves, staged tiles, per-wave mel projection by segment sums, clip-resident "
                     "MFCC; the whole step)" if segments else
                     "stft2048_kernel<16,2,3> (16 waves, staged tiles, clip-resident MFCC; the whole step)")
            kfn = lambda: ops.stft2048_mfcc(y, SR, HOP, True, "hann", N_MELS, N_MFCC)
            kbytes = B * algo_bytes_per_clip(N_MFCC)
        elif a.config == "c2":
            kname = f"stft2048_kernel<16,2,0> (16 waves, staged tiles, mel out) on all but {reserve} CUs"
            kfn = lambda: ops.stft2048_mel(y, SR, HOP, True, "hann", 2048, N_MELS)
            kbytes = B * (4 * L + 4 * N_MELS * T_FRAMES)
        else:
            from sygnals_amd.core.features.manager import feature_block_dominant
            kname, kfn, krows = feature_block_dominant(y, SR, HOP, N_MELS, N_MFCC)
            kbytes = B * (4 * L + 4 * krows * T_FRAMES)
        for _ in range(200 if steady is None else 20):     # (without the steady_clock leg: bring the clock up here)
            kfn()
        e0.record()
        for _ in range(reps):
            kfn()
        e1.record()
        e1.synchronize()
        kdur = e0.elapsed_time(e1) * 1e-3 / reps
        achieved = kbytes / kdur / 1e9
        # traffic / binding: from the committed rocprofv3 PMC passes of this kernel (profiles/traffic.json says which
        # file); PMC collection needs the profiler, so it is not re-measured inside this run
        traffic = binding = None
        tp = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tp) and a.config == "c2" and one_launch:
            try:
                tj = json.load(open(tp))
                traffic = tj.get("dominant_kernel_bytes_per_launch")
                binding = tj.get("binding")
            except Exception:
                traffic = binding = None
        roof = {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                "traffic_source": "profiles/traffic.json (committed rocprofv3 --pmc passes; not re-measured in this run)" if traffic else None,
                "binding": binding, "kernel": kname,
                "kernel_avg_us": round(kdur * 1e6, 2), "kernel_launches_timed": reps,
                "algorithmic_bytes_per_launch": kbytes}
        if a.config == "c2" and one_launch:
            fl = mfma_flops_per_launch(ops, B, segments)
            roof["mfma_util"] = {"flops_per_launch": fl, "achieved_tflops": round(fl / kdur / 1e12, 2),
                                 "peak_tflops": MFMA_F32_PEAK_TFLOPS,
                                 "dtype": "f32 (v_mfma_f32_16x16x4_f32 DCT" + (")" if segments else ", v_mfma_f32_4x4x1_16b_f32 filterbank)"),
                                 "frac": round(fl / kdur / 1e12 / MFMA_F32_PEAK_TFLOPS, 4),
                                 "note": ("dB/DCT epilogue only: FFT and mel projection (segment sums) run on the vector pipe; " if segments else
                                          "mel filterbank (block-sparse, four-row groups) + dB/DCT epilogue; the FFT runs on the vector pipe, and ")
                                         + "fp32 MFMA shares the SIMD's fp32 lanes with the vector pipe (tools/ubench/mfma_valu_coexec.hip), "
                                         "so a low figure here is the goal, not a shortfall"}

    if rank == 0:
        samples = n_total * L * a.steps
        value = samples / elapsed / 1e6
        job_bytes = n_total * algo_bytes_per_clip(rows) * a.steps
        if a.config == "c2":
            workload = ("C2: 1024 x 1 s @ 48 kHz clips per GPU, STFT n_fft=2048 hop=512 hann center -> 40-band Slaney "
                        "mel -> power_to_db(ref=max, top_db=80) -> 13 MFCC (DCT-II ortho)")
        else:
            workload = ("C4: 2048 x 1 s @ 48 kHz clips per GPU (16 384 over 8), STFT n_fft=2048 hop=512 -> 13 MFCC "
                        "(40 mel) + spectral centroid + rolloff (0.85) + contrast (6 bands + delta) -> [B, 22, 94] "
                        "block per step, gathered to rank 0")
        par = f"clip-sharded x{world}"
        if world > 1:
            par += (f", asynchronous gather to rank 0 in the timed region ({backend})" +
                    (f", two-launch form on all but {reserve} CUs" if reserve else ", same kernels as N = 1"))
        if dry:
            value = 0.0
        line = {
            "metric": "Msamples/s STFT->MFCC (n_fft=2048, hop=512)", "value": round(value, 1), "unit": "Msamples/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(elapsed / a.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic" if not dry else "DRY RUN: launcher rehearsal without a GPU, zeros instead of kernels, no measurement",
            "steady_clock": None if steady is None or dry else {
                "value": round(samples / steady / 1e6, 1), "ms_per_step": round(steady / a.steps * 1e3, 4),
                "launches_before": a.prewarm + a.warmup,
                "note": "the same K steps timed again behind further launches (GPU at its steady clock); not the value"},
            "compute_only_ms": None if compute_only is None else round(compute_only / a.steps * 1e3, 4),
            "gather_wait_ms": None if world == 1 else round(gather_wait / a.steps * 1e3, 4),
            "ranks_seen": ranks_seen, "reserve_cus_retry": retry,
            "config": {"workload": workload, "clips_per_gpu": B, "clip_samples": L, "sr": SR, "n_fft": N_FFT, "hop": HOP,
                       "n_mels": N_MELS, "n_mfcc": N_MFCC, "rows_per_clip": rows, "parallelism": par},
            "hbm_roofline_frac_whole_step": round(job_bytes / elapsed / 1e9 / (HBM_PEAK_GBS * world), 5),
            "roofline": roof, "cpu_baseline": cpu,
        }
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
