"""`python bench.py --gpus N` must run by itself for N > 1 (the driver's invocation shape): the parent starts one rank per
GPU under torch.distributed.run as a child process, hands the CPU baseline through the environment and relays rank 0's
JSON line.  Rehearsed here without a GPU (SYG_BENCH_DRY=1: gloo ranks, zeros instead of kernels -- the line says so)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra_env, *args):
    env = dict(os.environ, SYG_BENCH_DRY="1", **extra_env)
    env.pop("WORLD_SIZE", None); env.pop("RANK", None); env.pop("LOCAL_RANK", None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], env=env, capture_output=True, text=True,
                       timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout
    return json.loads(lines[0])


def test_plain_invocation_with_two_gpus_launches_its_own_ranks():
    cpu = {"value": 1.0, "unit": "Msamples/s", "cores": 1, "kind": "port", "sample": "stand-in handed to the launcher"}
    line = _run({"SYG_BENCH_CPU_BASELINE": json.dumps(cpu)}, "--gpus", "2", "--steps", "3", "--warmup", "1", "--prewarm", "2",
                "--clips", "8")
    assert line["n_gpus"] == 2 and line["steps"] == 3 and line["warmup"] == 1
    assert line["scaling"] == "weak" and line["unit"] == "Msamples/s"
    assert line["cpu_baseline"] == cpu                       # non-null for N > 1: the parent's baseline reaches rank 0
    assert line["data"].startswith("DRY RUN")                # and a rehearsal can never pass for a measurement
    assert "x2" in line["config"]["parallelism"] and "RCCL" not in line["config"]["parallelism"]   # (gloo rehearsal)
    # the fields the first multi-GPU hardware run diagnoses itself with
    assert line["compute_only_ms"] is not None and line["compute_only_ms"] >= 0
    assert line["gather_wait_ms"] is not None and line["gather_wait_ms"] >= 0
    rs = line["ranks_seen"]
    assert rs["world_size"] == 2 and len(rs["devices"]) == 2 and rs["backend"].startswith("gloo")
    assert line["reserve_cus_retry"] is None
    assert "prewarm_steps" not in line and "no_prewarm" not in line      # `value` is the driver's protocol, nothing ahead of it


def test_c4_payload_through_the_launcher():
    line = _run({}, "--gpus", "2", "--config", "c4", "--steps", "2", "--warmup", "1", "--prewarm", "0", "--clips", "4",
                "--no-cpu-baseline")
    assert line["n_gpus"] == 2 and line["config"]["rows_per_clip"] == 22 and line["cpu_baseline"] is None
    assert line["steady_clock"] is None and line["ranks_seen"]["world_size"] == 2


def test_single_gpu_dry_line():
    line = _run({}, "--steps", "2", "--warmup", "1", "--prewarm", "0", "--clips", "4", "--no-cpu-baseline")
    assert line["n_gpus"] == 1 and line["ms_per_step"] >= 0
    assert line["compute_only_ms"] is None and line["gather_wait_ms"] is None and line["ranks_seen"] is None


def test_reserve_cus_auto_is_accepted():
    line = _run({}, "--gpus", "2", "--steps", "2", "--warmup", "1", "--prewarm", "0", "--clips", "4", "--no-cpu-baseline",
                "--reserve-cus", "auto")
    assert line["n_gpus"] == 2 and line["reserve_cus_retry"] is None     # (a dry run never re-times)
