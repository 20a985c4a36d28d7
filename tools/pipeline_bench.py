"""End-to-end (PCIe-inclusive) rate of the ingest pipeline: pinned host PCM -> HBM -> pcm_to_f32 -> one-launch MFCC ->
results back on the host, for 16-bit PCM and float32 batches of the C2 shape (1024 clips x 48000).  Never `value`
of bench.py (that one starts with the clips resident in HBM); the numbers go to DESIGN.md section 5."""
import json, sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from sygnals_amd import ops
from sygnals_amd.pipeline import DevicePipeline

B, L, SR, NB = 1024, 48000, 48000, 40
rng = np.random.default_rng(0)
pcm = rng.integers(-20000, 20000, (B, L), dtype=np.int16)
f32 = (pcm.astype(np.float32) / 32768.0)
compute = lambda x: ops.mfcc_batch(x, SR, n_mels=40)
from concurrent.futures import ThreadPoolExecutor
pool = ThreadPoolExecutor(16)
out = {}
for name, host in (("pcm16", torch.from_numpy(pcm)), ("float32", torch.from_numpy(f32))):
    for depth in (1, 2, 3):
        pipe = DevicePipeline(compute, depth=depth)
        def fill(buf, h=host):                              # 16 host threads write the pinned slot, as decoders would
            list(pool.map(lambda i: buf[i * 64:(i + 1) * 64].copy_(h[i * 64:(i + 1) * 64]), range(B // 64)))
        list(pipe.run([(fill, host.shape, host.dtype)] * 4))   # warm-up: allocations, tables
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = sum(1 for _ in pipe.run([(fill, host.shape, host.dtype)] * NB))
        dt = time.perf_counter() - t0
        r = {"ms_per_batch": round(dt / NB * 1e3, 3), "Msamples_per_s": round(NB * B * L / dt / 1e6, 1),
             "h2d_GBps": round(NB * host.numel() * host.element_size() / dt / 1e9, 2)}
        out[f"{name}_depth{depth}"] = r
        print(name, "depth", depth, json.dumps(r), flush=True)
    # the link alone: pinned -> device copies back to back
    pin = host.pin_memory(); dev = torch.empty_like(host, device="cuda")
    for _ in range(3): dev.copy_(pin, non_blocking=True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): dev.copy_(pin, non_blocking=True)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    out[f"{name}_h2d_only"] = {"GBps": round(20 * host.numel() * host.element_size() / dt / 1e9, 2)}
    print(name, "H2D only", out[f"{name}_h2d_only"], flush=True)
    # without the host memcpy into the slot (data already pinned): upper bound of the overlapped pipeline
    pipe = DevicePipeline(compute, depth=2)
    nofill = lambda buf: None
    list(pipe.run([(nofill, host.shape, host.dtype)] * 4)); torch.cuda.synchronize()
    t0 = time.perf_counter(); list(pipe.run([(nofill, host.shape, host.dtype)] * NB)); dt = time.perf_counter() - t0
    out[f"{name}_prepinned"] = {"ms_per_batch": round(dt / NB * 1e3, 3), "Msamples_per_s": round(NB * B * L / dt / 1e6, 1)}
    print(name, "pre-pinned", out[f"{name}_prepinned"], flush=True)
json.dump(out, open("gpurun_out/pipeline_r01.json", "w"), indent=1)
