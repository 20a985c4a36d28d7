"""What a kernel of another stream that holds a few CUs costs the MFCC step at C2, and what setting CUs aside buys.
tools/ubench/spin.hip keeps N workgroups of 512 threads busy for ~100 us on a second stream, started behind every
MFCC launch -- a stand-in for RCCL's send / receive workgroups while the previous batch's results are gathered.  A
workgroup of the fused kernels fills a CU, so the two kernels cannot share one: whichever is dispatched first makes
the other wait (rocprofv3 timeline: tools/queue_trace.py)."""
import ctypes as C, json, os, subprocess, sys
import numpy as np, torch
sys.path.insert(0, ".")
from sygnals_amd import ops
from oracle import cpu_ref as O
from tools.row_bench_util import timeit

so = "/tmp/libspin.so"
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-fPIC", "-shared", "tools/ubench/spin.hip", "-o", so], check=True)
spin = C.CDLL(so)
spin.spin_launch.argtypes = [C.c_int, C.c_longlong, C.c_void_p]
B, L, SR = 1024, 48000, 48000
y = ops.to_device_f32(np.tile(O.synth_clips(64, L, SR, seed=1), (B // 64, 1)))
for _ in range(300): ops.mfcc_batch(y, SR, n_mels=40)
side = torch.cuda.Stream()
res = {}
for busy in (0, 8, 16):
    for name, fused, reserve in (("one launch, all CUs", True, 0), ("two launches, all CUs", False, 0),
                                 ("two launches, 16 CUs set aside", False, 16), ("one launch, 16 CUs set aside", True, 16)):
        ops.set_reserved_cus(reserve)
        def step():
            if busy:
                side.wait_stream(torch.cuda.current_stream())
                spin.spin_launch(busy, 250000, C.c_void_p(side.cuda_stream))
            ops.mfcc_batch(y, SR, n_mels=40, fused=fused)
        t = timeit(step, 200, 50)
        torch.cuda.synchronize()
        res[f"{name}; {busy} CUs held by another kernel"] = round(t * 1e6, 1)
        print(f"{name:32s} {busy:3d} CUs held by another kernel: {t * 1e6:7.1f} us per step", flush=True)
ops.set_reserved_cus(0)
json.dump(res, open("gpurun_out/queue_r01.json", "w"), indent=1)
