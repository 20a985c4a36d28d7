#!/bin/bash
# round-4 GPU batch D: A/B of the front-end forms, the persistent sosfiltfilt kernel, fused / sos tests
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/r04d; mkdir -p $OUT
python -m pytest tests/test_gpu_generic.py tests/test_gpu_fused.py tests/test_gpu_api.py -m gpu -x -q > $OUT/gputest.txt 2>&1; echo "gputest rc=$?"; tail -3 $OUT/gputest.txt
python3 tools/tri_forms_bench.py > $OUT/tri_forms.txt 2>&1; echo "forms rc=$?"
cat $OUT/tri_forms.txt
timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 > $OUT/bench_driver.json 2> $OUT/bench_driver.err; echo "bench driver rc=$?"
