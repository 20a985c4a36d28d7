export TMPDIR=/tmp
timeout -k 10 300 python -m pytest tests/test_gpu_api.py -q -x -k "cqt or decimate" 2>&1 | tail -3
SYGNALS_AMD_LIB=$PWD/sygnals_amd/lib/variants/libsyg_stamp.so SYGNALS_AMD_ALLOW_VARIANT=1 timeout -k 10 200 python3 tools/cqt_stamp_probe.py 2>&1 | tail -6
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_w -- python3 tools/row_bench.py "C5 a15" > gpurun_out/prof_w.log 2>&1
grep -o '"ms": [0-9.]*' gpurun_out/prof_w.log; python3 tools/kernel_times.py gpurun_out/prof_w 7 staged bf16x3
