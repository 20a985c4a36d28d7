#!/bin/bash
# round-4 GPU batch C: full GPU suite, the C4 fixed-cost evidence (breakdown + per-phase stamps, with and without the
# per-frame lane constants), bench lines, the frame-length-1024 rows
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/r04c; mkdir -p $OUT
python -m pytest tests -m gpu -x -q > $OUT/gputest.txt 2>&1; echo "gputest rc=$?"; tail -3 $OUT/gputest.txt
python3 tools/c4_breakdown.py > $OUT/c4_breakdown_after.txt 2>&1; echo "breakdown rc=$?"
SYGNALS_AMD_LIB=build/dev/libsyg_norelc.so python3 tools/c4_breakdown.py > $OUT/c4_breakdown_before.txt 2>&1; echo "breakdown(before) rc=$?"
for v in c4one c4one_band0 c4one_cen; do
  SYGNALS_AMD_LIB=build/dev/libsyg_dev.so SYGNALS_AMD_ALLOW_VARIANT=9 python3 tools/timeline.py $v > $OUT/timeline_${v}_after.txt 2>&1; echo "timeline $v rc=$?"
  SYGNALS_AMD_LIB=build/dev/libsyg_dev_norelc.so SYGNALS_AMD_ALLOW_VARIANT=9 python3 tools/timeline.py $v > $OUT/timeline_${v}_before.txt 2>&1; echo "timeline(before) $v rc=$?"
done
timeout -k 10 400 python3 bench.py --config c4 --no-cpu-baseline > $OUT/bench_c4.json 2> $OUT/bench_c4.err; echo "bench c4 rc=$?"
timeout -k 10 400 python3 bench.py --steps 20 --warmup 5 > $OUT/bench_driver.json 2> $OUT/bench_driver.err; echo "bench driver rc=$?"
ROWS_OUT=$OUT/rows.json python3 tools/row_bench.py "n_fft=1024" "C4-style" "C4 share" "C2 a1" > $OUT/rows.log 2>&1; echo "rows rc=$?"
cut -c1-260 $OUT/rows.log
cat $OUT/bench_c4.json | cut -c1-600
