#!/bin/bash
# Round evidence on the GPU box: headline trace + PMC (tools/profile.sh), C4 PMC, every row, the bench lines, and the
# two-rank same-GPU rehearsal of `python3 bench.py --gpus 2` (gloo exchange; RCCL refuses two ranks on one device).
#   tools/round_evidence.sh <tag>      -> gpurun_out/<tag>/
set -u
cd "$(dirname "$0")/.."
TAG=${1:-r04}
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
python3 bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"; echo "bench rc=$?"
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > "$OUT/bench_driver_flags.json" 2>> "$OUT/bench.err"; echo "bench driver flags rc=$?"
python3 bench.py --config c4 --no-cpu-baseline > "$OUT/bench_c4.json" 2>> "$OUT/bench.err"; echo "bench c4 rc=$?"
SYG_BENCH_SAME_GPU=1 python3 bench.py --gpus 2 --steps 50 --warmup 10 > "$OUT/rehearsal_2ranks_c2.log" 2>&1; echo "rehearsal c2 rc=$?"
SYG_BENCH_SAME_GPU=1 python3 bench.py --gpus 2 --config c4 --steps 20 --warmup 5 --no-cpu-baseline > "$OUT/rehearsal_2ranks_c4.log" 2>&1; echo "rehearsal c4 rc=$?"
ROWS_OUT=$OUT/rows.json python3 tools/row_bench.py > "$OUT/rows.log" 2>&1; echo "rows rc=$?"
tools/profile.sh "$OUT/prof" > "$OUT/prof.log" 2>&1; echo "profile rc=$?"
SETS="1 2" tools/profile_c4.sh "$OUT/c4_pmc" mfcc c4one > "$OUT/c4_pmc.log" 2>&1; echo "c4 pmc rc=$?"
python3 tools/c4_probe.py > "$OUT/c4_probe.txt" 2>&1; echo "c4 probe rc=$?"
