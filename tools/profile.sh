#!/bin/bash
# Run on the GPU box (via gpurun): kernel-trace stats + PMC counters for the headline bench.
# Each PMC set is its own run (gpurun refuses --pmc combined with the trace domains).
#   tools/profile.sh [outdir]
set -u
cd "$(dirname "$0")/.."
OUT=${1:-gpurun_out/prof_r02}
mkdir -p "$OUT"
export TMPDIR=/tmp
CMD="python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline"
# the trace pass runs the default step counts (300 pre-warm + 50 warm-up + 500 timed + 100 for the roofline block): its
# per-dispatch average is then dominated by steady-clock launches, like the figure bench.py prints
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 bench.py --no-cpu-baseline > "$OUT/trace.log" 2>&1
echo "trace rc=$?"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY --output-format csv -d "$OUT/pmc1" -- $CMD > "$OUT/pmc1.log" 2>&1
echo "pmc1 rc=$?"
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_MFMA SQ_WAVES --output-format csv -d "$OUT/pmc2" -- $CMD > "$OUT/pmc2.log" 2>&1
echo "pmc2 rc=$?"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc3" -- $CMD > "$OUT/pmc3.log" 2>&1
echo "pmc3 rc=$?"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc4" -- $CMD > "$OUT/pmc4.log" 2>&1
echo "pmc4 rc=$?"
rocprofv3 --pmc GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d "$OUT/pmc5" -- $CMD > "$OUT/pmc5.log" 2>&1
echo "pmc5 rc=$?"
python3 tools/summarize_prof.py "$OUT" > "$OUT/summary.txt" 2>&1
cat "$OUT/summary.txt"
