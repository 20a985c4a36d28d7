#!/bin/bash
# Run on the GPU box: PMC counters of the dominant kernels of the BASELINE configurations other than C2 (C3: sosfiltfilt +
# MFCC; C5: Welch, CQT decimation chain and octave products; the n_fft = 1024 fused kernel).  One rocprofv3 pass per counter
# set (gpurun refuses --pmc together with the trace domains); the kernel-trace pass is separate.
#   tools/profile_rows.sh [outdir]
set -u
cd "$(dirname "$0")/.."
OUT=${1:-gpurun_out/rows_pmc}
mkdir -p "$OUT"
export TMPDIR=/tmp
S1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY"
S2="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_WR SQ_WAVES"
S3="FETCH_SIZE GRBM_GUI_ACTIVE"
S4="WRITE_SIZE SQ_INSTS_VMEM_RD SQ_VALU_MFMA_BUSY_CYCLES"
ROWS='"C3 filtfilt" "C5 a14" "C5 a15" "n_fft=1024 hop=256 (mfcc"'
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 tools/row_bench.py "C3 filtfilt" "C5 a14" "C5 a15" "n_fft=1024 hop=256 (mfcc" > "$OUT/trace.log" 2>&1
echo "trace rc=$?"
i=0
for S in "$S1" "$S2" "$S3" "$S4"; do
  i=$((i+1))
  rocprofv3 --pmc $S --output-format csv -d "$OUT/pmc$i" -- python3 tools/row_bench.py "C3 filtfilt" "C5 a14" "C5 a15" "n_fft=1024 hop=256 (mfcc" > "$OUT/pmc$i.log" 2>&1
  echo "pmc$i rc=$?"
done
{
  echo "== kernel-trace stats (row_bench.py rows C3, C5 Welch, C5 CQT, n_fft 1024) =="
  f=$(ls $OUT/trace/*/*kernel_stats.csv 2>/dev/null | head -1)
  [ -n "$f" ] && head -25 "$f"
  echo "== PMC, mean per dispatch =="
  for i in 1 2 3 4; do python3 tools/pmc_kernels.py "$OUT/pmc$i" sos_clip welch_wave cqt_fused decimate2_chain cqt_bf16x3 cqt_gemm stft_mel_w1024 stft2048_kernel logmel 2>/dev/null; done
} > "$OUT/summary.txt"
tail -5 "$OUT/summary.txt"
