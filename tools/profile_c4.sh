#!/bin/bash
# Run on the GPU box: PMC counters of the fused C4 kernel, one rocprofv3 pass per (variant, counter set).
#   tools/profile_c4.sh [outdir] [variants...]
set -u
cd "$(dirname "$0")/.."
OUT=${1:-gpurun_out/c4_pmc}
shift || true
VARS=${*:-"mel stats contrast c4"}
mkdir -p "$OUT"
export TMPDIR=/tmp
S1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY"
S2="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_WR SQ_WAVES"
S3="SQ_IFETCH SQ_WAIT_INST_LDS SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT SQ_INSTS_BRANCH SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE"
SETS=${SETS:-"1 2 3"}
for v in $VARS; do
  i=0
  for S in "$S1" "$S2" "$S3"; do
    i=$((i+1))
    case " $SETS " in *" $i "*) ;; *) continue ;; esac
    rocprofv3 --pmc $S --output-format csv -d "$OUT/$v.$i" -- python3 tools/c4_pmc.py $v 20 > "$OUT/$v.$i.log" 2>&1
    echo "$v set $i rc=$?"
  done
done
for v in $VARS; do
  echo "=== $v ==="
  for i in $SETS; do python3 tools/pmc_kernels.py "$OUT/$v.$i" stft2048 2>/dev/null; done
done > "$OUT/summary.txt"
cat "$OUT/summary.txt"
