#!/bin/bash
# Build ablation variants of the fused kernel on the GPU box and time each (outputs are WRONG by design).
# Every variant goes to its own library under build/ (never the product path) and is loaded through
# SYGNALS_AMD_LIB + SYGNALS_AMD_ALLOW_VARIANT; an interrupted run leaves the product library untouched.
cd "$(dirname "$0")/.."
for abl in 0 1 2 3 4; do
  out="build/ablate/libsygnals_hip_abl$abl.so"
  EXTRA_HIPCC_FLAGS="-DSYG_ABL=$abl" SYG_LIB_OUT="$out" ./build_lib.sh > /dev/null 2>&1
  echo "== SYG_ABL=$abl (0 base, 1 no frame/window loads, 2 no LDS exchanges, 3 no LDS tw1 table, 4 = 2+3)"
  SYGNALS_AMD_LIB="$PWD/$out" SYGNALS_AMD_ALLOW_VARIANT=$abl SYGNALS_AMD_WAVES=${W:-16} \
    timeout -k 10 100 python tools/quick_bench.py 1024 2>&1 | grep -E "^rep 2|fft-only"
done
