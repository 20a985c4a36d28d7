#!/bin/bash
# Build ablation variants of the fused kernel on the GPU box and time each (outputs are WRONG by design).
cd "$(dirname "$0")/.."
for abl in 0 1 2 3 4; do
  EXTRA_HIPCC_FLAGS="-DSYG_ABL=$abl" ./build_lib.sh > /dev/null 2>&1
  echo "== SYG_ABL=$abl (0 base, 1 no frame/window loads, 2 no LDS exchanges, 3 no LDS tw1 table, 4 = 2+3)"
  SYGNALS_AMD_WAVES=${W:-16} timeout -k 10 100 python tools/quick_bench.py 1024 2>&1 | grep -E "^rep 2|fft-only"
done
EXTRA_HIPCC_FLAGS="" ./build_lib.sh > /dev/null 2>&1
