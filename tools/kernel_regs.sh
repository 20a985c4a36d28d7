#!/bin/bash
# Register / spill / scratch figures of the kernels of one source file (device-only compile with the product flags).
#   tools/kernel_regs.sh sygnals_amd/csrc/stft_mel.hip [name-substring] [extra hipcc flags]
set -e
SRC=$1; PAT=${2:-}; shift; shift || true
TMP=$(mktemp -d)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fno-strict-aliasing -fno-slp-vectorize -std=c++17 -Xclang -target-feature -Xclang -load-store-opt \
  -Wno-pass-failed --cuda-device-only "$@" -c "$SRC" -o "$TMP/b.co" 2> >(grep -v "is not a recognized feature" >&2)
/opt/rocm/lib/llvm/bin/clang-offload-bundler --unbundle --type=o --input="$TMP/b.co" --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output="$TMP/k.co"
/opt/rocm/lib/llvm/bin/llvm-readelf --notes "$TMP/k.co" | PAT="$PAT" python3 -c "
import sys,re,os,subprocess
txt=sys.stdin.read()
for blk in txt.split('- .agpr_count')[1:]:
    name=re.search(r'\.name:\s+(\S+)',blk).group(1)
    try: dem=subprocess.run(['/opt/rocm/lib/llvm/bin/llvm-cxxfilt',name],capture_output=True,text=True).stdout.strip().split('(')[0]
    except Exception: dem=name
    pat=os.environ.get('PAT','')
    if pat and pat not in dem and pat not in name: continue
    g=lambda k: re.search(k+r':\s+(\d+)',blk).group(1)
    print(f'{dem[:70]:70s} vgpr {g(\".vgpr_count\"):>3s} vspill {g(\".vgpr_spill_count\"):>2s} sgpr {g(\".sgpr_count\"):>3s} sspill {g(\".sgpr_spill_count\"):>3s} scratch {g(\".private_segment_fixed_size\"):>4s} lds {g(\".group_segment_fixed_size\")}')
"
[ -n "$KEEP_CO" ] && cp "$TMP/k.co" "$KEEP_CO"
rm -rf "$TMP"
