#!/bin/bash
# Build libsygnals_hip.so for gfx950 (MI355X).  hipcc cross-compiles without a GPU.
#   SYG_LIB_OUT=<path>        output library (default: the product path sygnals_amd/lib/libsygnals_hip.so);
#                             development variants (ablations, timeline builds) MUST go somewhere else
#   EXTRA_HIPCC_FLAGS="..."   extra compiler flags (e.g. -DSYG_DEV=1: per-phase stamps; such a build reports itself through
#                             syg_build_variant() and sygnals_amd._lib refuses to load it as the product)
#   SYG_BUILD_JOBS=n          parallel compile jobs (default 8)
set -e
cd "$(dirname "$0")"
OUT="${SYG_LIB_OUT:-sygnals_amd/lib/libsygnals_hip.so}"
mkdir -p "$(dirname "$OUT")"
TAG=$(echo "$EXTRA_HIPCC_FLAGS" | md5sum | cut -c1-8)
OBJ="build/obj-$TAG"
mkdir -p "$OBJ"
SRCS="capi stft_mel stft_mel_pow2 stft_mel_w4096 stft_mel_w1024_seg stft_mel_wseg_small logmel_dct fft_generic spectral sosfilt sosfilt_clip cqt cqt_fused frame_stats dsp_extra ingest ml_utils fft_mixed welch_wave"
# -load-store-opt off (an AMDGPU feature; the host pass ignores it with a warning): keeps LDS accesses as single
# ds_read_b64 / ds_write_b64.  The merged forms (ds_read2_b64 ...) run at half the LDS rate on gfx950 and bank on 32
# instead of 64 dwords, which turns the conflict-free FFT exchange patterns into 2-way conflicts
# (MI355X_MICROARCH.md, LDS table).
FLAGS="--offload-arch=gfx950 -O3 -fPIC -fno-strict-aliasing -fno-slp-vectorize -std=c++17 -Xclang -target-feature -Xclang -load-store-opt -Wno-pass-failed $EXTRA_HIPCC_FLAGS"
compile_one() {
  s="sygnals_amd/csrc/$1.hip"; o="$OBJ/$1.o"
  [ -f "$s" ] || exit 0
  if [ ! -f "$o" ] || [ "$s" -nt "$o" ] || [ sygnals_amd/csrc/common.h -nt "$o" ] || [ sygnals_amd/csrc/mel_segments.h -nt "$o" ] || [ sygnals_amd/csrc/wave_fft.h -nt "$o" ] || [ sygnals_amd/csrc/row_features.h -nt "$o" ] || [ sygnals_amd/csrc/stft_dev.h -nt "$o" ] || [ include/sygnals_hip.h -nt "$o" ]; then
    /opt/rocm/bin/hipcc $FLAGS -c "$s" -o "$o" 2> >(grep -v "is not a recognized feature for this target" >&2)
  fi
}
export -f compile_one
export OBJ FLAGS
echo $SRCS | tr ' ' '\n' | xargs -P "${SYG_BUILD_JOBS:-8}" -I{} bash -c 'compile_one {}'
OBJS=""
for s in $SRCS; do [ -f "$OBJ/$s.o" ] && OBJS="$OBJS $OBJ/$s.o"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS -o "$OUT"
echo "built $OUT"
