#!/bin/bash
# Build libsygnals_hip.so for gfx950 (MI355X).  hipcc cross-compiles without a GPU.
set -e
cd "$(dirname "$0")"
mkdir -p sygnals_amd/lib
SRC="sygnals_amd/csrc/capi.hip sygnals_amd/csrc/stft_mel.hip sygnals_amd/csrc/logmel_dct.hip sygnals_amd/csrc/fft_generic.hip sygnals_amd/csrc/spectral.hip sygnals_amd/csrc/sosfilt.hip sygnals_amd/csrc/cqt.hip"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -shared -fno-strict-aliasing -fno-slp-vectorize -std=c++17 \
  -Wno-pass-failed $EXTRA_HIPCC_FLAGS $SRC -o sygnals_amd/lib/libsygnals_hip.so
echo "built sygnals_amd/lib/libsygnals_hip.so"
