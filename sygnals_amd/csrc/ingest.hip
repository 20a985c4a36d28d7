// Batched audio ingest on the device: integer PCM frames as they sit in a WAV file -> float32 mono clips.
//
// Replaces, per clip, the host-side conversion the reference reaches through sygnals/core/audio/io.py:84-90
// (librosa.load -> soundfile: integer PCM scaled by 2^-(bits-1), channels averaged for mono=True) and the mix-down of
// sygnals/cli/features_cmd.py:66-68 (np.mean over channels).  Moving the conversion behind the host-to-device copy
// halves the PCIe bytes for 16-bit audio (the copy, not the kernels, bounds the end-to-end rate: DESIGN.md section 5).
// Pure streaming: each thread converts four consecutive frames (a 16-byte float4 store).
#include "common.h"

namespace syg {
namespace {

template <typename T> struct Pcm;
template <> struct Pcm<int16_t> { static __device__ int cvt(int16_t v) { return (int)v; } };
template <> struct Pcm<int32_t> { static __device__ int cvt(int32_t v) { return v; } };
template <> struct Pcm<uint8_t> { static __device__ int cvt(uint8_t v) { return (int)v - 128; } };

// out[r, i] = (sum_c pcm[r, i, c]) * scale / C   (interleaved channels; row r starts at pcm + r * ld elements).
// One channel: int -> float -> * 2^-(bits-1), a single rounding (none below 25 bits).  Several channels: the integer
// sum is exact, the mean is formed in float64 and rounded to float32 once -- bit-identical to the host's float64
// conversion + np.mean + astype(float32); the float64 division is irrelevant next to the PCIe copy in front of it.
template <typename T>
__global__ __launch_bounds__(256) void pcm_to_f32_kernel(const T* __restrict__ pcm, int64_t frames, int channels,
                                                         int64_t ld, double scale, float* __restrict__ out,
                                                         int64_t ldo) {
  const int64_t r = blockIdx.y;
  const T* p = pcm + r * ld;
  float* o = out + r * ldo;
  const float fscale = (float)scale;
  const double dc = (double)channels;
  for (int64_t i = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) * 4; i < frames;
       i += (int64_t)gridDim.x * blockDim.x * 4) {
    float v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      v[j] = 0.f;
      if (i + j < frames) {
        if (channels == 1) {
          v[j] = (float)Pcm<T>::cvt(p[i + j]) * fscale;
        } else {
          long long s = 0;
          for (int c = 0; c < channels; ++c) s += Pcm<T>::cvt(p[(i + j) * channels + c]);
          v[j] = (float)(((double)s * scale) / dc);
        }
      }
    }
    if (i + 3 < frames && ((ldo & 3) == 0)) {
      *reinterpret_cast<float4*>(o + i) = make_float4(v[0], v[1], v[2], v[3]);
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (i + j < frames) o[i + j] = v[j];
    }
  }
}

template <typename T>
int launch_pcm(const void* pcm, int64_t rows, int64_t frames, int channels, int64_t ld, double scale, float* out,
               int64_t ldo, hipStream_t stream) {
  int64_t bx = (frames + 1023) / 1024;
  const int64_t cap = rows >= 64 ? 64 : 4096 / rows;
  if (bx > cap) bx = cap;
  if (bx < 1) bx = 1;
  hipLaunchKernelGGL(pcm_to_f32_kernel<T>, dim3((unsigned)bx, (unsigned)rows), dim3(256), 0, stream, (const T*)pcm,
                     frames, channels, ld, scale, out, ldo);
  SYG_CHECK_LAUNCH("pcm_to_f32");
  return SYG_OK;
}

}  // namespace
}  // namespace syg

using namespace syg;

extern "C" int syg_pcm_to_f32(const void* pcm, int bits, int64_t rows, int64_t frames, int channels, int64_t ld,
                              float* out, int64_t ldo, void* stream) {
  SYG_REQUIRE(pcm && out, "pcm_to_f32: null pointer argument");
  SYG_REQUIRE(bits == 8 || bits == 16 || bits == 32, "pcm_to_f32: bits must be 8 (unsigned), 16 or 32 (got %d)", bits);
  SYG_REQUIRE(rows >= 1 && rows <= 65535 && frames >= 1 && channels >= 1 && channels <= 256,
              "pcm_to_f32: need 1 <= rows <= 65535, frames >= 1, 1 <= channels <= 256");
  SYG_REQUIRE(ld >= frames * channels && ldo >= frames, "pcm_to_f32: row strides shorter than a row");
  SYG_REQUIRE(((uintptr_t)out & 15) == 0, "pcm_to_f32: out must be 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  if (bits == 16) return launch_pcm<int16_t>(pcm, rows, frames, channels, ld, 1.0 / 32768.0, out, ldo, st);
  if (bits == 32) return launch_pcm<int32_t>(pcm, rows, frames, channels, ld, 1.0 / 2147483648.0, out, ldo, st);
  return launch_pcm<uint8_t>(pcm, rows, frames, channels, ld, 1.0 / 128.0, out, ldo, st);
}
