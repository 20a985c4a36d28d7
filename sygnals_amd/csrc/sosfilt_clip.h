// Interface of sosfilt_clip.hip (the clip-resident zero-phase SOS filter) for sosfilt.hip's entry point.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace syg {

constexpr int SOSC_MAXS = 4;             // sections the clip-resident kernels are built for
constexpr int SOSC_MAXD = 2 * SOSC_MAXS;

struct SosClipParams {
  double b0[SOSC_MAXS], b1[SOSC_MAXS], b2[SOSC_MAXS], a1[SOSC_MAXS], a2[SOSC_MAXS];   // a0 = 1
  double zi[SOSC_MAXD];                  // sosfilt_zi, [z0, z1] per section
  double apow[SOSC_MAXD * SOSC_MAXD];    // A^chunk, row-major, SOSC_MAXD-strided
};

int sos_clip_chunk(int64_t lext);        // samples per lane for an extended length, 0 = does not fit
bool sos_clip_supported(int n_sections);
void sos_clip_launch(const float* x, int64_t B, int L, int64_t ldx, const SosClipParams& P, int n_sections, int cs, int pad,
                     float* y, int64_t ldy, hipStream_t st);

}  // namespace syg
