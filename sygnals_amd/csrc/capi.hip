// Error string + version entry points of the C ABI.
#include "common.h"
#include <string.h>

namespace syg {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
}  // namespace syg

#ifndef SYG_ABL
#define SYG_ABL 0
#endif
extern "C" int syg_abi_version(void) { return SYG_ABI_VERSION; }
// 0 for the product build; the SYG_ABL number of a development build (ablation / timeline variants compute WRONG
// results by design; build_lib.sh writes them to their own path and the Python binding refuses to load one)
#if defined(SYG_SOSC_ABL)
extern "C" int syg_build_variant(void) { return 100 + SYG_SOSC_ABL; }      // sosfilt_clip.hip timing ablations
#elif defined(SYG_TRIX)
extern "C" int syg_build_variant(void) { return 200 + SYG_TRIX; }          // stft_mel.hip MODE 6 timing experiments
#elif defined(SYG_R7ABL)
extern "C" int syg_build_variant(void) { return 300 + SYG_R7ABL; }         // stft_mel.hip MODE 7 row-function stand-ins
#else
extern "C" int syg_build_variant(void) { return SYG_ABL; }
#endif
extern "C" const char* syg_last_error(void) { return syg::g_err; }
