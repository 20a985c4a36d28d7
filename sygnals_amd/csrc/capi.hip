// Error string + version entry points of the C ABI.
#include "common.h"
#include <string.h>
#include <atomic>

namespace syg {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
// Process-wide options (syg_set_option): plain atomics, read where a launch is planned.  They select between kernels that
// the tests hold to the same results, or size the grid; nothing here is read from the environment.
static std::atomic<int> g_opt[SYG_OPT_COUNT] = {{0}, {-1}, {1}, {-1}};
int option(int key) { return (key >= 0 && key < SYG_OPT_COUNT) ? g_opt[key].load(std::memory_order_relaxed) : 0; }
}  // namespace syg

extern "C" int syg_set_option(int key, int value) {
  SYG_REQUIRE(key >= 0 && key < SYG_OPT_COUNT, "syg_set_option: unknown option %d", key);
  switch (key) {
    case SYG_OPT_RESERVED_CUS: SYG_REQUIRE(value >= 0 && value < 256, "syg_set_option: reserved CUs must be in [0, 256)"); break;
    case SYG_OPT_STFT_LOAD: SYG_REQUIRE(value >= -1 && value <= 2, "syg_set_option: frame load path must be -1 (default), 0, 1 or 2"); break;
    case SYG_OPT_SOS_CLIP: SYG_REQUIRE(value == 0 || value == 1, "syg_set_option: sos_clip must be 0 or 1"); break;
    case SYG_OPT_CQT_STAGED: SYG_REQUIRE(value >= -1 && value <= 2, "syg_set_option: cqt_staged must be -1 (default), 0, 1 or 2"); break;
  }
  syg::g_opt[key].store(value, std::memory_order_relaxed);
  return SYG_OK;
}
extern "C" int syg_get_option(int key) { return syg::option(key); }

extern "C" int syg_abi_version(void) { return SYG_ABI_VERSION; }
// 0 for the product build; non-zero for a development build (timeline / ablation variants whose results are wrong by
// design or whose outputs carry stamps; build_lib.sh writes them to their own path and the Python binding refuses one)
#if defined(SYG_SOSC_ABL)
extern "C" int syg_build_variant(void) { return 100 + SYG_SOSC_ABL; }      // sosfilt_clip.hip timing ablations
#elif defined(SYG_DEV) && SYG_DEV
extern "C" int syg_build_variant(void) { return 9; }                       // stft_mel.hip with per-phase stamps (stft_dev.h)
#else
extern "C" int syg_build_variant(void) { return 0; }
#endif
extern "C" const char* syg_last_error(void) { return syg::g_err; }
