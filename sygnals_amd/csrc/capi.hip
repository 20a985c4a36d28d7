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

extern "C" int syg_abi_version(void) { return SYG_ABI_VERSION; }
extern "C" const char* syg_last_error(void) { return syg::g_err; }
