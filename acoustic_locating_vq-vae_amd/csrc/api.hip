// Error reporting + version for libalvq.
#include <stdarg.h>

#include "alvq_common.h"

namespace alvq {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
}  // namespace alvq

extern "C" const char* alvq_version(void) { return "alvq 0.1.0 (gfx950)"; }
extern "C" const char* alvq_last_error(void) { return alvq::g_err; }
