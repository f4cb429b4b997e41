// Library-wide bits of librecattend.so: version and the thread-local last-error string.
#include <cstdarg>
#include <cstdio>

#include "ra_common.h"

namespace ra {
namespace {
thread_local char g_err[512] = "";
}
void set_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
}  // namespace ra

extern "C" int ra_version(void) { return 100; }
extern "C" const char *ra_last_error_string(void) { return ra::g_err; }
