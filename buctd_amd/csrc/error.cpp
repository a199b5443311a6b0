// Error channel + version of libbuctd_hip.so. One message slot per thread: entry points are
// re-entrant and keep no other global mutable state.
#include <stdarg.h>
#include <stdio.h>

static thread_local char g_err[512] = "";

void buctd_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* buctd_last_error(void) { return g_err; }
extern "C" int buctd_version(void) { return 100; /* 0.1.0: round 1 */ }
