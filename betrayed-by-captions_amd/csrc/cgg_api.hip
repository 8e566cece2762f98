// Library-level entry points of libcgg_hip.so: version + thread-local error string.
#include "cgg_common.h"

static thread_local char g_cgg_err[512] = "";

void cgg_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_cgg_err, sizeof(g_cgg_err), fmt, ap);
  va_end(ap);
}

extern "C" int cgg_version(void) { return CGG_VERSION; }

extern "C" const char* cgg_last_error_string(void) { return g_cgg_err; }
