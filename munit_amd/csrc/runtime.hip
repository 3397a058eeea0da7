// Error reporting + version for libmunit_hip.so.
#include "common.h"
#include <cstring>

namespace {
thread_local char g_err[512] = "";
}

void munit_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* munit_last_error(void) { return g_err; }
extern "C" int munit_version(void) { return 1; }
