// errors.cpp — thread-local error string + version for libdl3.
#include <stdarg.h>
#include <stdio.h>
#include "../../include/dl3.h"

static thread_local char g_err[512] = "";

void dl3_set_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char *dl3_last_error(void) { return g_err; }
extern "C" int dl3_version(void) { return 100; }
