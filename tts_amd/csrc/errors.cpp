// Thread-local last-error text for the C ABI.
#include <stdarg.h>
#include <stdio.h>

#include "../../include/ttsx.h"

static thread_local char g_err[512] = "no error";

void ttsx_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* ttsx_last_error_string(void) { return g_err; }
extern "C" int ttsx_version(void) { return 1; }
