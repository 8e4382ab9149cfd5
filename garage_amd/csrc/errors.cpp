// Thread-local error string behind the C ABI (include/garage_amd.h).
#include <stdarg.h>
#include <stdio.h>

static thread_local char g_err[512] = "";

void ga_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* ga_last_error(void) { return g_err; }

extern "C" int ga_abi_version(void) { return 4; }
