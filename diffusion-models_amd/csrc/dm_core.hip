// Error reporting shared by every entry point of libdm_hip.so.
#include "dm_common.h"
#include <string.h>

static thread_local char g_err[512] = "";

void dm_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* dm_last_error(void) { return g_err; }
extern "C" int dm_abi_version(void) { return 1; }
