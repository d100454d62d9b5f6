// Error reporting and build identification for libpgt_hip.so.
#include <stdarg.h>
#include <stdio.h>

#include "pgt_common.h"

static thread_local char g_err[512] = "";

void pgt_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" int pgt_abi_version(void) { return PGT_ABI_VERSION; }
extern "C" const char* pgt_last_error(void) { return g_err; }
extern "C" const char* pgt_build_target(void) { return PGT_TARGET; }
