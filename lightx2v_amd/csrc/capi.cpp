// Error plumbing shared by every entry point of liblx2v_hip.so.
#include <stdarg.h>
#include <stdio.h>

#include "../../include/lx2v_hip.h"

static thread_local char g_err[512] = "";

void lx2v_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* lx2v_last_error(void) { return g_err; }

extern "C" int lx2v_abi_version(void) { return 1; }
