// api.hip -- C-ABI plumbing shared by every entry point of libsigman_gsplat.so (error string, version).
#include <stdarg.h>
#include <stdio.h>

#include "common.h"

static thread_local char g_err[512] = "";

void sgr_set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char *sgr_last_error(void) { return g_err; }
extern "C" int sgr_abi_version(void) { return SGR_ABI_VERSION; }
