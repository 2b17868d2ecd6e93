// api.cpp -- error plumbing of the C ABI (include/unislam_hip.h); no C++ exception crosses the boundary
#include "us_common.h"
#include <string.h>

static thread_local char g_err[512] = "";

void us_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* us_last_error(void) { return g_err; }
extern "C" int us_abi_version(void) { return US_ABI_VERSION; }
