// Error reporting + ABI version of libdehaze_hip.so (see include/dehaze_hip.h).
#include <stdarg.h>
#include "common.h"

static thread_local char g_err[512] = "";

void dhz_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* dhz_last_error(void) { return g_err; }
extern "C" int dhz_abi_version(void) { return 1; }
