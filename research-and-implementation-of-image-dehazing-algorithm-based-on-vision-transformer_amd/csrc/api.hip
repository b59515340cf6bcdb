// Error reporting + ABI version of libdehaze_hip.so (see include/dehaze_hip.h).
#include <stdarg.h>
#include "common.h"
#include "build_id.h"

static thread_local char g_err[512] = "";

void dhz_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int dhz_num_cus() {
    static thread_local int cached_dev = -1, cached = 256;          // per thread: no shared mutable state
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 256;
    if (dev != cached_dev) {
        int n = 0;
        cached = (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) ? n : 256;
        cached_dev = dev;
    }
    return cached;
}

extern "C" const char* dhz_last_error(void) { return g_err; }
extern "C" int dhz_abi_version(void) { return 1; }
#ifdef DHZ_VARIANT_TAG          // diagnostic builds (tools/variants.sh, tools/abl_fused.sh) link variant kernels: their id must not be the product's
extern "C" const char* dhz_build_id(void) { return DHZ_BUILD_ID "-" DHZ_VARIANT_TAG; }
#else
extern "C" const char* dhz_build_id(void) { return DHZ_BUILD_ID; }
#endif
