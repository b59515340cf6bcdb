// Error reporting + ABI version of libdehaze_hip.so (see include/dehaze_hip.h).
#include <stdarg.h>
#include <atomic>
#include "common.h"
#include "build_id.h"

static thread_local char g_err[512] = "";

void dhz_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

// CUs that the persistent grids of this library leave to somebody else - RCCL's collective kernels when the gradient exchange overlaps
// the backward pass (one process per GPU; dehaze_hip.train.GradReducer sets it from DHZ_COMM_RESERVE_CUS when world > 1).  Every
// persistent grid is "resident workgroups per CU x dhz_num_cus()", so a reservation shrinks all of them at once; 0 (default) = the whole
// device.  Results do not depend on it (grid-stride loops), only the schedule does.
static std::atomic<int> g_reserved_cus{0};

static int physical_cus() {
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

int dhz_num_cus() {
    const int n = physical_cus() - g_reserved_cus.load(std::memory_order_relaxed);
    return n > 8 ? n : 8;
}

extern "C" int dhz_set_reserved_cus(int k) {
    DHZ_REQUIRE(k >= 0 && k < physical_cus(), "dhz_set_reserved_cus: k=%d (0 .. %d)", k, physical_cus() - 1);
    g_reserved_cus.store(k, std::memory_order_relaxed);
    return DHZ_OK;
}
extern "C" int dhz_get_reserved_cus(void) { return g_reserved_cus.load(std::memory_order_relaxed); }
extern "C" int dhz_grid_cus(void) { return dhz_num_cus(); }

extern "C" const char* dhz_last_error(void) { return g_err; }
extern "C" int dhz_abi_version(void) { return 1; }
#ifdef DHZ_VARIANT_TAG          // diagnostic builds (tools/variants.sh, tools/abl_fused.sh) link variant kernels: their id must not be the product's
extern "C" const char* dhz_build_id(void) { return DHZ_BUILD_ID "-" DHZ_VARIANT_TAG; }
#else
extern "C" const char* dhz_build_id(void) { return DHZ_BUILD_ID; }
#endif
