// Gradient exchange of the data-parallel path as plain C entry points: thin wrappers over RCCL, for a host that binds
// only this library (the Python host reaches RCCL through torch.distributed("nccl") instead - same library, same collectives).
// RCCL is resolved at run time (dlopen / dlsym): libdehaze_hip.so carries no link-time dependency on it, a process that already
// holds an RCCL (PyTorch ships one) shares that copy, and a single-GPU user never loads it.
//   replaces: nn.DataParallel's gradient reduction, My_train.py:97  (one process per GPU, batch-axis sharding, SUM all-reduce of the
//   flat fp32 gradient buckets on the caller's HIP stream; the 1/world factor is folded into dhz_adamw_step's grad_scale)
#include <dlfcn.h>
#include <stdio.h>
#include <string.h>
#include "common.h"

namespace {

// the slice of rccl.h this file uses (ABI-stable: NCCL 2.x)
typedef struct ncclComm* ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef int ncclResult_t;                      // ncclSuccess == 0
constexpr int kNcclFloat = 7, kNcclSum = 0;    // ncclFloat32, ncclSum

struct Rccl {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    bool ok = false;
    char why[256] = "";                          // why loading failed (kept once: dlerror() clears itself when read)
};

const Rccl& rccl() {
    static const Rccl r = [] {
        Rccl x;
        const char* names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so.1"};
        for (const char* n : names)                                   // a copy the process already holds (PyTorch's) first
            if (!x.handle) x.handle = dlopen(n, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL);
        for (const char* n : names)
            if (!x.handle) x.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (!x.handle) {                                              // dlerror() clears itself when read: keep the text once
            const char* e = dlerror();
            snprintf(x.why, sizeof(x.why), "dlopen(librccl.so): %s", e ? e : "no such library");
            return x;
        }
        x.GetUniqueId = (decltype(x.GetUniqueId))dlsym(x.handle, "ncclGetUniqueId");
        x.CommInitRank = (decltype(x.CommInitRank))dlsym(x.handle, "ncclCommInitRank");
        x.AllReduce = (decltype(x.AllReduce))dlsym(x.handle, "ncclAllReduce");
        x.CommDestroy = (decltype(x.CommDestroy))dlsym(x.handle, "ncclCommDestroy");
        x.GetErrorString = (decltype(x.GetErrorString))dlsym(x.handle, "ncclGetErrorString");
        x.ok = x.GetUniqueId && x.CommInitRank && x.AllReduce && x.CommDestroy;
        if (!x.ok)
            snprintf(x.why, sizeof(x.why), "library loaded, symbol missing:%s%s%s%s", x.GetUniqueId ? "" : " ncclGetUniqueId",
                     x.CommInitRank ? "" : " ncclCommInitRank", x.AllReduce ? "" : " ncclAllReduce", x.CommDestroy ? "" : " ncclCommDestroy");
        return x;
    }();
    return r;
}

int fail(const char* who, ncclResult_t rc) {
    const Rccl& r = rccl();
    dhz_set_error("%s: RCCL error %d (%s)", who, (int)rc, r.GetErrorString ? r.GetErrorString(rc) : "?");
    return DHZ_ELAUNCH;
}

}  // namespace

extern "C" int dhz_comm_unique_id(void* id128) {
    const char* who = "dhz_comm_unique_id";
    DHZ_REQUIRE(id128, "%s: null pointer", who);
    DHZ_REQUIRE(rccl().ok, "%s: RCCL could not be loaded: %s", who, rccl().why);
    ncclUniqueId id;
    const ncclResult_t rc = rccl().GetUniqueId(&id);
    if (rc) return fail(who, rc);
    memcpy(id128, &id, sizeof(id));
    return DHZ_OK;
}

extern "C" int dhz_comm_init(void** comm, int rank, int nranks, const void* id128) {
    const char* who = "dhz_comm_init";
    DHZ_REQUIRE(comm && id128, "%s: null pointer", who);
    DHZ_REQUIRE(nranks >= 1 && rank >= 0 && rank < nranks, "%s: rank %d of %d", who, rank, nranks);
    DHZ_REQUIRE(rccl().ok, "%s: RCCL could not be loaded: %s", who, rccl().why);
    ncclUniqueId id;
    memcpy(&id, id128, sizeof(id));
    ncclComm_t c = nullptr;
    const ncclResult_t rc = rccl().CommInitRank(&c, nranks, id, rank);      // binds the CURRENT HIP device (hipSetDevice first)
    if (rc) return fail(who, rc);
    *comm = c;
    return DHZ_OK;
}

extern "C" int dhz_comm_allreduce_sum_f32(void* comm, float* buf, int64_t n, void* stream) {
    const char* who = "dhz_comm_allreduce_sum_f32";
    DHZ_REQUIRE(comm && buf && n > 0, "%s: null pointer or n=%lld", who, (long long)n);
    DHZ_REQUIRE(rccl().ok, "%s: RCCL could not be loaded: %s", who, rccl().why);
    const ncclResult_t rc = rccl().AllReduce(buf, buf, (size_t)n, kNcclFloat, kNcclSum, (ncclComm_t)comm, (hipStream_t)stream);
    if (rc) return fail(who, rc);
    return DHZ_OK;
}

extern "C" int dhz_comm_destroy(void* comm) {
    const char* who = "dhz_comm_destroy";
    DHZ_REQUIRE(comm, "%s: null pointer", who);
    DHZ_REQUIRE(rccl().ok, "%s: RCCL could not be loaded: %s", who, rccl().why);
    const ncclResult_t rc = rccl().CommDestroy((ncclComm_t)comm);
    if (rc) return fail(who, rc);
    return DHZ_OK;
}
