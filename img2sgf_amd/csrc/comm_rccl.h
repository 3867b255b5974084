// Multi-GPU exchange of the board records (BASELINE configs[3]): one process per GPU, contiguous shards of the
// batch, NO data-path collective; the only exchange is one all-gather of the 384-byte i2s_board records, device to
// device over RCCL (xGMI), on the context's stream.  The reference has no counterpart (it is single-process).
// librccl is opened lazily with dlopen the first time a communicator is asked for (it is a 0.5 GB library that the
// single-GPU path never needs); if torch already mapped its own librccl.so.1 the same copy is reused.
#pragma once
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>

#include <mutex>

#include "../../include/i2s.h"

namespace i2s {

// the slice of the NCCL/RCCL C API used here (rccl.h: ncclUniqueId is 128 opaque bytes passed BY VALUE, ncclUint8 = 1)
struct RcclId { char internal[I2S_COMM_ID_BYTES]; };
typedef void* rccl_comm_t;
struct RcclApi {
    void* handle = nullptr;
    int (*GetUniqueId)(RcclId*) = nullptr;
    int (*CommInitRank)(rccl_comm_t*, int, RcclId, int) = nullptr;
    int (*CommDestroy)(rccl_comm_t) = nullptr;
    int (*AllGather)(const void*, void*, size_t, int, rccl_comm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    char err[256] = {0};
};

static void rccl_open(RcclApi& api);

// the library is opened once per process, whichever thread asks first
static RcclApi& rccl_state()
{
    static RcclApi api;
    static std::once_flag once;
    std::call_once(once, [] { rccl_open(api); });
    return api;
}
static RcclApi* rccl_api() { RcclApi& a = rccl_state(); return a.handle ? &a : nullptr; }
// why rccl_api() returned null
static const char* rccl_open_error() { return rccl_state().err; }

static void rccl_open(RcclApi& api)
{
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void* h = nullptr;
    for (const char* n : names) if ((h = dlopen(n, RTLD_NOW | RTLD_LOCAL))) break;
    if (!h) { snprintf(api.err, sizeof(api.err), "librccl not found: %s", dlerror()); return; }
    api.GetUniqueId = (int (*)(RcclId*))dlsym(h, "ncclGetUniqueId");
    api.CommInitRank = (int (*)(rccl_comm_t*, int, RcclId, int))dlsym(h, "ncclCommInitRank");
    api.CommDestroy = (int (*)(rccl_comm_t))dlsym(h, "ncclCommDestroy");
    api.AllGather = (int (*)(const void*, void*, size_t, int, rccl_comm_t, hipStream_t))dlsym(h, "ncclAllGather");
    api.GetErrorString = (const char* (*)(int))dlsym(h, "ncclGetErrorString");
    if (!api.GetUniqueId || !api.CommInitRank || !api.CommDestroy || !api.AllGather || !api.GetErrorString) {
        snprintf(api.err, sizeof(api.err), "librccl lacks an ncclGetUniqueId/CommInitRank/CommDestroy/AllGather/GetErrorString symbol");
        dlclose(h);
        return;
    }
    api.handle = h;
}

}  // namespace i2s

struct i2s_comm {
    int device = 0, world = 1, rank = 0, cap = 0;     // cap = records per rank in the gather buffer
    i2s::rccl_comm_t comm = nullptr;
    i2s_board* d_all = nullptr;                        // [world][cap] records; rank r's shard starts at d_all + r * cap
    char err[256] = {0};
};
