// The multi-GPU exchange steps behind the C ABI, over RCCL (one process per GPU, point-to-point xGMI links).
//
// The path itself shards without communication (a pixel-sample depends only on (x, y, frame index)); what is
// exchanged is the per-GPU result, once per render: for a sample split the per-pixel Welford moments (28 B per
// pixel), for a tile partition the disjoint accumulation buffers.  RCCL has no user-defined reduction, so the moment
// merge is a slice exchange (every rank sends slice j of its moments to rank j: all peer links busy at once, 1/N of
// the frame per link), Chan's update applied by ptx_merge_kernel in fixed rank order (deterministic), and a second
// slice exchange that hands every rank the merged slices.  The library resolves RCCL at run time (dlopen) and has
// no link-time dependency on it: single-GPU hosts never load it.
#pragma once
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <stdint.h>
#include <string>

namespace coll {

struct Api {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
    ncclResult_t (*CommUserRank)(const ncclComm_t, int*) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    std::string error;
};

// One RCCL per process: an instance some other component has already loaded (a framework's bundled copy) is reused,
// otherwise ROCm's is loaded.
inline Api& state() {
    static Api a;
    return a;
}
inline Api* api() {
    Api& a = state();
    static bool tried = false;
    if (tried) return a.handle ? &a : nullptr;
    tried = true;
    const char* loaded[] = {"librccl.so", "librccl.so.1"};
    for (const char* n : loaded)
        if (!a.handle) a.handle = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
    const char* fresh[] = {"librccl.so.1", "/opt/rocm/lib/librccl.so.1", "librccl.so"};
    for (const char* n : fresh)
        if (!a.handle) a.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL);
    if (!a.handle) {
        const char* why = dlerror();  // one call: it returns the message and clears it
        a.error = std::string("RCCL not found: ") + (why ? why : "dlopen failed");
        return nullptr;
    }
    bool ok = true;
    auto sym = [&](const char* name) { void* p = dlsym(a.handle, name); if (!p) { ok = false; a.error = std::string("RCCL symbol missing: ") + name; } return p; };
    a.GetUniqueId = (decltype(a.GetUniqueId))sym("ncclGetUniqueId");
    a.CommInitRank = (decltype(a.CommInitRank))sym("ncclCommInitRank");
    a.CommDestroy = (decltype(a.CommDestroy))sym("ncclCommDestroy");
    a.CommCount = (decltype(a.CommCount))sym("ncclCommCount");
    a.CommUserRank = (decltype(a.CommUserRank))sym("ncclCommUserRank");
    a.GroupStart = (decltype(a.GroupStart))sym("ncclGroupStart");
    a.GroupEnd = (decltype(a.GroupEnd))sym("ncclGroupEnd");
    a.Send = (decltype(a.Send))sym("ncclSend");
    a.Recv = (decltype(a.Recv))sym("ncclRecv");
    a.AllReduce = (decltype(a.AllReduce))sym("ncclAllReduce");
    a.GetErrorString = (decltype(a.GetErrorString))sym("ncclGetErrorString");
    if (!ok) { dlclose(a.handle); a.handle = nullptr; return nullptr; }
    return &a;
}

// Why api() returned null.
inline std::string api_error() {
    (void)api();
    return state().error.empty() ? std::string("RCCL unavailable") : state().error;
}

// Contiguous split of n pixels into `world` slices whose sizes differ by at most one (the first n % world are longer).
inline void slice_of(uint64_t n, uint32_t world, uint32_t r, uint64_t* first, uint64_t* count) {
    const uint64_t base = n / world, extra = n % world;
    *count = base + (r < extra ? 1u : 0u);
    *first = (uint64_t)r * base + (r < extra ? r : extra);
}

}  // namespace coll
