// Test infrastructure: an in-process stand-in for RCCL ("librccl.so.1"), for boxes with one GPU.
//
// libptx355.so resolves RCCL at run time and adopts a copy the process has already loaded.  A test process loads THIS
// library first; the ranks of a "world" are then threads of that process, each with its own ptx context on the same
// GPU, and the sends, receives and all-reduces the product issues are matched through a table in this library and
// carried out as device copies / a sum kernel.  Only what the product calls is implemented: ncclGetUniqueId,
// ncclCommInitRank / Destroy / Count / UserRank, ncclGroupStart / End, ncclSend, ncclRecv, ncclAllReduce (float sum),
// ncclGetErrorString.  Semantics kept from RCCL: operations are ordered with the stream they are given (this
// implementation synchronises the stream where RCCL would enqueue), the k-th send of rank a to rank b inside a group
// meets the k-th receive of b from a, counts must agree, and a group is per thread.
// Never linked into the product and never loaded by it on its own.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <chrono>
#include <condition_variable>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

namespace {

struct Op { bool send; void* buf; size_t bytes; int peer; hipStream_t stream; };
struct Reduce { const void* send; void* recv; size_t count; hipStream_t stream; };

struct World {
    int n = 0;
    std::mutex mu;
    std::condition_variable cv;
    int arrived = 0;
    uint64_t generation = 0;
    int joined = 0, left = 0;
    std::vector<std::vector<Op>> posted;   // per rank: the operations of its current group
    std::vector<Reduce> reduces;           // per rank: its current all-reduce
    // all ranks of the world meet here; false when some rank does not come (a rank that failed earlier, a test bug)
    bool barrier() {
        std::unique_lock<std::mutex> lk(mu);
        const uint64_t gen = generation;
        if (++arrived == n) { arrived = 0; generation++; cv.notify_all(); return true; }
        const char* ms = getenv("LOOPBACK_TIMEOUT_MS");
        const bool ok = cv.wait_for(lk, std::chrono::milliseconds(ms ? atol(ms) : 60000L), [&] { return generation != gen; });
        if (!ok) arrived--;  // this rank gives up: the count stays right for those who come later
        return ok;
    }
};

std::mutex g_mu;
std::map<std::string, std::shared_ptr<World>> g_worlds;
uint64_t g_next_id = 1;
int g_fail_sends = 0;  // test hook: this many of the next ncclSend calls fail

thread_local int tl_depth = 0;
thread_local std::vector<std::pair<ncclComm_t, Op>> tl_ops;

}  // namespace

struct ncclComm {
    std::shared_ptr<World> world;
    std::string key;
    int rank = 0;
};

__global__ void loopback_sum_kernel(float* out, const float* const* ins, int n, size_t count) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    float s = ins[0][i];
    for (int r = 1; r < n; r++) s += ins[r][i];  // rank order
    out[i] = s;
}

static ncclResult_t run_group(std::vector<std::pair<ncclComm_t, Op>>& ops) {
    if (ops.empty()) return ncclSuccess;
    ncclComm_t comm = ops[0].first;
    for (auto& o : ops)
        if (o.first != comm) return ncclInvalidUsage;  // one communicator per group is all the product uses
    World& w = *comm->world;
    ncclResult_t result = ncclSuccess;
    for (auto& o : ops)
        if (hipStreamSynchronize(o.second.stream) != hipSuccess) result = ncclUnhandledCudaError;
    {
        std::lock_guard<std::mutex> lk(w.mu);
        w.posted[comm->rank].clear();
        for (auto& o : ops) w.posted[comm->rank].push_back(o.second);
    }
    if (!w.barrier()) return ncclSystemError;
    // every receive of this rank: the k-th from peer p is the k-th send of p to this rank
    std::vector<size_t> taken(w.n, 0);
    for (auto& o : ops) {
        if (o.second.send) continue;
        const int p = o.second.peer;
        const Op* match = nullptr;
        size_t seen = 0;
        for (const Op& s : w.posted[p]) {
            if (!s.send || s.peer != comm->rank) continue;
            if (seen++ == taken[p]) { match = &s; break; }
        }
        taken[p]++;
        if (!match || match->bytes != o.second.bytes) { result = ncclInvalidArgument; continue; }
        if (o.second.bytes && hipMemcpy(o.second.buf, match->buf, o.second.bytes, hipMemcpyDeviceToDevice) != hipSuccess) result = ncclUnhandledCudaError;
    }
    // a send nobody received is an error for the sender as well
    for (auto& o : ops) {
        if (!o.second.send) continue;
        size_t sends_to = 0, recvs_from = 0;
        for (auto& q : ops) if (q.second.send && q.second.peer == o.second.peer) sends_to++;
        for (const Op& r : w.posted[o.second.peer]) if (!r.send && r.peer == comm->rank) recvs_from++;
        if (sends_to != recvs_from) result = ncclInvalidArgument;
    }
    if (!w.barrier()) return ncclSystemError;  // the senders' buffers are free again
    return result;
}

extern "C" {

// Test hooks.
void loopback_fail_next_sends(int n) { std::lock_guard<std::mutex> lk(g_mu); g_fail_sends = n; }
int loopback_group_depth() { return tl_depth; }
int loopback_worlds() { std::lock_guard<std::mutex> lk(g_mu); return (int)g_worlds.size(); }

ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
    if (!id) return ncclInvalidArgument;
    std::lock_guard<std::mutex> lk(g_mu);
    memset(id->internal, 0, sizeof(id->internal));
    const uint64_t v = g_next_id++;
    memcpy(id->internal, "loopback", 8);
    memcpy(id->internal + 8, &v, 8);
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t* comm, int nranks, ncclUniqueId id, int rank) {
    if (!comm || nranks < 1 || rank < 0 || rank >= nranks) return ncclInvalidArgument;
    std::shared_ptr<World> w;
    const std::string key(id.internal, sizeof(id.internal));
    {
        std::lock_guard<std::mutex> lk(g_mu);
        auto it = g_worlds.find(key);
        if (it == g_worlds.end()) {
            w = std::make_shared<World>();
            w->n = nranks;
            w->posted.resize(nranks);
            w->reduces.resize(nranks);
            g_worlds[key] = w;
        } else {
            w = it->second;
            if (w->n != nranks) return ncclInvalidArgument;
        }
        w->joined++;
    }
    ncclComm* c = new ncclComm();
    c->world = w;
    c->key = key;
    c->rank = rank;
    if (!w->barrier()) { delete c; return ncclSystemError; }  // collective: returns when every rank has called
    *comm = c;
    return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm) {
    if (!comm) return ncclSuccess;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        if (++comm->world->left == comm->world->n) g_worlds.erase(comm->key);
    }
    delete comm;
    return ncclSuccess;
}

ncclResult_t ncclCommCount(const ncclComm_t comm, int* count) {
    if (!comm || !count) return ncclInvalidArgument;
    *count = comm->world->n;
    return ncclSuccess;
}

ncclResult_t ncclCommUserRank(const ncclComm_t comm, int* rank) {
    if (!comm || !rank) return ncclInvalidArgument;
    *rank = comm->rank;
    return ncclSuccess;
}

ncclResult_t ncclGroupStart() { tl_depth++; return ncclSuccess; }

ncclResult_t ncclGroupEnd() {
    if (tl_depth <= 0) return ncclInvalidUsage;
    if (--tl_depth > 0) return ncclSuccess;
    std::vector<std::pair<ncclComm_t, Op>> ops;
    ops.swap(tl_ops);
    return run_group(ops);
}

static ncclResult_t enqueue(bool send, void* buf, size_t count, ncclDataType_t type, int peer, ncclComm_t comm, hipStream_t stream) {
    if (!comm || peer < 0 || peer >= comm->world->n || (count && !buf)) return ncclInvalidArgument;
    if (type != ncclFloat) return ncclInvalidArgument;
    Op o{send, buf, count * sizeof(float), peer, stream};
    tl_ops.push_back({comm, o});
    if (tl_depth == 0) {  // outside a group: a group of one
        std::vector<std::pair<ncclComm_t, Op>> ops;
        ops.swap(tl_ops);
        return run_group(ops);
    }
    return ncclSuccess;
}

ncclResult_t ncclSend(const void* buf, size_t count, ncclDataType_t type, int peer, ncclComm_t comm, hipStream_t stream) {
    {
        std::lock_guard<std::mutex> lk(g_mu);
        if (g_fail_sends > 0) { g_fail_sends--; return ncclInternalError; }
    }
    return enqueue(true, const_cast<void*>(buf), count, type, peer, comm, stream);
}

ncclResult_t ncclRecv(void* buf, size_t count, ncclDataType_t type, int peer, ncclComm_t comm, hipStream_t stream) {
    return enqueue(false, buf, count, type, peer, comm, stream);
}

ncclResult_t ncclAllReduce(const void* sendbuff, void* recvbuff, size_t count, ncclDataType_t type, ncclRedOp_t op, ncclComm_t comm, hipStream_t stream) {
    if (!comm || (count && (!sendbuff || !recvbuff))) return ncclInvalidArgument;
    if (type != ncclFloat || op != ncclSum) return ncclInvalidArgument;
    World& w = *comm->world;
    ncclResult_t result = ncclSuccess;
    if (hipStreamSynchronize(stream) != hipSuccess) result = ncclUnhandledCudaError;
    {
        std::lock_guard<std::mutex> lk(w.mu);
        w.reduces[comm->rank] = Reduce{sendbuff, recvbuff, count, stream};
    }
    if (!w.barrier()) return ncclSystemError;
    float* tmp = nullptr;
    const float** d_ins = nullptr;
    std::vector<const float*> ins(w.n);
    for (int r = 0; r < w.n; r++) {
        if (w.reduces[r].count != count) result = ncclInvalidArgument;
        ins[r] = (const float*)w.reduces[r].send;
    }
    if (result == ncclSuccess && count) {
        if (hipMalloc((void**)&tmp, count * sizeof(float)) != hipSuccess || hipMalloc((void**)&d_ins, w.n * sizeof(float*)) != hipSuccess ||
            hipMemcpy(d_ins, ins.data(), w.n * sizeof(float*), hipMemcpyHostToDevice) != hipSuccess) {
            result = ncclUnhandledCudaError;
        } else {
            hipLaunchKernelGGL(loopback_sum_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, stream, tmp, (const float* const*)d_ins, w.n, count);
            if (hipStreamSynchronize(stream) != hipSuccess) result = ncclUnhandledCudaError;
        }
    }
    if (!w.barrier()) result = ncclSystemError;  // every rank has read every send buffer: in-place results may be written
    if (result == ncclSuccess && count && hipMemcpy(recvbuff, tmp, count * sizeof(float), hipMemcpyDeviceToDevice) != hipSuccess) result = ncclUnhandledCudaError;
    if (tmp) (void)hipFree(tmp);
    if (d_ins) (void)hipFree((void*)d_ins);
    return result;
}

const char* ncclGetErrorString(ncclResult_t r) {
    switch (r) {
        case ncclSuccess: return "no error";
        case ncclUnhandledCudaError: return "loop-back: HIP error";
        case ncclSystemError: return "loop-back: a rank did not arrive at the rendezvous";
        case ncclInternalError: return "loop-back: injected failure";
        case ncclInvalidArgument: return "loop-back: invalid argument or unmatched send / receive";
        case ncclInvalidUsage: return "loop-back: invalid usage";
        default: return "loop-back: error";
    }
}

}  // extern "C"
